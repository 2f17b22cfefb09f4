"""`run_midas.py genes` pinned against the REFERENCE'S OWN count_mapped_bp / keep_read / normalize / write_results:
tests/golden/genes_vectors.json holds seeded inputs and what those functions produced (generated in the build container
by tests/golden/make_genes_vectors.py, which executes them from /root/reference around a BAM double -- pysam stays [EXT]).
Held to it: the oracle, the product's host code, and on the GPU box midas_genes_count through the C-ABI."""
import json
import os

import numpy as np
import pytest

from midas_amd import abi
from oracle import genes_oracle as go
from oracle import pileup_oracle as po
from tests import helpers as H

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(HERE, "golden", "genes_vectors.json")) as h:
        return json.load(h)


def test_oracle_matches_the_reference(vectors):
    for c in vectors['cases']:
        alns = po.alns_from_soa(H.reads_from_dicts(c['reads']).as_dict())
        recs = []
        for aln, rid in zip(alns, c['ref_ids']):
            a = max(0, po.query_alignment_end(aln) - po.query_alignment_start(aln))
            recs.append((rid, a, len(aln.seq), aln.nm, aln.qual, aln.mapq))
        aligned, mapped, depth, species = go.count_mapped_bp(c['args'], recs, c['gene_ids'], c['gene_species'], c['gene_length'])
        assert [repr(x) for x in depth] == c['gene_depth']
        assert mapped == c['gene_mapped'] and aligned == c['gene_aligned']
        copies = go.normalize(depth, c['gene_species'], c['gene_marker'], species)
        tables, summary = go.write_results(c['gene_ids'], c['gene_species'], mapped, depth, copies, species)
        assert tables == c['tables'] and summary == c['summary']


@pytest.mark.gpu
def test_device_counts_and_product_text_match_the_reference(vectors, tmp_path):
    from midas_amd.run import genes as mgenes
    with abi.Context(0) as ctx:
        for n, c in enumerate(vectors['cases']):
            reads = H.reads_from_dicts(c['reads'])
            thr = abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, **c['args']))
            aligned, mapped, depth, ms = ctx.genes_count(thr, reads, c['ref_ids'], c['gene_length'])
            assert [repr(float(x)) for x in depth] == c['gene_depth']
            assert mapped.tolist() == c['gene_mapped'] and aligned.tolist() == c['gene_aligned']
            # the product's host side: species summaries, copy numbers, text
            out = tmp_path / ("case%d" % n)
            os.makedirs(out / "genes" / "output")
            species = {sp: mgenes.Species(sp) for sp in c['species_ids']}
            genes = {}
            for gid, sp, ln, mk in zip(c['gene_ids'], c['gene_species'], c['gene_length'], c['gene_marker']):
                g = mgenes.Gene(gid, sp, ln)
                g.marker_id = mk
                genes[gid] = g
                species[sp].pangenome_size += 1
            mgenes.fold_counts(species, genes, c['gene_ids'], aligned, mapped, depth)
            mgenes.normalize({}, species, genes)
            mgenes.write_results(dict(outdir=str(out)), species, genes)
            import gzip
            for sp in c['species_ids']:
                assert gzip.open(out / "genes" / "output" / (sp + ".genes.gz"), "rt").read() == c['tables'][sp]
            assert open(out / "genes" / "summary.txt").read() == c['summary']
