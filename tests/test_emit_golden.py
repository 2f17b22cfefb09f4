"""The emit side of `run_midas.py snps` pinned against the REFERENCE'S OWN species_pileup / keep_read / snps_summary:
tests/golden/emit_vectors.json holds seeded inputs and the text those functions wrote (generated in the build container
by tests/golden/make_emit_vectors.py, which executes them from /root/reference around a count_coverage double -- pysam
itself stays [EXT]).  Held to it: the oracle, and on the GPU box the product end to end through the C-ABI and the native
table writer."""
import gzip
import json
import os

import numpy as np
import pytest

from midas_amd import abi
from oracle import pileup_oracle as po

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(HERE, "golden", "emit_vectors.json")) as h:
        return json.load(h)


def inputs(case):
    dt = abi._SOA_DTYPES
    reads = abi.ReadsSoA(**{k: np.array(v, dtype=dt[k]) for k, v in case['reads'].items()})
    species_ids = sorted(set(case['contig_species']), key=case['contig_species'].index)
    ref = np.frombuffer("".join(case['contig_seqs']).encode(), dtype=np.uint8)
    contigs = abi.ContigTable(length=[len(s) for s in case['contig_seqs']],
                              species=[species_ids.index(s) for s in case['contig_species']],
                              read_begin=case['read_begin'], ref=ref, n_species=len(species_ids),
                              ids=list(case['contig_ids']), species_ids=species_ids)
    return contigs, reads


def test_oracle_text_and_summary_match_the_reference(vectors):
    for case in vectors['cases']:
        contigs, reads = inputs(case)
        alns = po.alns_from_soa(reads.as_dict())
        oc, by = {}, {}
        for k, cid in enumerate(contigs.ids):
            oc[cid] = po.OContig(id=cid, seq=case['contig_seqs'][k].upper(), species_id=case['contig_species'][k])
            by[cid] = alns[case['read_begin'][k]:case['read_begin'][k + 1]]
        stats = {}
        for sp in contigs.species_ids:
            text, st = po.species_pileup(case['args'], sp, oc, by)
            assert text == case['tables'][sp]
            stats[sp] = st
        assert po.snps_summary_text(stats) == case['summary']


@pytest.mark.gpu
def test_product_tables_and_summary_match_the_reference(vectors, tmp_path):
    from midas_amd.run import snps as msnps
    with abi.Context(0) as ctx:
        for n, case in enumerate(vectors['cases']):
            contigs, reads = inputs(case)
            counts, allele, stats = ctx.pileup(abi.Thresholds.from_args(case['args']), contigs, reads)
            out = tmp_path / ("case%d" % n)
            os.makedirs(out / "snps" / "output")
            args = dict(case['args'], outdir=str(out), threads=3)
            species = {}
            for i, sp in enumerate(contigs.species_ids):
                msnps._write_species(args, sp, contigs, counts, allele)
                assert gzip.open(out / "snps" / "output" / (sp + ".snps.gz"), "rt").read() == case['tables'][sp]
                o = msnps.Species(sp)
                o.genome_length = int(sum(l for l, s in zip(contigs.length, contigs.species) if s == i))
                o.covered_bases = int(stats[i, abi.STAT_COVERED_BASES])
                o.total_depth = int(stats[i, abi.STAT_TOTAL_DEPTH])
                o.aligned_reads = int(stats[i, abi.STAT_ALIGNED_READS])
                o.mapped_reads = int(stats[i, abi.STAT_MAPPED_READS])
                if o.genome_length > 0:
                    o.fraction_covered = o.covered_bases / float(o.genome_length)
                if o.covered_bases > 0:
                    o.mean_coverage = o.total_depth / float(o.covered_bases)
                species[sp] = o
            msnps.snps_summary(args, species)
            assert open(out / "snps" / "summary.txt").read() == case['summary']
