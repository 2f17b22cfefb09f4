"""CPU oracle for `run_midas.py genes` (SURVEY.md 8f "next" #4): reads per pangenome gene, depth, copy number.

TEST INFRASTRUCTURE ONLY -- never imported from midas_amd/ or scripts/.

Restates /root/reference/midas/run/genes.py: keep_read (:148-163), count_mapped_bp (:165-199), normalize (:201-215),
write_results (:217-244), on plain records instead of pysam objects.  What a BAM record's query_alignment_sequence /
query_length / tags / query_qualities / mapping_quality are is [EXT] pysam (oracle/pileup_oracle.py restates it); what
the reference itself decides -- the filter, the per-gene running fp64 sum in BAM order, the species summaries, the
median marker depth, the text -- is PINNED AGAINST THE REFERENCE ITSELF: tests/golden/make_genes_vectors.py executes the
reference's own functions around a BAM double and commits inputs + outputs (tests/golden/genes_vectors.json).
"""
from collections import defaultdict

import numpy as np


def keep_read(align_len, query_len, nm, quals, mapq, min_pid, min_readq, min_mapq, min_aln_cov):
    """genes.py:148-163; nm None / quals None raise like dict(aln.tags)['NM'] / np.mean(None) would."""
    if nm is None:
        raise KeyError('NM')
    if 100 * (align_len - nm) / float(align_len) < min_pid:
        return False
    elif np.mean(quals) < min_readq:
        return False
    elif mapq < min_mapq:
        return False
    elif align_len / float(query_len) < min_aln_cov:
        return False
    return True


def count_mapped_bp(args, reads, gene_ids, gene_species, gene_length):
    """genes.py:165-199.  reads: iterable of (gene index, align_len, query_len, nm, quals, mapq) in BAM order.
    -> per-gene aligned, mapped, depth; per-species aligned, mapped, covered_genes, mean_coverage, fraction_covered."""
    n = len(gene_ids)
    aligned, mapped, depth = [0] * n, [0] * n, [0.0] * n
    sp_aligned, sp_mapped = defaultdict(int), defaultdict(int)
    for g, align_len, query_len, nm, quals, mapq in reads:
        sp_aligned[gene_species[g]] += 1
        aligned[g] += 1
        if not keep_read(align_len, query_len, nm, quals, mapq, args['mapid'], args['readq'], args['mapq'], args['aln_cov']):
            continue
        sp_mapped[gene_species[g]] += 1
        mapped[g] += 1
        depth[g] += align_len / float(gene_length[g])
    species = {}
    for sp in dict.fromkeys(gene_species):
        d = [depth[g] for g in range(n) if gene_species[g] == sp]
        nz = [x for x in d if x > 0]
        size = len(d)
        species[sp] = dict(pangenome_size=size, aligned_reads=sp_aligned[sp], mapped_reads=sp_mapped[sp],
                           covered_genes=len(nz), mean_coverage=np.mean(nz) if len(nz) > 0 else 0,
                           fraction_covered=len(nz) / float(size))
    return aligned, mapped, depth, species


def normalize(depth, gene_species, gene_marker, species):
    """genes.py:201-215 -> copies per gene; sets species[...]['marker_coverage'] (np.median of the markers' depths)."""
    markers = {sp: defaultdict(float) for sp in species}
    for g, m in enumerate(gene_marker):
        if m is not None:
            markers[gene_species[g]][m] += depth[g]
    for sp in species:
        species[sp]['marker_coverage'] = np.median(list(markers[sp].values()))
    copies = [0.0] * len(depth)
    for g in range(len(depth)):
        mc = species[gene_species[g]]['marker_coverage']
        if mc > 0:
            copies[g] = depth[g] / mc
    return copies


def write_results(gene_ids, gene_species, mapped, depth, copies, species):
    """genes.py:217-244 -> ({species: text of <species>.genes}, text of summary.txt)"""
    out = {sp: '\t'.join(['gene_id', 'count_reads', 'coverage', 'copy_number']) + '\n' for sp in species}
    for g in sorted(range(len(gene_ids)), key=lambda k: gene_ids[k]):
        out[gene_species[g]] += '\t'.join(str(_) for _ in [gene_ids[g], mapped[g], depth[g], copies[g]]) + '\n'
    header = ['species_id', 'pangenome_size', 'covered_genes', 'fraction_covered', 'mean_coverage', 'marker_coverage',
              'aligned_reads', 'mapped_reads']
    summary = '\t'.join(header) + '\n'
    for sp, s in species.items():
        summary += '\t'.join(str(_) for _ in [sp, s['pangenome_size'], s['covered_genes'], s['fraction_covered'],
                                               s['mean_coverage'], s['marker_coverage'], s['aligned_reads'],
                                               s['mapped_reads']]) + '\n'
    return out, summary
