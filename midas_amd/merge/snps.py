"""MI355X-native `merge_midas.py snps`: the host side (SURVEY.md 8f "next" #1).

Mirrors /root/reference/midas/merge/snps.py: same outputs (<outdir>/<species>/snps_{info,freq,depth,summary}.txt,
readme.txt), same site numbering and filters.  What changes underneath: the per-site cross-sample arithmetic of
GenomicSite (pooled counts, allele calls, per-sample depth/MAF, prevalence, flag -- :13-114) runs on the GPU for all
sites of a species at once (midas_merge_sites in include/midas_snps.h); the temporary acgt_counts matrices and the
per-thread shard files of the reference are not needed.  Annotation (:116-174) and text emission (:176-201) stay on
the host and touch only the sites that survive the filters.  No CPU fallback for the arithmetic.
"""

import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from midas_amd import abi, dist
from midas_amd.merge import annotate, merge


def replace_none(input_string, replace_string="NA"):
    return input_string if input_string is not None else replace_string


INFO_FIELDS = ['site_id', 'ref_id', 'ref_pos', 'ref_allele', 'major_allele', 'minor_allele', 'count_samples',
               'count_a', 'count_c', 'count_g', 'count_t', 'locus_type', 'gene_id', 'snp_type', 'site_type', 'amino_acids']


def load_sample_tables(species, args):
    """read_run_midas_snps + the zip of build_temp_count_matrix (midas/merge/snps.py:236-271), without the
    temporary matrices: per sample the [n_sites,4] counts, plus the site keys of the first sample."""
    max_rows = -1 if args['max_sites'] == float('Inf') else int(args['max_sites'])
    paths = ['%s/snps/output/%s.snps.gz' % (s.dir, species.id) for s in species.samples]
    nthreads = max(1, min(len(paths), int(args.get('threads', 1) or 1)))
    with ThreadPoolExecutor(nthreads) as ex:
        futs = [ex.submit(abi.read_snps_table, p, max_rows, i == 0) for i, p in enumerate(paths)]
        tabs = [f.result() for f in futs]
    n = min(t[0].shape[0] for t in tabs)     # the reference's zip stops at the shortest file
    counts = [np.ascontiguousarray(t[0][:n]) for t in tabs]
    return counts, tabs[0][1], tabs[0][2][:n + 1]


def merge_species(species, args, ctx):
    """build_sharded_tables + merge_sharded_tables (midas/merge/snps.py:324-420) for one species."""
    counts, keys, key_off = load_sample_tables(species, args)
    n = counts[0].shape[0]
    prm = abi.MergeParams.from_args(args)
    try:
        res = ctx.merge_sites(prm, counts, species.sample_depth)
    except abi.MidasSnpsError as e:
        sys.exit("\nError: %s\n" % e.message)
    genes = annotate.GeneCursor.from_db(species.id, args['db'])
    keep = np.nonzero(res['flag'] == 0)[0]
    outdir = '%s/%s' % (args['outdir'], species.id)
    # snps_freq.txt / snps_depth.txt: one number per (kept site, sample) -- formatted natively
    header = '\t'.join(['site_id'] + [s.id for s in species.samples]) + '\n'
    threads = int(args.get('threads', 1) or 1)
    abi.write_merge_matrix(outdir + '/snps_freq.txt', header, keep, res['depth'], res['minor_count'], threads=threads)
    abi.write_merge_matrix(outdir + '/snps_depth.txt', header, keep, res['depth'], None, threads=threads)
    # snps_info.txt: annotation of the kept sites (the reference's forward cursor over the sorted genes, codon
    # degeneracy) + the per-site calls, formatted natively
    abi.write_merge_info(outdir + '/snps_info.txt', '\t'.join(INFO_FIELDS) + '\n', keep, keys, key_off, res, genes.genes,
                         threads=threads)
    return n, len(keep), res['kernel_ms']


README = """merge_midas.py snps -- files in this directory (species %s)

snps_info.txt     one row per kept site: site_id (row number in the per-sample tables), ref_id, ref_pos, ref_allele,
                  major_allele, minor_allele (by pooled count over the samples), count_samples (samples that pass the
                  depth filters at the site), count_a/c/g/t (pooled), locus_type / gene_id / site_type / amino_acids
                  (annotation from the species' genome.features: CDS, degeneracy 1D-4D, the four possible residues),
                  snp_type (mono, bi, tri, quad)
snps_freq.txt     site_id, then per sample the minor-allele frequency minor / (major + minor), 0 when uncovered
snps_depth.txt    site_id, then per sample the reads on the major + minor allele
snps_summary.txt  the samples' own summary rows for this species (from run_midas.py snps)

Genome and gene annotation of the species: %s/rep_genomes/%s
"""


def write_snps_readme(args, sp):
    with open('%s/%s/readme.txt' % (args['outdir'], sp.id), 'w') as handle:
        handle.write(README % (sp.id, args['db'], sp.id))


def run_pipeline(args):
    """midas/merge/snps.py:471-508"""
    rank, ws = dist.init_from_env()
    if rank == 0:
        print("Identifying species and samples")
    species_list = merge.select_species(args, dtype='snps')
    if rank == 0:
        for species in species_list:
            print("  %s" % species.id)
            print("    count samples: %s" % len(species.samples))
        print("\nMerging snps")
    error = None
    try:        # a rank that fails takes the others down with it at the end, instead of leaving them at the barrier
        ctx = abi.Context(int(os.environ.get("LOCAL_RANK", "0")))
        for species in merge.species_for_rank(species_list, rank, ws):
            print("  %s" % species.id)
            print("    calling SNPs")
            n, kept, ms = merge_species(species, args, ctx)
            print("    %d sites, %d written (%.3f ms on the GPU)" % (n, kept, ms))
            print("    finishing")
            write_snps_readme(args, species)
            species.write_sample_info(dtype='snps', outdir=args['outdir'])
        ctx.close()
    except abi.MidasSnpsError as e:
        error = "\nError: %s\n" % e.message
    except SystemExit as e:
        error = str(e.code)
    dist.agree_or_exit(error)
