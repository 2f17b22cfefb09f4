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


def _table_paths(species):
    return ['%s/snps/output/%s.snps.gz' % (s.dir, species.id) for s in species.samples]


def load_sample_tables(species, args, row_range=None):
    """read_run_midas_snps + the zip of build_temp_count_matrix (midas/merge/snps.py:236-271), without the
    temporary matrices: per sample the [n_sites,4] counts, plus the site keys of the first sample.  row_range = (lo, hi):
    only those rows of every table (one rank's share of a site-sharded merge)."""
    max_rows = -1 if args['max_sites'] == float('Inf') else int(args['max_sites'])
    lo, hi = row_range if row_range is not None else (0, max_rows)
    paths = _table_paths(species)
    # tables written by this library announce their rows: every gzip member of every sample is one task of a single
    # parallel region; the site keys come from the first sample's table alone, as in the reference
    with ThreadPoolExecutor(1) as ex:      # the first sample's table (keys + counts) beside the others' counts
        first = ex.submit(abi.read_snps_table, paths[0], hi, True, lo)
        rest = abi.read_snps_counts(paths[1:], lo, hi) if len(paths) > 1 else []
        c0, keys, key_off = first.result()
    if rest is not None:
        n = min([c0.shape[0]] + [c.shape[0] for c in rest])
        return [np.ascontiguousarray(c[:n]) for c in [c0] + rest], keys, key_off[:n + 1]
    # a table written by the reference among them: one table per thread, each read whole
    nthreads = max(1, min(len(paths), int(args.get('threads', 1) or 1)))
    with ThreadPoolExecutor(nthreads) as ex:
        futs = [ex.submit(abi.read_snps_table, p, hi, False, lo) for p in paths[1:]]
        tabs = [(c0, keys, key_off)] + [f.result() for f in futs]
    n = min(t[0].shape[0] for t in tabs)     # the reference's zip stops at the shortest file
    counts = [np.ascontiguousarray(t[0][:n]) for t in tabs]
    return counts, tabs[0][1], tabs[0][2][:n + 1]


def merge_species(species, args, ctx, row_range=None, part=None):
    """build_sharded_tables + merge_sharded_tables (midas/merge/snps.py:324-420) for one species -- or, with row_range and
    part, for one rank's rows of it (the reference shards by line range too, :366-386): the three tables are then written
    as <name>.partNNN, site ids counted from the table's first row, the column header only in part 0."""
    outdir = '%s/%s' % (args['outdir'], species.id)
    suffix = '' if part is None else '.part%03d' % part
    header = '\t'.join(['site_id'] + [s.id for s in species.samples]) + '\n' if part in (None, 0) else ''
    info_header = '\t'.join(INFO_FIELDS) + '\n' if part in (None, 0) else ''
    base = row_range[0] if row_range is not None else 0
    if row_range is not None and row_range[1] <= row_range[0]:      # more ranks than rows: an empty part
        for name, h in (('snps_freq.txt', header), ('snps_depth.txt', header), ('snps_info.txt', info_header)):
            with open('%s/%s%s' % (outdir, name, suffix), 'w') as handle:
                handle.write(h)
        return 0, 0, 0.0
    with ThreadPoolExecutor(1) as ex:      # the gene table is Python work: it runs while the native reader has the cores
        genes_job = ex.submit(annotate.GeneCursor.from_db, species.id, args['db'])
        counts, keys, key_off = load_sample_tables(species, args, row_range)
        genes = genes_job.result()
    n = counts[0].shape[0]
    prm = abi.MergeParams.from_args(args)
    try:
        res = ctx.merge_sites(prm, counts, species.sample_depth)
    except abi.MidasSnpsError as e:
        if e.status == abi.ERR_MERGE_ZERO_MEAN_DEPTH and e.read_index >= 0 and base:
            sys.exit("\nError: %s [row %d of the species' tables]\n" % (e.message, base + e.read_index + 1))
        sys.exit("\nError: %s\n" % e.message)
    keep = np.nonzero(res['flag'] == 0)[0]
    # snps_freq.txt / snps_depth.txt: one number per (kept site, sample) -- formatted natively
    threads = int(args.get('threads', 1) or 1)
    abi.write_merge_matrix(outdir + '/snps_freq.txt' + suffix, header, keep, res['depth'], res['minor_count'], threads=threads,
                           site_id_base=base)
    abi.write_merge_matrix(outdir + '/snps_depth.txt' + suffix, header, keep, res['depth'], None, threads=threads, site_id_base=base)
    # snps_info.txt: annotation of the kept sites (the reference's forward cursor over the sorted genes, codon
    # degeneracy) + the per-site calls, formatted natively
    abi.write_merge_info(outdir + '/snps_info.txt' + suffix, info_header, keep, keys, key_off, res, genes.genes,
                         threads=threads, site_id_base=base)
    return n, len(keep), res['kernel_ms']


def join_parts(species, args, n_parts):
    """The three tables of a site-sharded species: the ranks' parts one after the other (rows are in site order)."""
    import shutil
    outdir = '%s/%s' % (args['outdir'], species.id)
    for name in ('snps_freq.txt', 'snps_depth.txt', 'snps_info.txt'):
        with open('%s/%s.tmp' % (outdir, name), 'wb') as dst:
            for k in range(n_parts):
                with open('%s/%s.part%03d' % (outdir, name, k), 'rb') as src:
                    shutil.copyfileobj(src, dst, 1 << 24)
        os.replace('%s/%s.tmp' % (outdir, name), '%s/%s' % (outdir, name))
        for k in range(n_parts):
            os.remove('%s/%s.part%03d' % (outdir, name, k))


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


def _device_context():
    return abi.Context(int(os.environ.get("LOCAL_RANK", "0")))


def run_pipeline(args, make_context=_device_context):
    """midas/merge/snps.py:471-508.  N ranks: a species whose sample tables say how many rows they hold (every table
    written by this library does) is sharded by SITE RANGE -- rank r reads, merges and writes rows [n r / N, n (r+1) / N) of
    it, so one species with fifty samples keeps every GPU busy, as the reference's line-range shards keep every core
    (:366-386); the parts are concatenated afterwards.  Any other species goes whole to rank (index mod N).
    make_context: tests substitute a CPU double of the device."""
    # (N ranks meet in the output directory: no process group, no torch -- midas_amd/dist.py; they exchange nothing but
    # "still standing" flags: a species is sharded by site range and every rank writes its own part)
    rank, ws = dist.init_from_env(rendezvous_dir=args['outdir'])
    if rank == 0:
        print("Identifying species and samples")
    species_list = merge.select_species(args, dtype='snps')
    if rank == 0:
        for species in species_list:
            print("  %s" % species.id)
            print("    count samples: %s" % len(species.samples))
        print("\nMerging snps")
    max_rows = None if args['max_sites'] == float('Inf') else int(args['max_sites'])
    sharded = {}
    if ws > 1:
        for species in species_list:
            rows = [abi.count_snps_rows(p) for p in _table_paths(species)]      # gzip member headers only: cheap
            if rows and min(rows) >= 0:
                sharded[species.id] = min(rows) if max_rows is None else min(min(rows), max_rows)
    error = None
    try:        # a rank that fails takes the others down with it at the end, instead of leaving them at the barrier
        with make_context() as ctx:
            for k, species in enumerate(species_list):
                if species.id in sharded:
                    n = sharded[species.id]
                    lo, hi = n * rank // ws, n * (rank + 1) // ws
                    rows, kept, ms = merge_species(species, args, ctx, (lo, hi), part=rank)
                    print("  %s: rows %d-%d of %d, %d written (%.3f ms on the GPU)" % (species.id, lo + 1, hi, n, kept, ms))
                elif k % ws == rank:
                    print("  %s" % species.id)
                    print("    calling SNPs")
                    n, kept, ms = merge_species(species, args, ctx)
                    print("    %d sites, %d written (%.3f ms on the GPU)" % (n, kept, ms))
                    print("    finishing")
                    write_snps_readme(args, species)
                    species.write_sample_info(dtype='snps', outdir=args['outdir'])
    except abi.MidasSnpsError as e:
        error = "\nError: %s\n" % e.message
    except SystemExit as e:
        error = dist.exit_message(e)
    except Exception as e:      # (an OSError from the decoder, a MemoryError ...: the other ranks must not wait for this one)
        error = "\nError: %s: %s\n" % (type(e).__name__, e)
    dist.agree_or_exit(error)       # (also: every part is on disk)
    for k, species in enumerate(species_list):
        if species.id in sharded and k % ws == rank:
            join_parts(species, args, ws)
            write_snps_readme(args, species)
            species.write_sample_info(dtype='snps', outdir=args['outdir'])
    dist.barrier()
    dist.finalize()
