"""Developer script (GPU box): `merge_midas.py snps` end to end on one configs[1]-sized species x S samples,
phase by phase (read tables, device arithmetic, annotation + text).  Usage: python tools/merge_e2e.py [n_sites] [S] [preset]
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midas_amd import abi, synth  # noqa: E402
from midas_amd.merge import merge, snps as msnps  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 15_000_000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    preset = sys.argv[3] if len(sys.argv) > 3 else 'core_snps'
    root = tempfile.mkdtemp(prefix="merge_e2e_")
    rng = np.random.default_rng(0)
    t0 = time.time()
    n_contigs = 60
    lens = [n // n_contigs] * n_contigs
    lens[-1] += n - sum(lens)
    ids = sorted("sp1_c%03d" % k for k in range(n_contigs))
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref_idx = rng.integers(0, 4, n)
    ref = letters[ref_idx]
    db = os.path.join(root, "db")
    for d in ("marker_genes", "pan_genomes", "rep_genomes/sp1"):
        os.makedirs(os.path.join(db, d), exist_ok=True)
    synth.write_db_tables(db, ["sp1"])
    off = np.concatenate([[0], np.cumsum(lens)])
    with open(os.path.join(db, "rep_genomes/sp1/genome.fna"), "w") as h:
        for k, cid in enumerate(ids):
            h.write(">%s\n%s\n" % (cid, ref[off[k]:off[k + 1]].tobytes().decode()))
    synth.write_features(db, "sp1", synth.make_genes(rng, ids, lens))
    alt = (ref_idx + rng.integers(1, 4, n)) % 4
    snp = rng.random(n) < 0.03
    samples = []
    for s in range(S):
        depth = rng.poisson(10.0, n).astype(np.uint32)
        na = np.where(snp, rng.binomial(depth, 0.3), 0).astype(np.uint32)
        c = np.zeros((n, 4), np.uint32)
        c[np.arange(n), ref_idx] = depth - na
        c[np.arange(n), alt] += na
        sdir = os.path.join(root, "samples", "s%02d" % s)
        os.makedirs(os.path.join(sdir, "snps", "output"), exist_ok=True)
        abi.write_table(os.path.join(sdir, "snps", "output", "sp1.snps.gz"), ids, [ref[off[k]:off[k + 1]] for k in range(n_contigs)],
                        [c[off[k]:off[k + 1]] for k in range(n_contigs)], threads=0)
        tot = c.sum(1)
        cov = int((tot > 0).sum())
        with open(os.path.join(sdir, "snps", "summary.txt"), "w") as h:
            h.write("species_id\tgenome_length\tcovered_bases\tfraction_covered\tmean_coverage\taligned_reads\tmapped_reads\n")
            h.write("sp1\t%d\t%d\t%s\t%s\t%d\t%d\n" % (n, cov, cov / float(n), float(tot.sum()) / cov, 1000, 900))
        samples.append(sdir)
    print("setup (not part of the command): %.1f s; %d sites x %d samples" % (time.time() - t0, n, S), flush=True)

    args = dict(outdir=os.path.join(root, "out"), db=db, indirs=samples, species_id=None, max_samples=None, sample_depth=5.0,
                fract_cov=0.4, min_samples=1, max_species=None, threads=64, max_sites=float('Inf'), **abi.DEFAULT_MERGE_ARGS)
    if preset == 'all_sites':
        args.update(snp_type=['any'], site_prev=0.0)
    os.makedirs(args['outdir'], exist_ok=True)
    T = {}
    t = time.perf_counter(); species = merge.select_species(args, 'snps'); T['select species/samples'] = time.perf_counter() - t
    ctx = abi.Context(0)
    sp = species[0]
    t = time.perf_counter(); counts, keys, key_off = msnps.load_sample_tables(sp, args); T['read %d sample tables (native, threads)' % S] = time.perf_counter() - t
    t = time.perf_counter(); res = ctx.merge_sites(abi.MergeParams.from_args(args), counts, sp.sample_depth); T['device arithmetic incl. H2D/D2H'] = time.perf_counter() - t
    kms = res['kernel_ms']
    from midas_amd.merge import annotate
    outdir = '%s/%s' % (args['outdir'], sp.id)
    os.makedirs(outdir, exist_ok=True)
    t = time.perf_counter(); genes = annotate.GeneCursor.from_db(sp.id, args['db']); T['gene table + genome (GeneCursor.from_db)'] = time.perf_counter() - t
    t = time.perf_counter(); keep = np.nonzero(res['flag'] == 0)[0]; T['kept sites'] = time.perf_counter() - t
    hdr = '\t'.join(['site_id'] + [x.id for x in sp.samples]) + '\n'
    t = time.perf_counter()
    abi.write_merge_matrix(outdir + '/f.txt', hdr, keep, res['depth'], res['minor_count'], threads=64)
    abi.write_merge_matrix(outdir + '/d.txt', hdr, keep, res['depth'], None, threads=64)
    T['snps_freq + snps_depth (native)'] = time.perf_counter() - t
    t = time.perf_counter(); abi.write_merge_info(outdir + '/i.txt', 'x\n', keep, keys, key_off, res, genes.genes, threads=64); T['snps_info (native)'] = time.perf_counter() - t
    calls = []

    def timed(mod, name):
        f = getattr(mod, name)

        def g(*a, **k):
            t0 = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                calls.append((name, t0 - t_start, time.perf_counter() - t0))
        setattr(mod, name, g)
    for name in ('read_snps_counts', 'read_snps_table', 'write_merge_matrix', 'write_merge_info'):
        timed(abi, name)
    timed(annotate.GeneCursor, 'from_db')
    timed(type(ctx), 'merge_sites')
    t_start = t = time.perf_counter(); nn, kept, _ = msnps.merge_species(sp, args, ctx); T['whole merge_species (again: read + device + annotate + write)'] = time.perf_counter() - t
    for name, at, dur in calls:
        print("      inside merge_species: %-22s starts %6.3f s  takes %6.3f s" % (name, at, dur))
    total = T['select species/samples'] + T['whole merge_species (again: read + device + annotate + write)']
    for k, v in T.items():
        print("  %-62s %8.3f s" % (k, v))
    print("  kernel %.3f ms; %d of %d sites written; command total ~ %.2f s -> %.3e sites/s" % (kms, kept, nn, total, nn / total))


if __name__ == "__main__":
    main()
