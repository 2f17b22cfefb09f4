"""Developer script (GPU box): HIP path vs the C oracle on a few seeded datasets, plus kernel timings.
Usage: python tools/gpu_check.py [config ...]     (configs from midas_amd.synth.CONFIGS, default: tiny c2)
"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from midas_amd import abi, synth  # noqa: E402
from oracle import c_oracle  # noqa: E402


def check(name, contigs, reads, ctx, thr, steps=5):
    t0 = time.time()
    st, er, oc, oa, os_ = c_oracle.pileup(thr, contigs, reads)
    t_cpu = time.time() - t0
    b = ctx.batch(contigs, reads)
    b.enable_timing(1)
    b.run(thr)
    counts, allele, stats = b.fetch()
    ok_c = np.array_equal(counts, oc)
    ok_a = np.array_equal(allele, oa)
    ok_s = np.array_equal(stats, os_)
    info = b.info()
    ts = []
    for _ in range(steps):
        b.run(thr)
        b.sync()
        ts.append(b.last_timing())
    best = min(t['pileup_ms'] for t in ts)
    run = min(t['run_ms'] for t in ts)
    idx = min(t['index_ms'] for t in ts)
    alg = info.algorithmic_bytes
    print("%-8s sites=%d reads=%d tiles=%d lpr=%d | counts %s allele %s stats %s | cpu %.3fs | index %.3f ms pileup %.3f ms run %.3f ms"
          " | %.1f GB/s alg (kernel) %.2e sites/s (run)" % (
              name, info.n_sites, info.n_reads, info.n_tiles, info.lanes_per_read, ok_c, ok_a, ok_s, t_cpu,
              idx, best, run, alg / best / 1e6, info.n_sites / run * 1e3), flush=True)
    if not ok_c:
        bad = np.nonzero((counts != oc).any(axis=1))[0]
        print("  first mismatching sites:", bad[:10], "of", bad.size)
        for s in bad[:5]:
            print("   site", s, "hip", counts[s], "oracle", oc[s])
    if not ok_s:
        print("  stats hip", stats.tolist(), "oracle", os_.tolist())
    b.close()
    return ok_c and ok_a and ok_s


def main():
    names = sys.argv[1:] or ['tiny', 'c2']
    ctx = abi.Context(0)
    print(ctx.device_info())
    thr = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
    ok = True
    for nm in names:
        t0 = time.time()
        if nm == 'ragged':
            contigs, reads = synth.make_dataset(n_species=3, contigs_per_species=5, contig_len=30011, n_reads=40000,
                                                seed=7, var_len=True, lowercase_frac=0.1)
        elif nm == 'deep':
            contigs, reads = synth.make_dataset(n_species=1, contigs_per_species=2, contig_len=9000, n_reads=60000, seed=9)
        else:
            contigs, reads = synth.make_dataset(**synth.CONFIGS[nm])
        print("generated %s in %.1fs" % (nm, time.time() - t0), flush=True)
        ok &= check(nm, contigs, reads, ctx, thr)
    print("ALL OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
