"""Developer script (GPU box): midas_merge_sites on a configs[1]-sized species (15 M sites) x S samples: kernel time,
algorithmic GB/s, and a spot check of 2000 sites against the oracle.
Usage: python tools/merge_check.py [n_sites] [n_samples] [reps]
"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from midas_amd import abi  # noqa: E402
from oracle import merge_oracle as mo  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 15_000_000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rng = np.random.default_rng(0)
    t0 = time.time()
    ref = rng.integers(0, 4, n)
    alt = (ref + rng.integers(1, 4, n)) % 4
    snp = rng.random(n) < 0.03
    counts = []
    for s in range(S):
        depth = rng.poisson(10.0, n).astype(np.uint32)
        na = np.where(snp, rng.binomial(depth, 0.3), 0).astype(np.uint32)
        c = np.zeros((n, 4), np.uint32)
        c[np.arange(n), ref] = depth - na
        c[np.arange(n), alt] += na
        counts.append(c)
    mean = [10.0] * S
    print("generated %d sites x %d samples in %.1fs" % (n, S, time.time() - t0), flush=True)
    ctx = abi.Context(0)
    args = dict(abi.DEFAULT_MERGE_ARGS)
    prm = abi.MergeParams.from_args(args)
    alg = n * (24 * S + 40)      # 16 B read + 8 B written per (site, sample), 40 B of per-site outputs
    best = 1e9
    for r in range(reps):
        t0 = time.time()
        res = ctx.merge_sites(prm, counts, mean)
        wall = time.time() - t0
        best = min(best, res['kernel_ms'])
        print("rep %d: kernel %.3f ms  (%.1f GB/s algorithmic, %.2e sites/s)  call wall %.2fs" % (
            r, res['kernel_ms'], alg / res['kernel_ms'] / 1e6, n / res['kernel_ms'] * 1e3, wall), flush=True)
    sel = rng.integers(0, n, 2000)
    bad = 0
    for i in sel:
        c = [[int(x) for x in counts[s][i]] for s in range(S)]
        pooled = mo.pooled_counts(c)
        major, minor, st = mo.call_alleles(pooled, args['allele_freq'])
        mafs, depths = mo.per_sample(c, major, minor)
        cs, prev = mo.prevalence(mean, depths, args['site_depth'], args['site_ratio'])
        why = mo.flag_reason(prev, st, args['site_prev'], args['snp_type'])
        ok = (res['major'][i] == (255 if major is None else major) and res['minor'][i] == (255 if minor is None else minor)
              and res['snp_type'][i] == [None, 'mono', 'bi', 'tri', 'quad'].index(st) and res['count_samples'][i] == cs
              and res['flag'][i] == {None: 0, 'min_prev': 1, 'snp_type': 2}[why] and list(res['depth'][:, i]) == depths
              and list(res['pooled'][i]) == pooled)
        bad += not ok
    print("spot check vs oracle: %d / %d sites differ; kept %d sites" % (bad, len(sel), int((res['flag'] == 0).sum())))
    print("RESULT merge_sites n_sites=%d n_samples=%d kernel_ms=%.3f alg_bytes=%d alg_GBps=%.1f frac_of_8TBps=%.3f" % (
        n, S, best, alg, alg / best / 1e6, alg / best / 1e6 / 8000.0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
