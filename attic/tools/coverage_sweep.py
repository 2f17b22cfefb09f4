"""Developer script (GPU box): pileup kernel time vs coverage on the configs[1] genome (15 Mb), 150 bp reads."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from midas_amd import abi, synth

def main():
    covs = [float(x) for x in sys.argv[1:]] or [5, 7, 10, 14, 20]
    ctx = abi.Context(0)
    thr = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
    base = dict(synth.CONFIGS['c2'])
    for cov in covs:
        cfg = dict(base)
        cfg['n_reads'] = int(round(base['n_reads'] * cov / 10.0))
        if os.environ.get('PLAIN_ONLY'):
            cfg['plain_only'] = True
        contigs, reads = synth.make_dataset(**cfg)
        b = ctx.batch(contigs, reads)
        b.enable_timing(1)
        ts = []
        for _ in range(8):
            b.run(thr); b.sync(); ts.append(b.last_timing())
        info = b.info()
        k = min(t['pileup_ms'] for t in ts); r = min(t['run_ms'] for t in ts)
        print("coverage %5.1fx reads %8d | pileup %.3f ms run %.3f ms | %.1f GB/s alg = %.1f%% of 8 TB/s | %.2e sites/s" % (
            cov, info.n_reads, k, r, info.algorithmic_bytes / k / 1e6, info.algorithmic_bytes / k / 1e6 / 80.0, info.n_sites / r * 1e3), flush=True)
        b.close()

main()
