"""Developer timing of midas_genes_count on the GPU box: python tools/genes_check.py [n_reads] [genes_per_species] [n_species]
Prints wall time of the call (host derivation + sort + H2D + kernel + D2H) and the kernel's own time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midas_amd import abi, synth  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
gps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
nsp = int(sys.argv[3]) if len(sys.argv) > 3 else 10
t0 = time.time()
ds = synth.make_pangenome_dataset(n_species=nsp, genes_per_species=gps, n_reads=n_reads, seed=77)
lengths = [len(s) for s in ds['gene_seq']]
print("dataset: %d genes, %d reads (%.1f s to make)" % (len(lengths), ds['reads'].n_reads, time.time() - t0))
thr = abi.Thresholds.from_args(dict(abi.DEFAULT_ARGS, mapq=0))
with abi.Context(0) as ctx:
    for rep in range(4):
        t0 = time.time()
        aligned, mapped, depth, ms = ctx.genes_count(thr, ds['reads'], ds['refid'], lengths)
        wall = time.time() - t0
        print("call %d: wall %.1f ms, kernel %.3f ms, %.1f M reads/s whole call; mapped %d of %d" %
              (rep, wall * 1e3, ms, ds['reads'].n_reads / wall / 1e6, int(mapped.sum()), int(aligned.sum())))
    import numpy as np
    hot = np.where(np.arange(ds['refid'].size) % 3 == 0, 5, ds['refid']).astype(np.int32)     # a third of the reads on one gene
    for rep in range(2):
        t0 = time.time()
        aligned, mapped, depth, ms = ctx.genes_count(thr, ds['reads'], hot, lengths)
        print("hot gene (%d reads): wall %.1f ms, kernel %.3f ms" % (int(aligned[5]), (time.time() - t0) * 1e3, ms))
