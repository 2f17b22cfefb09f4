"""Developer script (GPU box): a batch of 250 bp reads (configs[2]'s genomes and depth) on the direct path -- the kernel's
instantiation with the long overhang (288 sites, three workgroups a CU, chunks of four tiles, a read visited once) against the
common one without chunks (MIDAS_SNPS_OVERHANG_MAX=160: every tile by itself, four workgroups a CU) -- each in a process of its own,
both held to each other's counts.   usage: python tools/overhang_ab.py [read_len=250] [config=c3]"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
CHILD = r'''
import sys, zlib
sys.path.insert(0, %(root)r)
import numpy as np
from midas_amd import abi, synth
cfg = dict(synth.CONFIGS[sys.argv[2]])
l = int(sys.argv[1])
cfg["n_reads"] = int(cfg["n_reads"] * cfg.get("read_len", 150) / l)
cfg["read_len"] = l
contigs, reads = synth.make_dataset(**cfg)
ctx = abi.Context(0)
thr = abi.Thresholds.from_args(abi.DEFAULT_ARGS)
b = ctx.batch(contigs, reads)
b.select_path(abi.PATH_DIRECT)
info = b.info()
for rep in range(2):
    b.enable_timing(20)
    for _ in range(20):
        b.run(thr)
    b.sync()
tm = [b.timing(i) for i in range(20)]
counts, allele, stats = b.fetch()
print("chunk tiles %%d, overhang %%d, lanes per read %%d: ranges %%.4f ms + pileup %%.4f ms; %%d reads, A = %%.3f GB -> %%.3f of 8 TB/s; counts crc %%08x, stats %%s" %% (
    info.direct_chunk_tiles, info.direct_overhang, info.lanes_per_read, np.mean([t['index_ms'] for t in tm]), np.mean([t['pileup_ms'] for t in tm]),
    info.n_reads, info.algorithmic_bytes / 1e9, info.algorithmic_bytes / (np.mean([t['index_ms'] + t['pileup_ms'] for t in tm]) * 1e-3) / 8e12,
    zlib.crc32(counts.tobytes()) & 0xffffffff, stats.sum(axis=0).tolist()), flush=True)
'''
l = sys.argv[1] if len(sys.argv) > 1 else '250'
cfg = sys.argv[2] if len(sys.argv) > 2 else 'c3'
for name, env in (("long overhang, chunks", {}), ("common overhang, no chunks", {"MIDAS_SNPS_OVERHANG_MAX": "160"}), ("long overhang, chunks", {}), ("common overhang, no chunks", {"MIDAS_SNPS_OVERHANG_MAX": "160"})):
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": os.path.abspath(ROOT)}, l, cfg], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env))
    print("%-28s %s" % (name, r.stdout.strip() or r.stderr[-400:]), flush=True)
