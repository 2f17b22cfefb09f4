"""Developer script (GPU box): the resident device decode of a configs[k] BAM alone -- one arena against the streamed decode at
several group sizes / slot counts (MIDAS_SNPS_DECODE_*), best of a few calls each, the refIDs compared between variants.
usage: python tools/decode_probe.py [config=c3] [workdir]"""
import os
import shutil
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from midas_amd import abi, bam, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
work = sys.argv[2] if len(sys.argv) > 2 else '/tmp/midas_decode_probe'
shutil.rmtree(work, ignore_errors=True)
os.makedirs(work)
if cfg == 'c4':
    contigs, reads, _ = synth.c4_share(0, 1)
else:
    contigs, reads = synth.make_dataset(**synth.CONFIGS[cfg])
path = os.path.join(work, 'genomes.bam')
refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
bam.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads)
print("%s: %d reads, BAM %.2f GB" % (cfg, reads.n_reads, os.path.getsize(path) / 1e9), flush=True)
del reads
KEYS = ("MIDAS_SNPS_DECODE_STREAM", "MIDAS_SNPS_DECODE_GROUP_BLOCKS", "MIDAS_SNPS_DECODE_SLOTS")
variants = [("one arena", {"MIDAS_SNPS_DECODE_STREAM": "0"}), ("streamed, the default groups", {})]
for blocks in (() if os.environ.get("DECODE_PROBE_BRIEF") else (40960, 24000, 16000, 12000, 8000)):      # (DECODE_PROBE_BRIEF=1: the two decodes only, e.g. under rocprofv3)
    for slots in (2, 3):
        variants.append(("streamed, groups of %d blocks, %d slots" % (blocks, slots), {"MIDAS_SNPS_DECODE_GROUP_BLOCKS": str(blocks), "MIDAS_SNPS_DECODE_SLOTS": str(slots)}))
with abi.Context(0) as ctx:
    abi.read_bam(path, ctx, resident=True)       # (the first call of a process: the pinned ring, the kernels' code objects)
    first = None
    for name, env in variants:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        best = 1e9
        for _ in range(2 if os.environ.get("DECODE_PROBE_BRIEF") else 4):
            t = time.perf_counter()
            _, _, rid, res = abi.read_bam(path, ctx, resident=True)
            best = min(best, time.perf_counter() - t)
            if first is None:
                first = np.array(rid)
            else:
                assert np.array_equal(first, rid)
            del res, rid
        print("%-52s %7.1f ms (best of 4 whole calls: map, block table, decode, refID down)" % (name, best * 1e3), flush=True)
