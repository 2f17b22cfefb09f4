"""Developer script (GPU box): the whole-file device decode of a configs[2] BAM, grouped (upload || inflate) against not,
alternating on one box.  usage: python tools/decode_ab.py [config] [workdir]"""
import os
import shutil
import sys
import time

sys.path.insert(0, '.')
from midas_amd import abi, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
work = sys.argv[2] if len(sys.argv) > 2 else '/tmp/midas_ab'
shutil.rmtree(work, ignore_errors=True)
contigs, reads = synth.make_dataset(**synth.CONFIGS[cfg])
out, db = os.path.join(work, 'sample'), os.path.join(work, 'db')
synth.write_sample(out, db, contigs, reads)
path = os.path.join(out, 'snps/temp/genomes.bam')
print("BAM %.0f MB" % (os.path.getsize(path) / 1e6), flush=True)
ctx = abi.Context(0)
abi.read_bam(path, ctx, payload_on_device=True)          # (arena, pinned ring, streams: once)
for rep in range(4):
    for name, val in (("grouped", None), ("one launch", str(1 << 60))):
        if val is None:
            os.environ.pop("MIDAS_SNPS_DECODE_GROUP_MIN", None)
        else:
            os.environ["MIDAS_SNPS_DECODE_GROUP_MIN"] = val
        t = time.perf_counter()
        d = abi.read_bam(path, ctx, payload_on_device=True)
        dt = time.perf_counter() - t
        del d
        print("%-11s %.1f ms" % (name, dt * 1e3), flush=True)
shutil.rmtree(work, ignore_errors=True)
