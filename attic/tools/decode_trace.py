"""Developer script: where midas_bam_load spends its time on a configs[k] BAM (library built with -DMIDAS_HOSTIO_TRACE:
tools/build_variant.sh trace -DMIDAS_HOSTIO_TRACE; MIDAS_SNPS_LIBRARY=midas_amd/lib/libmidas_snps_hip_trace.so).
DECODE_ON_DEVICE=1: the BGZF blocks are inflated on the device; =2: and SEQ / QUAL / CIGAR left there (midas_bam_load_device);
LOCAL_WORLD_SIZE=8 gives the host code the CPU budget of one rank of eight.   usage: python tools/decode_trace.py [config] [workdir]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from midas_amd import abi, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
work = sys.argv[2] if len(sys.argv) > 2 else '/tmp/midas_decode'
bam = os.path.join(work, 'sample/snps/temp/genomes.bam')
if not os.path.exists(bam):
    contigs, reads = synth.make_dataset(**synth.CONFIGS[cfg])
    synth.write_sample(os.path.join(work, 'sample'), os.path.join(work, 'db'), contigs, reads)
ctx = abi.Context(0) if os.environ.get("DECODE_ON_DEVICE") else None     # DECODE_ON_DEVICE=1: the device inflates the blocks
for k in range(3):
    t = time.perf_counter()
    d = abi.read_bam(bam, ctx, payload_on_device=os.environ.get("DECODE_ON_DEVICE") == "2")
    print("run %d: %.3f s, %d records" % (k + 1, time.perf_counter() - t, d[3].n_reads), flush=True)
    del d
