#!/bin/bash
# GPU box: a pileup-kernel variant against the shipped kernel on ONE box: time, tallies off, loads only, probes.  Build first:
#   python -m midas_amd.build -o midas_amd/lib/libmidas_snps_hip_old.so --replace pileup_direct.hip=<the kernel to compare with>
#   (+ _probe / _oldprobe with -DMIDAS_SNPS_DEBUG_BITS=256, _dbg1 / _dbg4 with =1 / =4)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04_fourth.txt
L=$PWD/midas_amd/lib/libmidas_snps_hip
( DIRECT_CHECK_PATHS=1 python tools/direct_check.py c3 20 2>&1 | tail -1 ) > $O
for V in old dbg1 dbg4; do
  ( echo "== variant $V"; MIDAS_SNPS_LIBRARY=${L}_$V.so python tools/direct_time.py c3 2>&1 | tail -1 ) >> $O
done
( echo "== product again"; python tools/direct_time.py c3 2>&1 | tail -1 ) >> $O
for V in probe oldprobe; do
  ( echo "== variant $V"; MIDAS_SNPS_LIBRARY=${L}_$V.so python tools/probe_direct.py c3 2>&1 | head -12 ) >> $O
done
cat $O
