#!/bin/bash
# GPU box: the batched pileup kernel against the one before it (tools/variants/pileup_direct_r04a.hip), same box: time, probes
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
