#!/bin/bash
# GPU box (round 5): parity of the product kernel, then product against named variants on ONE box, alternating.
#   tools/r05_ab.sh "<pytest targets or ->" <variant> [<variant> ...]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r05_ab_$(date +%H%M%S).txt
L=$PWD/midas_amd/lib/libmidas_snps_hip
T=$1; shift
if [ "$T" != "-" ]; then ( timeout 900 python -m pytest $T -x -q 2>&1 | tail -3 ) > $O; fi
( DIRECT_CHECK_PATHS=1 timeout 600 python tools/direct_check.py c3 20 2>&1 | tail -2 ) >> $O
for rep in 1 2; do
  for V in "$@"; do
    ( echo "== variant $V"; MIDAS_SNPS_LIBRARY=${L}_$V.so timeout 300 python tools/direct_time.py c3 2>&1 | tail -1 ) >> $O
  done
  ( echo "== product"; timeout 300 python tools/direct_time.py c3 2>&1 | tail -1 ) >> $O
done
cat $O
