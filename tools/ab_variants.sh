#!/bin/bash
# GPU box : parity of the product kernel, then product against named variants on ONE box, alternating.
#   tools/ab_variants.sh "<pytest targets or ->" <variant> [<variant> ...]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/ab_variants_$(date +%H%M%S).txt
L=$PWD/midas_amd/lib/libmidas_snps_hip
T=$1; shift
if [ "$T" != "-" ]; then ( timeout 900 python -m pytest $T -x -q 2>&1 | tail -3 ) > $O; fi
( timeout 900 python tools/direct_check.py c3 20 2>&1 | tail -5 ) >> $O
for V in "$@"; do      # every variant against the packed path on the same data: a variant that is not exact is not a result
  ( echo "== variant $V, direct == packed?"; MIDAS_SNPS_LIBRARY=${L}_$V.so timeout 900 python tools/direct_check.py c3 5 2>&1 | tail -1 ) >> $O
done
for rep in 1 2; do
  for V in "$@"; do
    ( echo "== variant $V"; MIDAS_SNPS_LIBRARY=${L}_$V.so timeout 300 python tools/direct_time.py c3 2>&1 | tail -1 ) >> $O
  done
  ( echo "== product"; timeout 300 python tools/direct_time.py c3 2>&1 | tail -1 ) >> $O
done
cat $O
