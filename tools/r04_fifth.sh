#!/bin/bash
# GPU box: direct tests + time of the product library against the kernel before it, same box
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04_fifth.txt
L=$PWD/midas_amd/lib/libmidas_snps_hip
( timeout 900 python -m pytest tests/test_gpu_direct.py -x -q 2>&1 | tail -3 ) > $O
( DIRECT_CHECK_PATHS=1 python tools/direct_check.py c3 20 2>&1 | tail -1 ) >> $O
for V in old $EXTRA_VARIANTS; do
  ( echo "== variant $V"; MIDAS_SNPS_LIBRARY=${L}_$V.so python tools/direct_time.py c3 2>&1 | tail -1 ) >> $O
done
( echo "== product again"; python tools/direct_time.py c3 2>&1 | tail -1 ) >> $O
cat $O
