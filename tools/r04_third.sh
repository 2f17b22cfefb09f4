#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04_third.txt
( timeout 900 python -m pytest tests/test_gpu_direct.py -x -q 2>&1 | tail -3 ) > $O
( DIRECT_CHECK_PATHS=1 python tools/direct_check.py c3 20 2>&1 | tail -1 ) >> $O
for V in b384d2 b384d1 b512d2; do
  ( echo "variant $V"; MIDAS_SNPS_LIBRARY=$PWD/midas_amd/lib/libmidas_snps_hip_$V.so timeout 900 python -m pytest tests/test_gpu_direct.py -x -q 2>&1 | tail -2;
    DIRECT_CHECK_PATHS=1,2 MIDAS_SNPS_LIBRARY=$PWD/midas_amd/lib/libmidas_snps_hip_$V.so python tools/direct_check.py c3 20 2>&1 | tail -3 ) >> $O
done
cat $O
