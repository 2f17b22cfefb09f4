#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04_second.txt
( timeout 900 python -m pytest tests/test_gpu_direct.py -x -q 2>&1 | tail -5 ) > $O
( DIRECT_CHECK_PATHS=1 python tools/direct_check.py c3 20 2>&1 | tail -2 ) >> $O
for D in 16 32 64; do ( echo "debug bits $D"; DIRECT_CHECK_PATHS=1 MIDAS_SNPS_LIBRARY=$PWD/midas_amd/lib/libmidas_snps_hip_dbg$D.so python tools/direct_check.py c3 20 2>&1 | tail -1 ) >> $O; done
cat $O
