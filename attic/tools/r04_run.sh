#!/bin/bash
# GPU box: direct-path tests, timing on configs[2] and configs[1], dynamic instruction counts
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r04_run.txt
( timeout 900 python -m pytest tests/test_gpu_direct.py -x -q 2>&1 | tail -8 ) > $O
( DIRECT_CHECK_PATHS=1,2 timeout 600 python tools/direct_check.py c3 20 2>&1 | tail -3 ) >> $O
( DIRECT_CHECK_PATHS=1 timeout 600 python tools/direct_check.py c2 20 2>&1 | tail -1 ) >> $O
bash tools/r04_pmc_insts.sh >> $O 2>&1
cat $O
