#!/bin/bash
# first GPU run of the fused direct path: parity, timing, probe, two ablations
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_direct.py tests/test_gpu_parity.py -x -q 2>&1 | tail -15 ) > gpurun_out/r04_first_tests.txt
( python tools/direct_check.py c3 20 2>&1 | tail -8 ) > gpurun_out/r04_first_c3.txt
( MIDAS_SNPS_LIBRARY=$PWD/midas_amd/lib/libmidas_snps_hip_probe.so python tools/probe_direct.py c3 2>&1 | tail -12 ) > gpurun_out/r04_first_probe.txt
for D in 4 1; do ( echo "debug bits $D"; DIRECT_CHECK_PATHS=1 MIDAS_SNPS_LIBRARY=$PWD/midas_amd/lib/libmidas_snps_hip_dbg$D.so python tools/direct_check.py c3 20 2>&1 | tail -1 ) >> gpurun_out/r04_first_ablate.txt; done
( python tools/direct_check.py c2 20 2>&1 | tail -4 ) > gpurun_out/r04_first_c2.txt
cat gpurun_out/r04_first_tests.txt gpurun_out/r04_first_c3.txt gpurun_out/r04_first_probe.txt gpurun_out/r04_first_ablate.txt gpurun_out/r04_first_c2.txt
