#!/bin/bash
# GPU box: workgroup / tile shapes of the pileup kernels against the shipped ones, on one box (variants built with tools/build_variant.sh)
cd ${GRAFT_REPO_ROOT:-.}
L=$PWD/midas_amd/lib/libmidas_snps_hip
for V in "$@"; do
  echo "== $V"; MIDAS_SNPS_LIBRARY=${L}_$V.so timeout 300 python -m pytest tests/test_gpu_direct.py -x -q 2>&1 | tail -1
  DIRECT_CHECK_PATHS=1,2 MIDAS_SNPS_LIBRARY=${L}_$V.so python tools/direct_check.py c3 20 2>&1 | tail -3
  DIRECT_CHECK_PATHS=1 MIDAS_SNPS_LIBRARY=${L}_$V.so python tools/direct_check.py c2 20 2>&1 | tail -1
done
echo "== shipped"; DIRECT_CHECK_PATHS=1,2 python tools/direct_check.py c3 20 2>&1 | tail -3; DIRECT_CHECK_PATHS=1 python tools/direct_check.py c2 20 2>&1 | tail -1
