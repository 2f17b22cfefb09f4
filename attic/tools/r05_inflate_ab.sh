#!/bin/bash
# GPU box (round 5): the device decode of a configs[2] BAM under rocprofv3, the product against variants of the inflater's shape
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=$PWD/gpurun_out/r05_inflate_ab.txt
L=$PWD/midas_amd/lib/libmidas_snps_hip
W=/tmp/midas_decode_c3
DECODE_ON_DEVICE=2 timeout 900 python tools/decode_trace.py c3 $W > $O 2>&1
timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q 2>&1 | tail -2 >> $O
cd /tmp && export TMPDIR=/tmp
for V in product "$@"; do
  LIB=${L}.so; [ $V != product ] && LIB=${L}_$V.so
  P=/tmp/prof_inf_$V; rm -rf $P
  ( echo "== $V"; MIDAS_SNPS_LIBRARY=$LIB DECODE_ON_DEVICE=2 timeout 600 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- python $GRAFT_REPO_ROOT/tools/decode_trace.py c3 $W 2>&1 | grep "^run" ) >> $O
  ( cd $GRAFT_REPO_ROOT && python tools/summarize_prof.py $P | grep -v "^JSON" | grep -i "inflate\|resolve\|crc\|walk\|offsets\|columns\|payload\|scan\|copy" | cut -c1-150 ) >> $O
done
cat $O
