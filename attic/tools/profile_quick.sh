#!/bin/bash
# GPU box: kernel trace + two SQ counter passes (quick look while iterating).  usage: tools/profile_quick.sh <tag>
set -u
TAG=${1:-q}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$N -o pmc -- $BENCH > $OUT/pmc_$N.log 2>&1
done
cd $REPO && python tools/summarize_prof.py $OUT | grep -v "^JSON" > $OUT/summary.txt; cat $OUT/summary.txt
