#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + PMC passes of the direct path (tools/direct_check.py, direct path only).
# usage: tools/profile_direct.sh <tag> [config] [quick]
set -u
TAG=${1:-d}; CFG=${2:-c3}; QUICK=${3:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DIRECT_CHECK_PATHS=1
CMD="python $REPO/tools/direct_check.py $CFG 10"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
if [ "$QUICK" = "trace" ]; then
  LIST=()
elif [ "$QUICK" = "quick" ]; then
  LIST=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM")
else
  LIST=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum")
fi
for C in "${LIST[@]}"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$N -o pmc -- $CMD > $OUT/pmc_$N.log 2>&1
done
cd $REPO && python tools/summarize_prof.py $OUT | grep -v "^JSON" > $OUT/summary.txt; cat $OUT/summary.txt
