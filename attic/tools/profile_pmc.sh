#!/bin/bash
# GPU box: arbitrary PMC passes.  usage: tools/profile_pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu"
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_p$i -o pmc -- $BENCH > $OUT/pmc_p$i.log 2>&1
done
cd $REPO && python tools/summarize_prof.py $OUT | grep -v "^JSON" | grep pileup
