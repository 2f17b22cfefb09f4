#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats only (no PMC passes) of the bench command.  usage: tools/profile_trace.sh <tag> [bench args...]
set -u
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu "$@" > $OUT/trace.log 2>&1
cd $REPO && python tools/summarize_prof.py $OUT | grep -v "^JSON"
