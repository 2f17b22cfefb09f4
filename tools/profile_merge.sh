#!/bin/bash
# GPU box: rocprofv3 kernel trace + HBM-traffic PMC pass of midas_merge_sites (tools/merge_check.py).
# usage: tools/profile_merge.sh <tag> [n_sites] [n_samples]
set -u
TAG=$1; N=${2:-15000000}; S=${3:-8}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/merge_check.py $N $S 5"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_p1 -o pmc -- $CMD > $OUT/pmc_p1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_p2 -o pmc -- $CMD > $OUT/pmc_p2.log 2>&1
cd $REPO
{ echo "# $CMD"; grep RESULT $OUT/trace.log; python tools/summarize_prof.py $OUT | grep -v "^JSON"; } > $OUT/summary.txt
cat $OUT/summary.txt
rm -rf $OUT/trace $OUT/pmc_p1 $OUT/pmc_p2      # the rocpd databases are tens of MiB each; gpurun_out/ carries 64 MiB back
