#!/bin/bash
# GPU box: A/B the product library against a variant on the same box, alternating.  usage: tools/ab.sh <suffix> [rounds]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
SUF=$1; N=${2:-3}
run() { python $REPO/bench.py --steps 40 --warmup 5 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s kernel %.2f us  step %.2f us' % (sys.argv[1], d['roofline']['kernel_ms_avg']*1e3, d['ms_per_step']*1e3))" $1; }
for i in $(seq $N); do
  run A
  MIDAS_SNPS_LIBRARY=$REPO/midas_amd/lib/libmidas_snps_hip_$SUF.so run B
done
