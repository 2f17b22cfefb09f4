#!/bin/bash
# GPU box: A/B one library under two settings of an environment variable, alternating on the same box.
# usage: tools/ab_env.sh VAR=valueB [rounds] [bench args...]      (A = variable unset)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
KV=$1; N=${2:-3}; shift; shift
run() { python $REPO/bench.py --steps 40 --warmup 5 --no-cpu "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s kernel %.2f us  step %.2f us  frac %.4f' % (sys.argv[1], d['roofline']['kernel_ms_avg']*1e3, d['ms_per_step']*1e3, d['roofline']['frac']))" $LABEL; }
for i in $(seq $N); do
  LABEL=A run "$@"
  LABEL=B env $KV bash -c "$(declare -f run); LABEL=B REPO=$REPO run $*"
done
