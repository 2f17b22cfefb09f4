#!/bin/bash
# GPU box: pileup kernel time on configs[1] with parts of the kernel switched off (MIDAS_SNPS_DEBUG bits:
# 1 = no LDS tallies, 2 = no tile write-out, 4 = no per-base work at all, 8 = write-out lands on the first tile only,
# i.e. stays in L2).  Results are WRONG for bits != 0.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for D in ${*:-0 1 4 2 6 7}; do
  echo -n "debug=$D grid=${GRID:-512} "
  MIDAS_SNPS_GRID=${GRID:-512} MIDAS_SNPS_DEBUG=$D python $REPO/bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; print('ms_per_step %.4f  kernel_us %.1f  frac %.3f' % (d['ms_per_step'], 500.5e6/ (r['achieved']*1e9)*1e6 if r['achieved'] else 0, r['frac']))"
done
