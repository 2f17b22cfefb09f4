#!/bin/bash
# Pileup kernel time with parts of the kernel switched off.  The switches are compile-time (MIDAS_SNPS_DEBUG_BITS:
# 1 = no LDS tallies, 2 = no tile write-out, 4 = no per-base work at all, 8 = write-out lands on the first tile only,
# i.e. stays in L2, 16 = static tile order, 64 = full __syncthreads between the phases); results are WRONG for bits != 0.
# Step 1 (here, no GPU needed):  tools/ablate.sh build 1 4 2 6 7     -> midas_amd/lib/libmidas_snps_hip_dbg<bits>.so
# Step 2 (GPU box):              tools/ablate.sh run c2 1 4 2 6 7
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MODE=$1; shift
if [ "$MODE" = build ]; then
  for D in "$@"; do bash $REPO/tools/build_variant.sh dbg$D -DMIDAS_SNPS_DEBUG_BITS=$D | tail -1; done
  exit 0
fi
CFG=$1; shift
run() { python $REPO/bench.py --config $CFG --steps 20 --warmup 3 --no-cpu --pack-steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; print('$1 ms_per_step %.4f  kernel_us %.1f  frac %.3f' % (d['ms_per_step'], r['kernel_ms_avg']*1e3, r['frac']))"; }
run product
for D in "$@"; do MIDAS_SNPS_LIBRARY=$REPO/midas_amd/lib/libmidas_snps_hip_dbg$D.so run debug=$D; done
