#!/bin/bash
# Direct pileup kernel with parts switched off (compile-time MIDAS_SNPS_DEBUG_BITS: 1 = no tallies, 2 = no tile write-out,
# 4 = loads only, 8 = write-out lands on the first tile (stays in L2), 128 = records only, no base loads); results are WRONG
# for bits != 0 -- timing only.
#   here (no GPU):  tools/ablate_direct.sh build 1 2 3 4 ...   ->  midas_amd/lib/libmidas_snps_hip_dbg<bits>.so
#   GPU box:        tools/ablate_direct.sh run c3 1 2 3 4 ...
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MODE=$1; shift
if [ "$MODE" = build ]; then
  for D in "$@"; do bash $REPO/tools/build_variant.sh dbg$D -DMIDAS_SNPS_DEBUG_BITS=$D | tail -1; done
  exit 0
fi
CFG=$1; shift
export DIRECT_CHECK_PATHS=1
echo "product:"; python $REPO/tools/direct_check.py $CFG 20 2>&1 | tail -1
for D in "$@"; do echo "debug bits $D:"; MIDAS_SNPS_LIBRARY=$REPO/midas_amd/lib/libmidas_snps_hip_dbg$D.so python $REPO/tools/direct_check.py $CFG 20 2>&1 | tail -1; done
