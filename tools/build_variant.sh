#!/bin/bash
# Build a developer variant of the library next to the product one: tools/build_variant.sh <suffix> <extra hipcc flags...>
# Use it with MIDAS_SNPS_LIBRARY=midas_amd/lib/libmidas_snps_hip_<suffix>.so (e.g. -DMIDAS_TILE_SHIFT=11 -DMIDAS_PILEUP_BLOCK=256,
# -DMIDAS_SNPS_DEBUG_BITS=bits for the ablation switches of the pileup kernels).
set -e
SUF=$1; shift
cd "$(dirname "$0")/.."
python -m midas_amd.build -o midas_amd/lib/libmidas_snps_hip_$SUF.so "$@" | tail -1
