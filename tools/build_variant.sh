#!/bin/bash
# Build a developer variant of the library next to the product one: tools/build_variant.sh <suffix> <extra hipcc flags...>
# Use it with MIDAS_SNPS_LIBRARY=midas_amd/lib/libmidas_snps_hip_<suffix>.so (e.g. -DMIDAS_TILE_SHIFT=11 -DMIDAS_PILEUP_BLOCK=256,
# -DMIDAS_SNPS_STREAM_KERNEL [-DMIDAS_STREAM_DRAIN_WAVES=n -DMIDAS_STREAM_ABLATE=bits] for the barrier-free pileup kernel,
# -DMIDAS_SNPS_DEBUG_BITS=bits for the ablation switches of the barrier-phased one).
set -e
SUF=$1; shift
cd "$(dirname "$0")/.."
C=midas_amd/csrc
EXTRA=""
case " $* " in *MIDAS_SNPS_STREAM_KERNEL*) EXTRA="tools/variants/pileup_stream.hip -I$C";; esac   # the barrier-free kernel lives outside the product tree
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-atomic-optimizer-strategy=None -x hip "$@" $C/pack.cpp $C/hostio.cpp $C/row_deflate.cpp $C/pack_reads.hip $C/index_reads.hip \
  $C/pileup_tiles.hip $C/index_direct.hip $C/pileup_direct.hip $C/rows_deflate.hip $C/bgzf_inflate.hip $EXTRA $C/merge_sites.hip $C/genes_count.hip $C/snps_abi.hip -o midas_amd/lib/libmidas_snps_hip_$SUF.so -lz -lpthread -ldl
echo built midas_amd/lib/libmidas_snps_hip_$SUF.so
