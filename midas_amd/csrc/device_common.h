// Device-side helpers shared by the gfx950 kernels (index_reads.hip, pileup_tiles.hip).
#pragma once
#include "kernels.h"

namespace midas {
namespace dev {

enum : uint32_t { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };

enum : uint32_t {
  E_NO_SEQ = 1,
  E_NO_NM = 2,
  E_ZERO_ALIGN = 3,
  E_NO_QUAL = 4,
  E_CIGAR_OVERRUN = 5,
};

typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));

__device__ __forceinline__ bool consumes_both(uint32_t op) { return op == OP_M || op == OP_EQ || op == OP_X; }

// Unpacked view of a 16-byte ReadRec held as a uint4.
__device__ __forceinline__ int rec_pos(const uint4& r) { return (int)r.x; }
__device__ __forceinline__ uint32_t rec_off8(const uint4& r) { return r.y; }
__device__ __forceinline__ int rec_l(const uint4& r) { return (int)(r.z & 0x7FFu); }
// A kRecSimple record is one match segment of its read; the read-level numbers of the filter ride in the n_cigar / nm
// fields (layout.h): l_seq of the read, its aligned length, NM, and whether this is the read's first segment.
__device__ __forceinline__ int seg_read_l(const uint4& r) { return (int)((r.z >> 16) & 0x3FFu); }
__device__ __forceinline__ int seg_align_len(const uint4& r) { return (int)((((r.z >> 26) & 0xFu) << 6) | ((r.w >> 10) & 0x3Fu)); }
__device__ __forceinline__ int seg_nm(const uint4& r) { return (int)(r.w & 0x3FFu); }
__device__ __forceinline__ bool seg_first(const uint4& r) { return ((r.z >> 30) & 1u) != 0u; }
__device__ __forceinline__ int rec_qmean(const uint4& r) { return (int)(((r.z >> 11) & 31u) | ((r.w >> 23) & 0xE0u)); }
__device__ __forceinline__ int rec_n(const uint4& r) { return (int)(r.z >> 16); }
__device__ __forceinline__ uint32_t rec_nm(const uint4& r) { return r.w & 0xFFFFu; }
__device__ __forceinline__ int rec_mapq(const uint4& r) { return (int)((r.w >> 16) & 0xFFu); }
__device__ __forceinline__ uint32_t rec_flags(const uint4& r) { return r.w >> 24; }


}  // namespace dev
}  // namespace midas
