// Helpers shared by the kernels of the direct path (index_direct.hip, pileup_direct.hip): views of a read's CIGAR in the
// BAM-native array, the clip rules of pysam's query_alignment_start / _end, and the layout of a general read's descriptor.
#pragma once
#include "device_common.h"

namespace midas {
namespace direct {

using namespace dev;

__device__ __forceinline__ bool op_is_match(uint32_t op) { return op == OP_M || op == OP_EQ || op == OP_X; }

// A read's CIGAR: the first four ops arrive with one 16-byte load (nearly every CIGAR is that short), the rest on demand.
// (the array has 64 bytes of slack behind its last op)
struct CigarView {
  uint32_t c0, c1, c2, c3;
  const uint32_t* p;
  __device__ __forceinline__ void load(const uint32_t* q) {
    p = q;
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(q);
    c0 = v.x; c1 = v.y; c2 = v.z; c3 = v.w;
  }
  __device__ __forceinline__ uint32_t operator[](uint32_t k) const {
    return k < 4u ? (k < 2u ? (k == 0u ? c0 : c1) : (k == 2u ? c2 : c3)) : p[k];
  }
};

// [EXT] pysam getQueryStart: leading soft clips, hard clips skipped.
__device__ __forceinline__ long long query_start(const CigarView& cg, uint32_t n) {
  long long start = 0;
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t v = cg[k], op = v & 15u;
    if (op == OP_H) continue;
    if (op == OP_S) start += (long long)(v >> 4); else break;
  }
  return start;
}
// [EXT] pysam getQueryEnd: the backward walk over indices n-1 .. 1 (index 0 is never inspected).
__device__ __forceinline__ long long query_end(const CigarView& cg, uint32_t n, long long l_seq) {
  long long end = l_seq;
  for (uint32_t k = n; k-- > 1u;) {
    const uint32_t v = cg[k], op = v & 15u;
    if (op == OP_H) continue;
    if (op == OP_S) end -= (long long)(v >> 4); else break;
  }
  return end;
}

// Descriptor of one (general read, tile) entry, kGenDescWords = 12 words (three 16-byte loads):
//   w0 read index            w1 pos                    w2 l_seq | n_cigar << 16       w3 nm16 (0xFFFF = absent) | mapq << 16 | flags << 24
//   w4 aligned length (pysam: query_alignment_end - _start, >= 0) | leading soft clip << 16
//   w5 / w6 / w7 low words of the byte offsets into seq4 / qual and of the element offset into cigar
//   w8 their bits 32-39: seq | qual << 8 | cigar << 16
struct GenDesc {
  uint32_t idx; int32_t pos; uint32_t l, nc, nm16, mapq, flags, align_len, lead;
  unsigned long long so, qo, co;
};
__device__ __forceinline__ void gdesc_store(uint32_t* g, const GenDesc& d) {
  uint4* q = reinterpret_cast<uint4*>(g);
  q[0] = make_uint4(d.idx, (uint32_t)d.pos, d.l | (d.nc << 16), d.nm16 | (d.mapq << 16) | (d.flags << 24));
  q[1] = make_uint4(d.align_len | (d.lead << 16), (uint32_t)d.so, (uint32_t)d.qo, (uint32_t)d.co);
  q[2] = make_uint4((uint32_t)((d.so >> 32) & 0xFF) | ((uint32_t)((d.qo >> 32) & 0xFF) << 8) | ((uint32_t)((d.co >> 32) & 0xFF) << 16), 0u, 0u, 0u);
}

}  // namespace direct
}  // namespace midas
