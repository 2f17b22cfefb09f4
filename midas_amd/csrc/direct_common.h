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

// Forward iteration over a read's CIGAR -- the first four ops from registers, the rest from memory; body(k, op word) returns
// false to stop.  (Indexing the view with a run-time k makes the compiler keep the four words in scratch memory.)
template <class F>
__device__ __forceinline__ void for_each_op(const CigarView& cg, uint32_t nc, F body) {
  bool go = true;
  if (go && nc > 0u) go = body(0u, cg.c0);
  if (go && nc > 1u) go = body(1u, cg.c1);
  if (go && nc > 2u) go = body(2u, cg.c2);
  if (go && nc > 3u) go = body(3u, cg.c3);
  for (uint32_t k = 4; go && k < nc; ++k) go = body(k, cg.p[k]);
}

// [EXT] pysam getQueryStart (leading soft clips, hard clips skipped) and getQueryEnd (the backward walk over the ops n-1 .. 1
// -- index 0 is never inspected -- skipping hard clips, taking soft clips off the end until anything else turns up), both
// from ONE forward pass: the soft clips since the last op (of index >= 1) that is neither S nor H are the trailing ones.
__device__ __forceinline__ void query_bounds(const CigarView& cg, uint32_t nc, long long l_seq, long long* start, long long* end) {
  long long lead = 0, trail = 0;
  bool leading = true;
  for_each_op(cg, nc, [&](uint32_t k, uint32_t v) {
    const uint32_t op = v & 15u;
    const long long len = (long long)(v >> 4);
    if (leading) {
      if (op == OP_S) lead += len; else if (op != OP_H) leading = false;
    }
    if (k >= 1u) {
      if (op == OP_S) trail += len; else if (op != OP_H) trail = 0;
    }
    return true;
  });
  *start = lead;
  *end = l_seq - trail;
}

// Index record of a class-0 read, kIdxRecBytes = 20 bytes (five words), written by the classify kernel for the pileup kernel:
//   w0 leading clip | aligned length << 10 | trailing clip << 21 (bit 31 clear)        w1 pos        w2 QUAL offset, low word
//   w3 NM (11 bits) | mapq << 11 | bits 32-39 of the QUAL offset << 19 | bits 32-36 of the SEQ offset << 27        w4 SEQ offset, low word
// The record of any other read (and the sentinel behind the last read): w0 = bit 31 | the read's contig -- skipped where it lies.
// Descriptor of one (general read, tile) entry, kGenDescWords = 8 words:
//   w0 aligned length (pysam: query_alignment_end - _start, >= 0) | leading soft clip << 11 | bits 32-39 of the CIGAR offset << 22
//   w1 pos        w2 NM (16 bits, 0xFFFF = absent) | mapq << 16 | bits 32-39 of the SEQ offset << 24        w3 SEQ offset, low word
//   w4 QUAL offset, low word        w5 bits 32-39 of the QUAL offset | kGen* flags << 8
//   w6 CIGAR offset (elements), low word        w7 l_seq | n_cigar << 16
// (the read's index of an entry, needed only for an error report, is kept apart: gidx[entry]).  The pileup kernel fetches
// either kind with the same two unconditional 16-byte loads (the second one of a record overhangs into the next record).
constexpr int kIdxRecBytes = 20;
constexpr uint32_t kIdxMaxNm = 2047;
struct GenDesc {
  uint32_t idx; int32_t pos; uint32_t l, nc, nm16, mapq, flags, align_len, lead;
  unsigned long long so, qo, co;
};
__device__ __forceinline__ void gdesc_store(uint32_t* g, const GenDesc& d) {
  uint4* q = reinterpret_cast<uint4*>(g);
  q[0] = make_uint4(d.align_len | (d.lead << 11) | ((uint32_t)((d.co >> 32) & 0xFF) << 22), (uint32_t)d.pos,
                    d.nm16 | (d.mapq << 16) | ((uint32_t)((d.so >> 32) & 0xFF) << 24), (uint32_t)d.so);
  q[1] = make_uint4((uint32_t)d.qo, (uint32_t)((d.qo >> 32) & 0xFF) | (d.flags << 8), (uint32_t)d.co, d.l | (d.nc << 16));
}
__device__ __forceinline__ void gdesc_store_idle(uint32_t* g) {
  uint4* q = reinterpret_cast<uint4*>(g);
  q[0] = make_uint4(0u, 0u, 0u, 0u);
  q[1] = make_uint4(0u, (uint32_t)kGenIdle << 8, 0u, 0u);
}
__device__ __forceinline__ bool idxrec_fits(int32_t nm, unsigned long long so) { return (uint32_t)nm <= kIdxMaxNm && (so >> 37) == 0ull; }
__device__ __forceinline__ void idxrec_store(uint8_t* rec, size_t i, uint32_t info, int32_t pos, uint32_t nm, uint32_t mapq,
                                             unsigned long long so, unsigned long long qo) {
  uint32_t* r = reinterpret_cast<uint32_t*>(rec + i * kIdxRecBytes);
  u32x4_a4 v;
  v.x = info; v.y = (uint32_t)pos; v.z = (uint32_t)qo;
  v.w = nm | (mapq << 11) | ((uint32_t)((qo >> 32) & 0xFF) << 19) | ((uint32_t)((so >> 32) & 0x1F) << 27);
  *reinterpret_cast<u32x4_a4*>(r) = v;
  r[4] = (uint32_t)so;
}
// the record of a read that is not class 0 (and the sentinel behind the last read): nothing to do where it lies
__device__ __forceinline__ void idxrec_store_idle(uint8_t* rec, size_t i, uint32_t contig) {
  *reinterpret_cast<uint32_t*>(rec + i * kIdxRecBytes) = kInfoGeneral | contig;
}

}  // namespace direct
}  // namespace midas
