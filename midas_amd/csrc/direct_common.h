// Helpers shared by the kernels of the direct path (index_direct.hip, pileup_direct.hip): views of a read's CIGAR in the
// BAM-native array, the clip rules of pysam's query_alignment_start / _end, and the CIGAR shapes the pileup kernel settles in registers.
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

// The shape of a read's CIGAR as the pileup kernel settles it in registers: ONE or TWO gap-free match runs,
//   `H* S? (M|=|X)+ ((I|D|N) (M|=|X)+)? S? H*`, every length >= 1, the lengths of S / M / I adding up to l_seq
// -- everything an end-to-end or local aligner writes for a read with at most one indel.  The query positions
// [lead, lead + m1) lie on the sites pos ..., [lead + m1 + ins, lead + alen) on pos + m1 + del ...; alen = m1 + ins + m2 is
// len(aln.query_alignment_sequence) (midas/run/snps.py:145: pysam takes soft clips off both ends, inserted bases stay).
// Any other read is walked op by op (pileup_direct.hip, the slow path).
struct ReadShape { uint32_t lead, m1, ins, del, alen; };
constexpr uint32_t kMaxFastGap = 65535;     // a longer deletion / skip goes the slow way (its arithmetic saturates)

// One forward pass over the first four ops (registers); nc <= 4.  Stages: 0 leading hard clips, 1 behind the leading soft
// clip, 2 in the first run, 3 behind the indel, 4 in the second run, 5 behind the trailing soft clip, 6 trailing hard clips.
__device__ __forceinline__ bool decode_shape(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t nc, uint32_t l, ReadShape* out) {
  uint32_t stage = 0, lead = 0, trail = 0, m1 = 0, m2 = 0, ins = 0, del = 0;
  bool ok = nc >= 1u && nc <= 4u && l >= 1u;
  auto step = [&](uint32_t v) {
    const uint32_t op = v & 15u, len = v >> 4;
    if (len == 0u) ok = false;
    else if (op_is_match(op)) { if (stage <= 2u) { m1 += len; stage = 2u; } else if (stage <= 4u) { m2 += len; stage = 4u; } else ok = false; }
    else if (op == OP_S) { if (stage == 0u) { lead = len; stage = 1u; } else if (stage == 2u || stage == 4u) { trail = len; stage = 5u; } else ok = false; }
    else if (op == OP_I || op == OP_D || op == OP_N) { if (stage == 2u) { if (op == OP_I) ins = len; else del = len; stage = 3u; } else ok = false; }
    else if (op == OP_H) { if (stage == 2u || stage == 4u || stage == 5u) stage = 6u; else if (stage != 0u && stage != 6u) ok = false; }
    else ok = false;
  };
  if (nc > 0u) step(c0);
  if (nc > 1u) step(c1);
  if (nc > 2u) step(c2);
  if (nc > 3u) step(c3);
  // (sums of at most four 28-bit lengths: no wrap)
  ok = ok && stage != 3u && stage >= 2u && lead + m1 + ins + m2 + trail == l && del <= kMaxFastGap;
  out->lead = lead; out->m1 = m1; out->ins = ins; out->del = del; out->alen = m1 + ins + m2;
  return ok;
}

}  // namespace direct
}  // namespace midas
