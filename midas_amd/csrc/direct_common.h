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

// Branch-free, on the first four ops (registers).  An op's class -- 0 match (M, =, X), 1 soft clip, 2 gap (I, D, N), 3 anything
// else -- comes out of a 32-bit table; the classes of the read's ops, two bits each, form a key that is compared with the one
// key its op count and leading clip allow:
//   M | S M | M S | S M S | M G M | S M G M | M G M S          (S M G M S has five ops: it is walked)
// Reads with hard clips or a match run written as several ops (4=2X ...) are walked op by op as well -- correct, only slower.
// Zero-length ops are harmless here (the formulas hold for them); an empty aligned part is not (keep_read divides by it).
constexpr uint32_t kOpClass = 0xFFFC3DA8u;      // two bits per op code 0..15
constexpr uint32_t kShapeKeys = 0x48080400u;    // key of the ops behind the leading clip, by their number 1..4: M, M S, M G M, M G M S
__device__ __forceinline__ bool decode_shape(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t nc, uint32_t l, ReadShape* out) {
  const uint32_t op0 = c0 & 15u, op1 = c1 & 15u, op2 = c2 & 15u, op3 = c3 & 15u;
  const uint32_t len0 = c0 >> 4, len1 = c1 >> 4, len2 = c2 >> 4, len3 = c3 >> 4;
  const uint32_t k0 = __builtin_amdgcn_ubfe(kOpClass, op0 << 1, 2u), k1 = __builtin_amdgcn_ubfe(kOpClass, op1 << 1, 2u);
  const uint32_t k2 = __builtin_amdgcn_ubfe(kOpClass, op2 << 1, 2u), k3 = __builtin_amdgcn_ubfe(kOpClass, op3 << 1, 2u);
  const uint32_t ncc = nc > 5u ? 5u : nc;
  const uint32_t key = (k0 | (k1 << 2) | (k2 << 4) | (k3 << 6)) & ((1u << (2u * ncc)) - 1u);        // (the ops behind the read's are not its own)
  const bool lead_p = k0 == 1u;
  const uint32_t nr = ncc - (lead_p ? 1u : 0u);                                                       // ops behind the leading clip
  const uint32_t want_r = (kShapeKeys >> (8u * ((nr - 1u) & 3u))) & 0xFFu;
  const uint32_t want = lead_p ? (1u | (want_r << 2)) : want_r;
  const uint32_t r0 = lead_p ? len1 : len0, r1 = lead_p ? len2 : len1, r2 = lead_p ? len3 : len2;
  const uint32_t gop = lead_p ? op2 : op1;
  const bool has_gap = nr >= 3u;
  const uint32_t lead = lead_p ? len0 : 0u;
  const uint32_t gap = has_gap ? r1 : 0u;
  const uint32_t ins = gop == OP_I ? gap : 0u;
  const uint32_t m2 = has_gap ? r2 : 0u;
  const uint32_t trail = nr == 2u ? r1 : (nr == 4u ? len3 : 0u);
  const uint32_t alen = r0 + ins + m2;
  out->lead = lead; out->m1 = r0; out->ins = ins; out->del = gap - ins; out->alen = alen;
  // (sums of at most four 28-bit lengths: no wrap)
  return nc >= 1u && nc <= 4u && nr >= 1u && key == want && lead + alen + trail == l && alen >= 1u && gap - ins <= kMaxFastGap;
}

}  // namespace direct
}  // namespace midas
