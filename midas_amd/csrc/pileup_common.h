// Helpers shared by the two gfx950 pileup kernels (pileup_tiles.hip: barrier-phased workgroups, kept for the parts of split
// hot-spot tiles; pileup_stream.hip: the barrier-free streaming kernel that processes whole tiles).
#pragma once
#include "device_common.h"

namespace midas {
namespace pile {

using namespace dev;

// Developer ablation switches (tools/ablate.sh builds variants with -DMIDAS_SNPS_DEBUG_BITS=<bits>; several of the bits make
// the kernel produce wrong counts on purpose, for timing only).  A compile-time constant: the shipped library is built
// with 0 and carries none of that code, and nothing in the environment can switch it on.
#ifndef MIDAS_SNPS_DEBUG_BITS
#define MIDAS_SNPS_DEBUG_BITS 0
#endif
constexpr int kDebug = MIDAS_SNPS_DEBUG_BITS;

// Eight bases of the hot loop, hand-scheduled.  Per base: byte compare against baseq (SDWA) -> lane mask in an SGPR
// pair; address = chunk base | call code (SDWA OR); returnless LDS atomic under that mask.  The eight address ORs and
// the eight compares are issued back to back (no dependency between them), then each atomic runs under its mask and
// exec is restored once -- instead of compare -> saveexec -> OR -> atomic -> restore chained per base (measured:
// 156 -> 147 us).  Written as asm because the compiler wraps every predicated ds_add in an s_cbranch_execz skip branch.
//   qa,qb / ca,cb: two words of four quality / call-code bytes; abase: LDS byte address of the chunk's first site
//   (multiple of 16, so OR-ing the call code 0/4/8/12 selects the counter); OFF = byte offset of the first site.
template <int OFF>
__device__ __forceinline__ void tally_pair_at(uint32_t qa, uint32_t qb, uint32_t ca, uint32_t cb, uint32_t bq, uint32_t abase,
                                              uint32_t one) {
  uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
  unsigned long long m0, m1, m2, m3, m4, m5, m6, m7, save;
  asm volatile(
      "v_or_b32_sdwa %0, %21, %19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
      "v_or_b32_sdwa %1, %21, %19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
      "v_or_b32_sdwa %2, %21, %19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
      "v_or_b32_sdwa %3, %21, %19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
      "v_or_b32_sdwa %4, %21, %20 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
      "v_or_b32_sdwa %5, %21, %20 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
      "v_or_b32_sdwa %6, %21, %20 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
      "v_or_b32_sdwa %7, %21, %20 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
      "v_cmp_ge_u32_sdwa %8, %17, %22 src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %9, %17, %22 src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %10, %17, %22 src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %11, %17, %22 src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %12, %18, %22 src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %13, %18, %22 src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %14, %18, %22 src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_cmp_ge_u32_sdwa %15, %18, %22 src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "s_mov_b64 %16, exec\n\t"
      "s_mov_b64 exec, %8\n\t"
      "ds_add_u32 %0, %23 offset:%24\n\t"
      "s_mov_b64 exec, %9\n\t"
      "ds_add_u32 %1, %23 offset:%24+16\n\t"
      "s_mov_b64 exec, %10\n\t"
      "ds_add_u32 %2, %23 offset:%24+32\n\t"
      "s_mov_b64 exec, %11\n\t"
      "ds_add_u32 %3, %23 offset:%24+48\n\t"
      "s_mov_b64 exec, %12\n\t"
      "ds_add_u32 %4, %23 offset:%24+64\n\t"
      "s_mov_b64 exec, %13\n\t"
      "ds_add_u32 %5, %23 offset:%24+80\n\t"
      "s_mov_b64 exec, %14\n\t"
      "ds_add_u32 %6, %23 offset:%24+96\n\t"
      "s_mov_b64 exec, %15\n\t"
      "ds_add_u32 %7, %23 offset:%24+112\n\t"
      "s_mov_b64 exec, %16"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&s"(m0), "=&s"(m1),
        "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7), "=&s"(save)
      : "v"(qa), "v"(qb), "v"(ca), "v"(cb), "v"(abase), "s"(bq), "v"(one), "n"(OFF)
      : "memory");
}
__device__ __forceinline__ void tally_chunk(const uint32_t (&q4)[kChunk / 4], const uint32_t (&cd)[kChunk / 4], uint32_t bq,
                                            uint32_t abase, uint32_t one) {
  static_assert(kChunk == 32, "tally_chunk is written for 8 words");
  tally_pair_at<0>(q4[0], q4[1], cd[0], cd[1], bq, abase, one);
  tally_pair_at<128>(q4[2], q4[3], cd[2], cd[3], bq, abase, one);
  tally_pair_at<256>(q4[4], q4[5], cd[4], cd[5], bq, abase, one);
  tally_pair_at<384>(q4[6], q4[7], cd[6], cd[7], bq, abase, one);
}

// Byte mask with bytes [0, hi) of a 32-bit word set.
__device__ __forceinline__ uint32_t low_bytes_mask(int hi) {
  return hi >= 4 ? 0xFFFFFFFFu : (hi <= 0 ? 0u : ((1u << (8 * hi)) - 1u));
}
// Four mask bits (bit k <-> byte k) -> 0xFF / 0x00 bytes.
__device__ __forceinline__ uint32_t bits_to_bytes(uint32_t nib) {
  const uint32_t b = (nib * 0x00204081u) & 0x01010101u;   // bit k -> bit 8k
  return (b << 8) - b;                                      // 0x01 -> 0xFF in every byte (mod 2^32)
}
// str.upper() on four ASCII letters at once: bytes in 'a'..'z' lose bit 5, everything else is untouched.
__device__ __forceinline__ uint32_t upper4(uint32_t x) {
  const uint32_t t = x & 0x7F7F7F7Fu;
  const uint32_t ge_a = t + 0x1F1F1F1Fu;   // bit 7 set iff t >= 0x61
  const uint32_t gt_z = t + 0x05050505u;   // bit 7 set iff t >= 0x7B
  const uint32_t lower = ge_a & ~gt_z & ~x & 0x80808080u;
  return x - (lower >> 2);
}

typedef uint32_t u32_a1 __attribute__((aligned(1)));

// the three read ranges of a tile as one virtual stream: range starts; |S|, |S|+|I|, |S|+|I|+|G|
struct Ranges { int sb, ib, gb, ns, nsi, total; };
// the six index words of a tile as loaded (index_reads.hip): rbinv / rend of S, G, I
struct RawRanges { uint32_t vs, vg, vi, se, ge, ie; };

// Per-tile tables are read through the constant address space: the tile index is workgroup-uniform, so these become
// s_load (SGPR results, no VGPRs, tracked by lgkmcnt) and can be issued a whole tile ahead of their use.
typedef const __attribute__((address_space(4))) uint32_t* ConstWords;
__device__ __forceinline__ Tile load_tile(ConstWords tiles, int t) {
  static_assert(sizeof(Tile) == 32, "eight words per tile");
  const ConstWords w = tiles + 8 * (size_t)t;
  Tile x;
  x.contig = (int32_t)w[0]; x.start = (int32_t)w[1]; x.len = (int32_t)w[2]; x.species = (int32_t)w[3];
  x.site_base = (int64_t)((unsigned long long)w[4] | ((unsigned long long)w[5] << 32));
  x.contig_len = (int32_t)w[6]; x.halo = (int32_t)w[7];
  return x;
}

}  // namespace pile
}  // namespace midas
