// gfx950 (CDNA4) pileup kernel of the MIDAS SNP path that reads the BAM-native arrays themselves: 4-bit SEQ, QUAL, CIGAR and
// the per-read columns where the decoder put them -- no packed payload, no sort, one pass per read.  Integer counting: no MFMA.
//
// Reference semantics implemented here (citations into /root/reference):
//   keep_read                       midas/run/snps.py:141-162  (query_alignment_sequence :145, np.mean(query_qualities) :151)
//   count_coverage call site        midas/run/snps.py:194-199  ([EXT] pysam: get_aligned_pairs(matches_only), qual >= quality_threshold,
//                                   only 'A','C','G','T' counted)
//   depth / covered / total_depth   midas/run/snps.py:204-213
//   str(rec.seq).upper()            midas/run/snps.py:62
//
// Work decomposition: as in pileup_tiles.hip -- tiles of <= 4096 sites, a persistent 512-thread workgroup per item, tallies
// in LDS as [site][A,C,G,T] u32, one coalesced write-out per tile, items handed out by per-XCD counters.  What differs is
// where a tile's reads come from and what a lane does with them:
//
// Stream.  The index pass (index_direct.hip) left, per tile, the range [tbegin, tend) of read indices holding every class-0
// read (one gap-free match segment) that touches the tile -- the input is position-sorted, so that is a contiguous run of
// the read arrays -- and a list of 48-byte descriptors of the general reads touching it.  Both are dealt to the waves as
// ONE virtual stream, range first: the leading wave-iterations are pure class 0 and branch over the CIGAR walk.
//
// Lane mapping.  A lane owns LB (30 or 32) consecutive bases of a read's STORED query: two 16-byte loads of QUAL, one of
// 4-bit SEQ (LB is even, so a lane's bases start on a byte).  A read of l_seq bases takes ceil(l_seq / LB) adjacent lanes
// (5 for 150 bp) and a wave works on floor(64 / lanes) reads at a time.  Loads are issued two iterations (per-read columns)
// and one iteration (bases) ahead of their use.
//
// Per base, from the raw bytes:  the 4-bit codes of eight bases (one dword) are split into their even and odd nibbles
// (two masks), mapped to v_perm_b32 selectors by `(n + 7) ^ 8` -- A, C, G, T (1, 2, 4, 8) land on table slots 0, 1, 3, 7,
// every other code on a slot or a selector constant that yields 0xFF -- and looked up twice: a THRESHOLD byte (baseq - 1
// for A/C/G/T, 0xFF for anything else) and the byte offset of the base's counter.  Then per base one SDWA compare
// `qual.byte > threshold.byte` into a lane mask (the byte selects of the two operands are independent, so the even / odd
// order of the looked-up bytes costs nothing), one SDWA OR forming the LDS address, one returnless ds_add under the mask.
// Clipping (soft clips, segment borders, tile edges, the read's tail) ORs 0xFF into threshold bytes: two table rows from LDS.
// The read's mean quality is v_sad_u8 over the lane's bytes and a sum over the read's lanes; sum(q) < readq * l_seq is the
// reference's np.mean(q) < readq exactly.
#include "direct_common.h"
#include "pileup_common.h"

#include <type_traits>

namespace midas {

using namespace dev;
using namespace pile;
using namespace direct;

namespace {

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Eight bases of one lane: q0 / q1 two words of four quality bytes (bases in order), the / tho threshold bytes and cde / cdo
// counter offsets of the even / odd bases (byte i of an `e` word: base 2i, of an `o` word: base 2i + 1).
template <int OFF, int NB>
__device__ __forceinline__ void tally_group(uint32_t q0, uint32_t q1, uint32_t the, uint32_t tho, uint32_t cde, uint32_t cdo,
                                            uint32_t abase, uint32_t one) {
  static_assert(NB == 8 || NB == 6, "a group holds 8 bases, or 6 at the end of a 30-base lane");
  uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
  unsigned long long m0, m1, m2, m3, m4, m5, m6, m7, save;
  if (NB == 8) {
    asm volatile(
        "v_or_b32_sdwa %[t0], %[ab], %[ce] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_or_b32_sdwa %[t1], %[ab], %[co] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_or_b32_sdwa %[t2], %[ab], %[ce] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_or_b32_sdwa %[t3], %[ab], %[co] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_or_b32_sdwa %[t4], %[ab], %[ce] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_or_b32_sdwa %[t5], %[ab], %[co] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_or_b32_sdwa %[t6], %[ab], %[ce] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
        "v_or_b32_sdwa %[t7], %[ab], %[co] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
        "v_cmp_gt_u32_sdwa %[m0], %[q0], %[te] src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
        "v_cmp_gt_u32_sdwa %[m1], %[q0], %[to] src0_sel:BYTE_1 src1_sel:BYTE_0\n\t"
        "v_cmp_gt_u32_sdwa %[m2], %[q0], %[te] src0_sel:BYTE_2 src1_sel:BYTE_1\n\t"
        "v_cmp_gt_u32_sdwa %[m3], %[q0], %[to] src0_sel:BYTE_3 src1_sel:BYTE_1\n\t"
        "v_cmp_gt_u32_sdwa %[m4], %[q1], %[te] src0_sel:BYTE_0 src1_sel:BYTE_2\n\t"
        "v_cmp_gt_u32_sdwa %[m5], %[q1], %[to] src0_sel:BYTE_1 src1_sel:BYTE_2\n\t"
        "v_cmp_gt_u32_sdwa %[m6], %[q1], %[te] src0_sel:BYTE_2 src1_sel:BYTE_3\n\t"
        "v_cmp_gt_u32_sdwa %[m7], %[q1], %[to] src0_sel:BYTE_3 src1_sel:BYTE_3\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[m0]\n\t"
        "ds_add_u32 %[t0], %[one] offset:%[off]\n\t"
        "s_mov_b64 exec, %[m1]\n\t"
        "ds_add_u32 %[t1], %[one] offset:%[off]+16\n\t"
        "s_mov_b64 exec, %[m2]\n\t"
        "ds_add_u32 %[t2], %[one] offset:%[off]+32\n\t"
        "s_mov_b64 exec, %[m3]\n\t"
        "ds_add_u32 %[t3], %[one] offset:%[off]+48\n\t"
        "s_mov_b64 exec, %[m4]\n\t"
        "ds_add_u32 %[t4], %[one] offset:%[off]+64\n\t"
        "s_mov_b64 exec, %[m5]\n\t"
        "ds_add_u32 %[t5], %[one] offset:%[off]+80\n\t"
        "s_mov_b64 exec, %[m6]\n\t"
        "ds_add_u32 %[t6], %[one] offset:%[off]+96\n\t"
        "s_mov_b64 exec, %[m7]\n\t"
        "ds_add_u32 %[t7], %[one] offset:%[off]+112\n\t"
        "s_mov_b64 exec, %[sv]"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6),
          [t7] "=&v"(t7), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4), [m5] "=&s"(m5),
          [m6] "=&s"(m6), [m7] "=&s"(m7), [sv] "=&s"(save)
        : [q0] "v"(q0), [q1] "v"(q1), [te] "v"(the), [to] "v"(tho), [ce] "v"(cde), [co] "v"(cdo), [ab] "v"(abase), [one] "v"(one),
          [off] "n"(OFF)
        : "memory");
  } else {
    asm volatile(
        "v_or_b32_sdwa %[t0], %[ab], %[ce] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_or_b32_sdwa %[t1], %[ab], %[co] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_or_b32_sdwa %[t2], %[ab], %[ce] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_or_b32_sdwa %[t3], %[ab], %[co] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_or_b32_sdwa %[t4], %[ab], %[ce] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_or_b32_sdwa %[t5], %[ab], %[co] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_cmp_gt_u32_sdwa %[m0], %[q0], %[te] src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
        "v_cmp_gt_u32_sdwa %[m1], %[q0], %[to] src0_sel:BYTE_1 src1_sel:BYTE_0\n\t"
        "v_cmp_gt_u32_sdwa %[m2], %[q0], %[te] src0_sel:BYTE_2 src1_sel:BYTE_1\n\t"
        "v_cmp_gt_u32_sdwa %[m3], %[q0], %[to] src0_sel:BYTE_3 src1_sel:BYTE_1\n\t"
        "v_cmp_gt_u32_sdwa %[m4], %[q1], %[te] src0_sel:BYTE_0 src1_sel:BYTE_2\n\t"
        "v_cmp_gt_u32_sdwa %[m5], %[q1], %[to] src0_sel:BYTE_1 src1_sel:BYTE_2\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[m0]\n\t"
        "ds_add_u32 %[t0], %[one] offset:%[off]\n\t"
        "s_mov_b64 exec, %[m1]\n\t"
        "ds_add_u32 %[t1], %[one] offset:%[off]+16\n\t"
        "s_mov_b64 exec, %[m2]\n\t"
        "ds_add_u32 %[t2], %[one] offset:%[off]+32\n\t"
        "s_mov_b64 exec, %[m3]\n\t"
        "ds_add_u32 %[t3], %[one] offset:%[off]+48\n\t"
        "s_mov_b64 exec, %[m4]\n\t"
        "ds_add_u32 %[t4], %[one] offset:%[off]+64\n\t"
        "s_mov_b64 exec, %[m5]\n\t"
        "ds_add_u32 %[t5], %[one] offset:%[off]+80\n\t"
        "s_mov_b64 exec, %[sv]"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [m0] "=&s"(m0),
          [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4), [m5] "=&s"(m5), [sv] "=&s"(save)
        : [q0] "v"(q0), [q1] "v"(q1), [te] "v"(the), [to] "v"(tho), [ce] "v"(cde), [co] "v"(cdo), [ab] "v"(abase), [one] "v"(one),
          [off] "n"(OFF)
        : "memory");
  }
}

// The per-tile stream: the index records rb .. rb + n0 (class 0; a general read in the range is skipped there), then the
// tile's general descriptors gb .. gb + ng -- as wave-iterations: na of the first kind, then ng_it of the second.
struct Stream { int rb, n0, gb, ng, na, total; };

template <int LB, bool BQ0>
__global__ __launch_bounds__(kPileupBlock, 4) void pileup_direct_kernel(DirectParams p) {
  constexpr int TILE = kTileSites;
  constexpr int NWAVES = kPileupBlock / 64;
  constexpr int OUT_IT = TILE / kPileupBlock;
  __shared__ __attribute__((aligned(16))) uint32_t lds[4 * TILE];
  __shared__ __attribute__((aligned(16))) uint32_t s_mhi[33 * 8];   // [h][w]: 0xFF in the bytes of the bases j >= h
  __shared__ __attribute__((aligned(16))) uint32_t s_mlo[33 * 8];   // [l][w]: 0xFF in the bytes of the bases j <  l
  __shared__ uint32_t s_qsum[NWAVES * 64];                           // per wave and read slot: sum of a read's quality bytes
  __shared__ unsigned long long s_stats[MIDAS_STATS];
  __shared__ uint32_t s_next_ticket;
  extern __shared__ __attribute__((aligned(16))) int32_t s_tables[];   // [min_match table_len][min_align table_len]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w_end = p.n_tiles;
  const bool dynamic = (gridDim.x % kSchedGroups) == 0;
  const int sched_group = (int)(blockIdx.x % kSchedGroups);
  uint32_t* const sched = p.sched;
  int w = (int)blockIdx.x;
  if (w >= w_end) return;
  int w_next = w + (int)gridDim.x;

  {
    uint4* z = reinterpret_cast<uint4*>(lds);
    for (int i = tid; i < TILE; i += kPileupBlock) z[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < p.table_len; i += kPileupBlock) {
      s_tables[i] = p.filt->min_match[i];
      s_tables[p.table_len + i] = p.filt->min_align[i];
    }
    for (int i = tid; i < 33 * 8; i += kPileupBlock) {
      const int h = i >> 3, wd = i & 7;
      uint32_t mh = 0, ml = 0;
      for (int b = 0; b < 4; ++b) {
        const int j = 8 * (wd >> 1) + 2 * b + (wd & 1);       // base held by byte b of word wd (even / odd split)
        if (j >= h) mh |= 0xFFu << (8 * b);
        if (j < h) ml |= 0xFFu << (8 * b);
      }
      s_mhi[i] = mh;
      s_mlo[i] = ml;
    }
    s_qsum[tid] = 0u;
    if (tid < MIDAS_STATS) s_stats[tid] = 0ull;
  }

  const int lpr = p.lanes_per_read;
  const int rpw = p.reads_per_wave;
  const int g = lane / lpr;
  const int c = lane - g * lpr;
  const bool lane_used = g < rpw;
  const int q0 = c * LB;                           // first base of the lane in the read's stored query
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds;
  const uint32_t qsum_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)(s_qsum + wave * 64 + g);
  // v_perm_b32 tables (slots 0, 1, 3, 7 = A, C, G, T): threshold bytes and counter offsets
  const uint32_t thr = BQ0 ? 0u : (uint32_t)(p.baseq > 256 ? 255 : p.baseq - 1);
  const uint32_t th_lo = thr | (thr << 8) | 0x00FF0000u | (thr << 24), th_hi = 0x00FFFFFFu | (thr << 24);
  const uint32_t cd_lo = 0x08000400u, cd_hi = 0x0C000000u;
  const int rq = p.readq < 0 ? 0 : (p.readq > 256 ? 256 : p.readq);   // sum(q) < rq * l  <=>  np.mean(q) < readq (q <= 255)
  const uint32_t one = 1u;

  const ConstWords c_tiles = (ConstWords)(size_t)p.tiles;
  const ConstWords c_tb = (ConstWords)(size_t)p.tbegin;
  const ConstWords c_te = (ConstWords)(size_t)p.tend;
  const ConstWords c_go = (ConstWords)(size_t)p.goff;
  auto load_stream = [&](int tt) -> Stream {
    Stream s;
    const uint32_t b = c_tb[tt], e = c_te[tt], g0 = c_go[tt], g1 = c_go[tt + 1];
    s.rb = e > b ? (int)b : 0;
    s.n0 = e > b ? (int)(e - b) : 0;
    s.gb = (int)g0;
    s.ng = (int)(g1 - g0);
    s.na = (s.n0 + rpw - 1) / rpw;
    s.total = s.na + (s.ng + rpw - 1) / rpw;
    return s;
  };

  // Sum of a read's quality bytes over its lanes (np.mean(aln.query_qualities), midas/run/snps.py:151): every lane adds its
  // part to the read's LDS slot, reads the slot back and clears it -- three LDS operations of one wave, executed in
  // order, instead of a shuffle per lane of the read.  Bit 31: QUAL absent.
  auto read_sum = [&](uint32_t part) -> uint32_t {
    uint32_t tot;
    asm volatile("ds_add_u32 %1, %2\n\tds_read_b32 %0, %1\n\tds_write_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(tot) : "v"(qsum_addr), "v"(part), "v"(0u) : "memory");
    return tot;
  };
  auto lane_qsum = [&](const uint32_t (&q)[8], int nb) -> uint32_t {
    uint32_t part = 0;
    if (nb == LB) {
#pragma unroll
      for (int k = 0; k < 7; ++k) part = __builtin_amdgcn_sad_u8(q[k], 0u, part);
      part = __builtin_amdgcn_sad_u8(LB == 32 ? q[7] : (q[7] & 0x0000FFFFu), 0u, part);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) part = __builtin_amdgcn_sad_u8(q[k] & low_bytes_mask(nb - 4 * k), 0u, part);
    }
    if (c == 0) part |= ((q[0] & 0xFFu) == 0xFFu) ? 0x80000000u : 0u;   // QUAL absent (BAM: first byte 0xFF)
    return part;
  };
  // Threshold bytes and counter offsets of eight bases from their 4-bit codes (one dword of SEQ): the even and the odd
  // nibbles become v_perm_b32 selectors by (n + 7) ^ 8 -- A, C, G, T (1, 2, 4, 8) select table slots 0, 1, 3, 7, every
  // other code a slot or a selector constant that reads 0xFF.
  // The lanes `go` tally bases [lo, hi) of their 30 / 32, the first of the lane at tile-relative site loc0.  Group by group
  // (decode eight bases, tally them), so that only one group's looked-up bytes are alive at a time.
  auto tally_range = [&](bool go, int lo, int hi, int loc0, const uint32_t (&qv)[8], const uint32_t (&sq)[4]) {
    const uint32_t abase = ((uint32_t)loc0 << 4) + lds_base;
    const bool masked = __ballot(go && (lo > 0 || hi < LB)) != 0ull;   // partial lanes: 0xFF into the threshold bytes
    if (!go) return;                                                    // outside [lo, hi) (a row of each table, LDS)
    const uint32_t* mh = s_mhi + 8 * (hi > 32 ? 32 : hi);
    const uint32_t* ml = s_mlo + 8 * (lo < 0 ? 0 : lo);
    auto group = [&](auto sidx, auto off, auto nbases) {
      constexpr int S = decltype(sidx)::value;
      const uint32_t x = sq[S];
      const uint32_t se = (((x >> 4) & 0x0F0F0F0Fu) + 0x07070707u) ^ 0x08080808u;   // even bases (high nibbles)
      const uint32_t so = ((x & 0x0F0F0F0Fu) + 0x07070707u) ^ 0x08080808u;          // odd bases
      uint32_t te = __builtin_amdgcn_perm(th_hi, th_lo, se), to = __builtin_amdgcn_perm(th_hi, th_lo, so);
      const uint32_t ce = __builtin_amdgcn_perm(cd_hi, cd_lo, se), co = __builtin_amdgcn_perm(cd_hi, cd_lo, so);
      if (masked) {
        const uint2 h = *reinterpret_cast<const uint2*>(mh + 2 * S), l = *reinterpret_cast<const uint2*>(ml + 2 * S);
        te |= h.x | l.x;
        to |= h.y | l.y;
      }
      tally_group<decltype(off)::value, decltype(nbases)::value>(qv[2 * S], qv[2 * S + 1], te, to, ce, co, abase, one);
    };
    using std::integral_constant;
    group(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, 8>{});
    group(integral_constant<int, 1>{}, integral_constant<int, 128>{}, integral_constant<int, 8>{});
    group(integral_constant<int, 2>{}, integral_constant<int, 256>{}, integral_constant<int, 8>{});
    group(integral_constant<int, 3>{}, integral_constant<int, 384>{}, integral_constant<int, (LB == 32 ? 8 : 6)>{});
  };

  // ---- stage F: the record / descriptor of this lane's read in wave-iteration `it` of a tile's stream.  Raw loads: nothing is
  // computed from them here, and NO load sits in a branch -- a lane without a read fetches the sentinel record, a lane
  // without bases the first bytes of the arrays -- so that the compiler can count the loads in flight (a load in a branch
  // makes it wait for every outstanding load, vmcnt(0), before the first use of any of them: the prefetch of the next
  // iteration's bases would be waited for at once).
  struct Raw { uint4 a, b; };
  const uint8_t* const gd_base = reinterpret_cast<const uint8_t*>(p.gdesc);
  const uint8_t* const idle_rec = p.rec + (size_t)p.n_reads * kIdxRecBytes;       // the two sentinels
  const uint8_t* const idle_gd = gd_base + (size_t)p.gdesc_capacity * (kGenDescWords * 4);
  auto fetch = [&](const Stream& st, int it) -> Raw {
    const bool gen = it >= st.na;                                   // (wave-uniform)
    const int v = (gen ? it - st.na : it) * rpw + g;
    const bool act = lane_used && it < st.total && v < (gen ? st.ng : st.n0);
    const uint8_t* src = gen ? gd_base + (size_t)(uint32_t)(st.gb + v) * (kGenDescWords * 4) : p.rec + (size_t)(uint32_t)(st.rb + v) * kIdxRecBytes;
    src = act ? src : (gen ? idle_gd : idle_rec);
    Raw f;
    const u32x4_a4 a = *reinterpret_cast<const u32x4_a4*>(src);
    const u32x4_a4 b = *reinterpret_cast<const u32x4_a4*>(src + 16);      // (of a 20-byte record: its last word and the next record's head)
    f.a = make_uint4(a.x, a.y, a.z, a.w);
    f.b = make_uint4(b.x, b.y, b.z, b.w);
    return f;
  };
  // ---- stage D: what the read's processing needs of its record, and the lane's bases (two 16-byte loads of QUAL, one of SEQ)
  //   a: class 0 the info word; general: aligned length | leading clip << 11 | CIGAR offset bits 32-39 << 22
  //   nmq: NM | mapq << 16 | kGen* flags << 24 (kGenIdle: nothing to do)        l_nc: l_seq | n_cigar << 16
  struct Rd { uint32_t a, pos, nmq, l_nc, co_lo; };
  struct Dat { uint32_t q[8]; uint32_t s[4]; };
  auto settle = [&](const Raw& f, bool gen, Rd& r, Dat& d) {
    unsigned long long so, qo;
    int l;
    r.a = f.a.x; r.pos = f.a.y;
    if (!gen) {       // (wave-uniform; no load in here)
      const bool act = !(f.a.x >> 31);
      l = (int)((f.a.x & 1023u) + ((f.a.x >> kInfoAlenShift) & 2047u) + ((f.a.x >> kInfoTrailShift) & 1023u));
      l = act ? l : 0;
      r.nmq = (f.a.w & 2047u) | (((f.a.w >> 11) & 0xFFu) << 16) | (act ? 0u : (uint32_t)kGenIdle << 24);
      r.l_nc = (uint32_t)l;
      r.co_lo = 0u;
      qo = (unsigned long long)f.a.z | ((unsigned long long)((f.a.w >> 19) & 0xFFu) << 32);
      so = (unsigned long long)f.b.x | ((unsigned long long)(f.a.w >> 27) << 32);
    } else {
      l = (int)(f.b.w & 0xFFFFu);                                    // (0 in the sentinel)
      r.nmq = (f.a.z & 0x00FFFFFFu) | (((f.b.y >> 8) & 0xFFu) << 24);
      r.l_nc = f.b.w;
      r.co_lo = f.b.z;
      qo = (unsigned long long)f.b.x | ((unsigned long long)(f.b.y & 0xFFu) << 32);
      so = (unsigned long long)f.a.w | ((unsigned long long)(f.a.z >> 24) << 32);
    }
    const bool has = q0 < l && !(kDebug & 128);
    const uint8_t* qp = p.qual + (has ? qo + (unsigned long long)q0 : 0ull);
    const uint8_t* sp = p.seq4 + (has ? so + (unsigned long long)(q0 >> 1) : 0ull);
    const u32x4_a1 qa = *reinterpret_cast<const u32x4_a1*>(qp);
    const u32x4_a1 qb = *reinterpret_cast<const u32x4_a1*>(qp + 16);
    const u32x4_a1 sv = *reinterpret_cast<const u32x4_a1*>(sp);
    d.q[0] = qa.x; d.q[1] = qa.y; d.q[2] = qa.z; d.q[3] = qa.w;
    d.q[4] = qb.x; d.q[5] = qb.y; d.q[6] = qb.z; d.q[7] = qb.w;
    d.s[0] = sv.x; d.s[1] = sv.y; d.s[2] = sv.z; d.s[3] = sv.w;
  };

  Tile tile = load_tile(c_tiles, w);
  Stream st = load_stream(w);
  Raw raw_n = fetch(st, wave + NWAVES);
  bool gen_n = wave + NWAVES >= st.na;       // kind of the iteration raw_n belongs to (wave-uniform)
  Rd rd_cur;
  Dat dat_cur;
  {
    const Raw raw_c = fetch(st, wave);
    settle(raw_c, wave >= st.na, rd_cur, dat_cur);
  }
  __syncthreads();   // LDS zeroed, tables in place

  uint32_t acc_cov = 0u;                 // (a thread's sites between two flushes: far below 2^32)
  unsigned long long acc_depth = 0ull;
  int t = w;
  for (;;) {
    const int tile_len = tile.len;
    const int tile_start = tile.start;
    const int it_hi = st.total;
    uint32_t w_aligned = 0, w_mapped = 0;
    constexpr int REF_IT = TILE / (4 * kPileupBlock);

    // the record / base prefetch runs across the tile boundary (as in pileup_tiles.hip): a wave's last two iterations fetch
    // the records of its first two iterations of the NEXT tile
    const int n_w = it_hi > wave ? (it_hi - wave + NWAVES - 1) / NWAVES : 0;
    const bool xt = w_next < w_end && n_w >= 2;
    Stream xs = st;
    if (xt) xs = load_stream(w_next);
    for (int it = wave; it < it_hi; it += NWAVES) {
      Rd rd_n;
      Dat dat_n;
      settle(raw_n, gen_n, rd_n, dat_n);       // (the record of the next iteration has arrived: its bases are requested ...)
      {                                        // ... then the record of the one after it
        const bool over = xt && it + 2 * NWAVES >= it_hi;
        Stream fs;
        fs.rb = over ? xs.rb : st.rb; fs.n0 = over ? xs.n0 : st.n0; fs.gb = over ? xs.gb : st.gb; fs.ng = over ? xs.ng : st.ng;
        fs.na = over ? xs.na : st.na; fs.total = over ? xs.total : st.total;
        const int fi = over ? ((it + NWAVES < it_hi) ? wave : wave + NWAVES) : it + 2 * NWAVES;
        raw_n = fetch(fs, fi);
        gen_n = fi >= fs.na;
      }
      const bool gen_cur = it >= st.na;        // (wave-uniform) a general-descriptor iteration

      if (kDebug & 4) {          // (developer timing variant: the stream of loads only)
        asm volatile("" :: "v"(dat_cur.q[0]), "v"(dat_cur.q[7]), "v"(dat_cur.s[0]), "v"(dat_cur.s[3]), "v"(rd_cur.a));
        rd_cur = rd_n; dat_cur = dat_n;
        continue;
      }
      const int pos = (int)rd_cur.pos;
      const int l = (int)(rd_cur.l_nc & 0xFFFFu);
      const int nb = l - q0 < LB ? (l - q0 < 0 ? 0 : l - q0) : LB;     // bases of the read in this lane
      const bool has = nb > 0;
      uint32_t qsum = read_sum(has ? lane_qsum(dat_cur.q, nb) : 0u);
      const bool t_noqual = (qsum >> 31) != 0u;
      qsum &= 0x7FFFFFFFu;
      const int nm = (int)(rd_cur.nmq & 0xFFFFu), mapq = (int)((rd_cur.nmq >> 16) & 0xFFu);
      uint32_t qv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) qv[k] = BQ0 ? 0x01010101u : dat_cur.q[k];
      bool keep, owner;
      uint32_t err;
      if (!gen_cur) {
        // ======================= class 0: one gap-free match segment ============================================================
        const bool act = !((rd_cur.nmq >> 24) & kGenIdle);
        const int lead = (int)(rd_cur.a & 1023u);
        const int align_len = (int)((rd_cur.a >> kInfoAlenShift) & 2047u);
        // ---- keep_read (midas/run/snps.py:141-162): a class-0 read has SEQ, NM and a non-empty aligned part -----------------
        const int min_match = s_tables[align_len < p.table_len ? align_len : 0];
        const int min_align = s_tables[p.table_len + (l < p.table_len ? l : 0)];
        const bool t_pid = align_len - nm < min_match;                                                       // pid < mapid
        const bool t_drop = ((int)qsum < rq * l) | (mapq < p.mapq_min) | (align_len < min_align);             // readq, mapq, aln_cov
        err = (act && !t_pid && t_noqual) ? (uint32_t)E_NO_QUAL : 0u;
        keep = act && !(t_pid | t_noqual | t_drop);
        const int rel = pos - tile_start;            // 0 <= pos < contig length: no wrap
        owner = act && rel >= 0 && rel < tile_len;
        // the segment: query [lead, lead + align_len) at sites pos ...; this lane's part of it, clipped to the tile
        const int loc0 = rel + (q0 - lead);
        int lo = lead - q0;
        lo = lo > -loc0 ? lo : -loc0;
        lo = lo > 0 ? lo : 0;
        int hi = lead + align_len - q0;
        hi = hi < tile_len - loc0 ? hi : tile_len - loc0;
        hi = hi < nb ? hi : nb;
        const bool go = keep && lo < hi;
        if (!(kDebug & 1) && __ballot(go) != 0ull) tally_range(go, lo, hi, loc0, qv, dat_cur.s);
      } else {
        // ======================= general reads: descriptors, walked op by op =================================================
        const uint32_t gflags = rd_cur.nmq >> 24;
        const bool act = !(gflags & kGenIdle);
        const int nc = (int)(rd_cur.l_nc >> 16);
        const int align_len = (int)(rd_cur.a & 2047u);
        const uint32_t* cig = p.cigar + ((size_t)rd_cur.co_lo | ((size_t)((rd_cur.a >> 22) & 0xFFu) << 32));
        // A read with ONE indel between two match runs (kGenInline) carries its geometry in the descriptor: no CIGAR is fetched
        // for it.  For anything else the first four ops come with one 16-byte load, issued here and first needed behind the
        // filter (not prefetched: it would cost eight registers of the double-buffered bases).
        const bool inl = (gflags & kGenInline) != 0u;
        uint32_t cg0 = 0u, cg1 = 0u, cg2 = 0u, cg3 = 0u;
        if (act && nc > 0 && !inl) {
          const u32x4_a4 cv = *reinterpret_cast<const u32x4_a4*>(cig);   // (may overhang into the array's slack)
          cg0 = cv.x; cg1 = cv.y; cg2 = cv.z; cg3 = cv.w;
        }
        // ---- keep_read, every test evaluated, the reference's order decides which outcome wins ---------------------------
        const int min_match = s_tables[align_len < p.table_len ? align_len : 0];
        const int min_align = s_tables[p.table_len + (l < p.table_len ? l : 0)];
        const bool t_noseq = l == 0;
        const bool t_nonm = (gflags & kGenNoNm) != 0u;
        const bool t_zero = align_len == 0;
        const bool t_pid = align_len - nm < min_match;
        const bool t_drop = ((int)qsum < rq * l) | (mapq < p.mapq_min) | (align_len < min_align);
        const bool t_over = (gflags & kGenOverrun) != 0u;
        err = t_over ? (uint32_t)E_CIGAR_OVERRUN : 0u;
        err = t_drop ? 0u : err;
        err = t_noqual ? (uint32_t)E_NO_QUAL : err;
        err = t_pid ? 0u : err;
        err = t_zero ? (uint32_t)E_ZERO_ALIGN : err;
        err = t_nonm ? (uint32_t)E_NO_NM : err;
        err = t_noseq ? (uint32_t)E_NO_SEQ : err;
        err = act ? err : 0u;
        keep = act && !(t_noseq | t_nonm | t_zero | t_pid | t_noqual | t_drop | t_over);
        // owner tile of a read = the tile holding its (clamped) start: it alone counts the read in the stats
        int cpos = pos < 0 ? 0 : pos;
        cpos = cpos > tile.contig_len - 1 ? tile.contig_len - 1 : cpos;
        owner = act && cpos >= tile_start && cpos < tile_start + tile_len && !(tile.halo && pos < 0);
        const int rel = pos - tile_start;                                     // may wrap for absurd positions:
        int rrel = (rel > (1 << 25) || rel < -(1 << 30)) ? (1 << 25) : rel;   // those are parked far right
        // ---- CIGAR walk ([EXT] get_aligned_pairs(matches_only=True)): one match segment at a time ------------------------
        // 32-bit saturating positions: a query position only matters below q1 <= 1024 and a tile-relative reference
        // position only below 4096, and both only ever grow.
        int k = 0, qpos = 0, jlo = 0, jhi = 0, loc0 = 0;
        const int q1 = q0 + nb;
        const int lead_g = (int)((rd_cur.a >> 11) & 2047u);
        const int in_m1 = (int)(rd_cur.co_lo & 1023u), in_ins = (int)((rd_cur.co_lo >> 10) & 1023u), in_del = (int)(rd_cur.co_lo >> 20);
        auto next_segment = [&]() -> bool {
          if (inl) {       // the two runs: query [lead, lead + m1) at pos ..., query [lead + m1 + ins, lead + alen) at pos + m1 + del ...
            while (k < 2) {
              const int qa = k == 0 ? lead_g : lead_g + in_m1 + in_ins;
              const int qb = k == 0 ? lead_g + in_m1 : lead_g + align_len;
              const int roff = k == 0 ? 0 : in_m1 + in_del;
              ++k;
              const int lo = qa > q0 ? qa : q0;
              const int hi = qb < q1 ? qb : q1;
              if (lo < hi) { jlo = lo - q0; jhi = hi - q0; loc0 = rrel + roff + (q0 - qa); return true; }
            }
            return false;
          }
          while (k < nc) {
            const uint32_t v = k < 4 ? (k == 0 ? cg0 : (k == 1 ? cg1 : (k == 2 ? cg2 : cg3))) : cig[k];
            ++k;
            const uint32_t op = v & 15u;
            const int len = (int)(v >> 4);
            const bool m = consumes_both(op);
            bool found = false;
            if (m) {
              const int lo = qpos > q0 ? qpos : q0;
              const int hi = (qpos + len) < q1 ? (qpos + len) : q1;
              found = lo < hi;
              if (found) { jlo = lo - q0; jhi = hi - q0; loc0 = rrel + (q0 - qpos); }
            }
            if (m || op == OP_I || op == OP_S || (op == OP_P && p.pad_advances)) { qpos += len; qpos = qpos > (1 << 29) ? (1 << 29) : qpos; }
            if (m || op == OP_D || op == OP_N) { rrel += len; rrel = rrel > (1 << 29) ? (1 << 29) : rrel; }
            if (found) return true;   // H, P and anything else: no effect
          }
          return false;
        };
        bool walking = keep && has;
        if (walking) walking = next_segment();
        while (__ballot(walking) != 0ull) {
          const int lo = jlo > -loc0 ? jlo : -loc0;
          const int hi = jhi < tile_len - loc0 ? jhi : tile_len - loc0;
          tally_range(walking && lo < hi, lo, hi, loc0, qv, dat_cur.s);
          walking = (walking && k < (inl ? 2 : nc)) ? next_segment() : false;
        }
      }
      // ---- per-species read counters: one ballot per wave ---------------------------------------------------------------
      const bool head = owner && c == 0;
      w_aligned += (uint32_t)__popcll(__ballot(head));
      w_mapped += (uint32_t)__popcll(__ballot(head && keep));
      // (a read of the piece in front, midas_snps_contigs.origin: its own piece reports what keep_read raises, this one the
      // overrun its walk runs into here)
      const bool walk_err = gen_cur && tile.halo && pos < 0 && c == 0 && err == (uint32_t)E_CIGAR_OVERRUN;
      if ((head && err) || walk_err) {     // (the read's index: its place in the range, or the entry's word of gidx)
        const uint32_t idx = gen_cur ? p.gidx[(size_t)(st.gb + (it - st.na) * rpw + g)] : (uint32_t)(st.rb + it * rpw + g);
        atomicMin(p.err, ((unsigned long long)idx << 8) | err);
      }

      rd_cur = rd_n;
      dat_cur = dat_n;
    }

    // the tile's reference letters: requested here, behind the stream loop (two registers less in it), they arrive while
    // the workgroup waits for its last wave and writes the counts out
    uint32_t refw[REF_IT];
    if (p.out_allele) {
      const uint8_t* ref = p.ref + tile.site_base;
#pragma unroll
      for (int it = 0; it < REF_IT; ++it) {
        const int i = 4 * (tid + it * kPileupBlock);
        if (i + 4 <= tile_len) refw[it] = *reinterpret_cast<const u32_a1*>(ref + i);
      }
    }
    if (lane == 0) {
      if (w_aligned) atomicAdd(&s_stats[MIDAS_STAT_ALIGNED], (unsigned long long)w_aligned);
      if (w_mapped) atomicAdd(&s_stats[MIDAS_STAT_MAPPED], (unsigned long long)w_mapped);
    }
    // ---- next tile ---------------------------------------------------------------------------------------------------------
    const int wn = w_next;
    const bool more = wn < w_end;
    const int tn = more ? wn : t;
    const Tile ntile = load_tile(c_tiles, tn);
    const Stream nst = load_stream(tn);
    Raw raw_c;
    if (more && !xt) {   // (with xt the pipeline already holds the next tile's first two iterations)
      raw_c = fetch(nst, wave);
      raw_n = fetch(nst, wave + NWAVES);
      gen_n = wave + NWAVES >= nst.na;
    }
    lds_barrier();       // every tally of this tile is in LDS
    if (more && !xt) settle(raw_c, wave >= nst.na, rd_cur, dat_cur);
    uint32_t ticket = 0;
    if (dynamic && more && tid == 0) ticket = atomicAdd(&sched[32 * sched_group], 1u);

    // ---- emit the tile: counts[site][A,C,G,T] (and re-zero LDS), covered / total-depth partials ---------------------------
    {
      uint4* out = reinterpret_cast<uint4*>(p.out_counts) + ((kDebug & 8) ? 0 : tile.site_base);
      uint4* lds4 = reinterpret_cast<uint4*>(lds);
      const int lim = (kDebug & 2) ? 0 : tile_len;
#pragma unroll
      for (int it = 0; it < OUT_IT; ++it) {
        const int i = tid + it * kPileupBlock;
        if (i < lim) {
          const uint4 v = lds4[i];
          lds4[i] = make_uint4(0u, 0u, 0u, 0u);
          u32x4_a8 nv; nv.x = v.x; nv.y = v.y; nv.z = v.z; nv.w = v.w;
          __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_a8*>(out + i));
          const uint32_t d = v.x + v.y + v.z + v.w;
          acc_cov += d > 0u ? 1u : 0u;
          acc_depth += d;
        }
      }
    }
    if (p.out_allele && !(kDebug & 2)) {
      const uint8_t* ref = p.ref + tile.site_base;
      uint8_t* al = p.out_allele + tile.site_base;
#pragma unroll
      for (int it = 0; it < REF_IT; ++it) {
        const int i = 4 * (tid + it * kPileupBlock);
        if (i + 4 <= tile_len) {
          __builtin_nontemporal_store(upper4(refw[it]), reinterpret_cast<u32_a1*>(al + i));
        } else {
          for (int j = i; j < tile_len; ++j) {
            uint32_t ch = ref[j];
            if (ch >= 'a' && ch <= 'z') ch -= 32u;
            al[j] = (uint8_t)ch;
          }
        }
      }
    }
    if (dynamic && more && tid == 0) s_next_ticket = ticket;
    lds_barrier();       // tallies re-zeroed, this tile's s_stats additions done
    if (more) {
      const long long nn = dynamic ? 2ll * (long long)gridDim.x + (long long)kSchedGroups * s_next_ticket + sched_group
                                   : (long long)wn + (long long)gridDim.x;
      w_next = __builtin_amdgcn_readfirstlane((int)(nn < (long long)w_end ? nn : (long long)w_end));
    }
    const bool flush = !more || ntile.species != tile.species;   // workgroup-uniform
    if (flush) {
      for (int d = 32; d >= 1; d >>= 1) {
        acc_cov += __shfl_down(acc_cov, d);
        acc_depth += __shfl_down(acc_depth, d);
      }
      if (lane == 0) {
        if (acc_cov) atomicAdd(&s_stats[MIDAS_STAT_COVERED], (unsigned long long)acc_cov);
        if (acc_depth) atomicAdd(&s_stats[MIDAS_STAT_DEPTH], acc_depth);
      }
      acc_cov = 0u;
      acc_depth = 0ull;
      lds_barrier();
      if (tid < MIDAS_STATS) {
        const unsigned long long v = s_stats[tid];
        if (v) atomicAdd(&p.stats[(size_t)tile.species * MIDAS_STATS + tid], v);
        s_stats[tid] = 0ull;
      }
      if (!more) {
        if (dynamic && tid == 0) {   // the last workgroup to leave rewinds the counters for the next launch
          if (atomicAdd(&sched[32 * kSchedGroups], 1u) == gridDim.x - 1u) {
            for (int k = 0; k <= kSchedGroups; ++k) sched[32 * k] = 0u;
          }
        }
        break;
      }
      lds_barrier();     // s_stats reset before the next tile adds to it
    }
    w = wn;
    t = tn;
    tile = ntile;
    st = nst;
  }
}

}  // namespace

int direct_lane_bases(int32_t max_l_seq) {
  // 30 bases per lane: the lanes of a read start 120 tally dwords apart and spread over the LDS banks; 32 only where it
  // saves a whole lane per read (151 bp: 5 lanes instead of 6)
  const int l = max_l_seq > 0 ? max_l_seq : 1;
  return (l + 31) / 32 < (l + 29) / 30 ? 32 : 30;
}

hipError_t launch_pileup_direct(const DirectParams& p, int lane_bases, hipStream_t stream) {
  if (p.n_tiles <= 0) return hipSuccess;
  const size_t dyn_lds = (size_t)p.table_len * 2 * sizeof(int32_t);
  const int grid = p.n_tiles < p.grid_blocks ? p.n_tiles : p.grid_blocks;
  const bool bq0 = p.baseq <= 0;
  if (lane_bases == 32) {
    if (bq0) hipLaunchKernelGGL((pileup_direct_kernel<32, true>), dim3(grid), dim3(kPileupBlock), dyn_lds, stream, p);
    else hipLaunchKernelGGL((pileup_direct_kernel<32, false>), dim3(grid), dim3(kPileupBlock), dyn_lds, stream, p);
  } else {
    if (bq0) hipLaunchKernelGGL((pileup_direct_kernel<30, true>), dim3(grid), dim3(kPileupBlock), dyn_lds, stream, p);
    else hipLaunchKernelGGL((pileup_direct_kernel<30, false>), dim3(grid), dim3(kPileupBlock), dyn_lds, stream, p);
  }
  return hipGetLastError();
}

}  // namespace midas
