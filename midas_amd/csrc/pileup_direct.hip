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

template <int LB>
__device__ __forceinline__ void tally_lane(const uint32_t (&q)[8], const uint32_t (&th)[8], const uint32_t (&cd)[8], uint32_t abase,
                                           uint32_t one) {
  tally_group<0, 8>(q[0], q[1], th[0], th[1], cd[0], cd[1], abase, one);
  tally_group<128, 8>(q[2], q[3], th[2], th[3], cd[2], cd[3], abase, one);
  tally_group<256, 8>(q[4], q[5], th[4], th[5], cd[4], cd[5], abase, one);
  tally_group<384, (LB == 32 ? 8 : 6)>(q[6], q[7], th[6], th[7], cd[6], cd[7], abase, one);
}

// The per-tile stream: positions [0, n0) are the read indices rb .. rb + n0 (class 0; a general read in the range is
// skipped there), positions [n0, total) the tile's general descriptors gb ...
struct Stream { int rb, n0, gb, total; };

// What is carried of a read from the arrival of its columns to its processing.
//   misc: mapq | gen flags << 8 | kind << 16 (kind 0 nothing to do, 1 class 0, 2 general; two bits)
struct Rd {
  uint32_t a;        // class 0: info (lead | alen << 10 | trail << 21); general: aligned length | leading clip << 16
  int32_t pos;
  uint32_t nm;       // general: 0xFFFF = absent
  uint32_t misc;     // ... | bits 24-31: bits 32-39 of the general read's CIGAR offset
  uint32_t l_nc;     // l_seq | n_cigar << 16
  uint32_t co_lo;    // general: element offset of its CIGAR, low word
};
constexpr uint32_t kKindC0 = 1u << 16, kKindGen = 2u << 16;

template <int LB, bool BQ0>
__global__ __launch_bounds__(kPileupBlock, 4) void pileup_direct_kernel(DirectParams p) {
  constexpr int TILE = kTileSites;
  constexpr int NWAVES = kPileupBlock / 64;
  constexpr int OUT_IT = TILE / kPileupBlock;
  __shared__ __attribute__((aligned(16))) uint32_t lds[4 * TILE];
  __shared__ __attribute__((aligned(16))) uint32_t s_mhi[33 * 8];   // [h][w]: 0xFF in the bytes of the bases j >= h
  __shared__ __attribute__((aligned(16))) uint32_t s_mlo[33 * 8];   // [l][w]: 0xFF in the bytes of the bases j <  l
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
    if (tid < MIDAS_STATS) s_stats[tid] = 0ull;
  }

  const int lpr = p.lanes_per_read;
  const int rpw = p.reads_per_wave;
  const int g = lane / lpr;
  const int c = lane - g * lpr;
  const bool lane_used = g < rpw;
  const int q0 = c * LB;                           // first base of the lane in the read's stored query
  const int vstep = NWAVES * rpw;
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds;
  // v_perm_b32 tables (slots 0, 1, 3, 7 = A, C, G, T): threshold bytes and counter offsets
  const uint32_t thr = BQ0 ? 0u : (uint32_t)(p.baseq > 256 ? 255 : p.baseq - 1);
  const uint32_t th_lo = thr | (thr << 8) | 0x00FF0000u | (thr << 24), th_hi = 0x00FFFFFFu | (thr << 24);
  const uint32_t cd_lo = 0x08000400u, cd_hi = 0x0C000000u;
  const int rq = p.readq < 0 ? 0 : (p.readq > 256 ? 256 : p.readq);   // sum(q) < rq * l  <=>  np.mean(q) < readq (q <= 255)

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
    s.total = s.n0 + (int)(g1 - g0);
    return s;
  };

  // ---- stage F: the per-read columns of stream position v (raw loads; nothing is computed from them here) -------------
  struct Raw { uint32_t r0, r1, r2, r3, r4, r5, r6, r7, r8; int kind; };    // kind: 0 none, 1 range position, 2 descriptor
  auto fetch_raw = [&](const Stream& st, int v) -> Raw {
    Raw f;
    f.kind = (lane_used && v < st.total) ? (v < st.n0 ? 1 : 2) : 0;
    if (f.kind == 1) {
      const size_t i = (size_t)(st.rb + v);
      f.r0 = p.info[i];
      f.r1 = (uint32_t)p.pos[i];
      f.r2 = (uint32_t)p.nm[i];
      f.r3 = p.mapq[i];
      const unsigned long long so = (unsigned long long)p.seq_off[i], qo = (unsigned long long)p.qual_off[i];
      f.r4 = (uint32_t)so; f.r5 = (uint32_t)(so >> 32);
      f.r6 = (uint32_t)qo; f.r7 = (uint32_t)(qo >> 32);
      f.r8 = 0u;
    } else if (f.kind == 2) {
      const uint4* gd = reinterpret_cast<const uint4*>(p.gdesc) + (size_t)(st.gb + (v - st.n0)) * 3;
      const uint4 a = gd[0], b = gd[1];
      f.r0 = a.x; f.r1 = a.y; f.r2 = a.z; f.r3 = a.w;
      f.r4 = b.x; f.r5 = b.y; f.r6 = b.z; f.r7 = b.w;
      f.r8 = gd[2].x;
    }
    return f;
  };
  // ---- stage D: the lane's bases (two 16-byte loads of QUAL, one of SEQ; a general read's first four CIGAR ops) ---------
  struct Dat { uint32_t q[8]; uint32_t s[4]; };
  auto settle = [&](const Raw& f, Rd& r, Dat& d) {
    // the columns have arrived: fold them into what the read's processing needs, and issue the loads of its bases
    r.a = 0u; r.pos = 0; r.nm = 0u; r.misc = 0u; r.l_nc = 0u; r.co_lo = 0u;
    unsigned long long so = 0, qo = 0;
    int l = 0;
    if (f.kind == 1) {
      if (!(f.r0 & kInfoGeneral)) {
        r.a = f.r0; r.pos = (int32_t)f.r1; r.nm = f.r2; r.misc = (f.r3 & 0xFFu) | kKindC0;
        l = (int)((f.r0 & 1023u) + ((f.r0 >> kInfoAlenShift) & 2047u) + (f.r0 >> kInfoTrailShift));
        r.l_nc = (uint32_t)l;
        so = (unsigned long long)f.r4 | ((unsigned long long)f.r5 << 32);
        qo = (unsigned long long)f.r6 | ((unsigned long long)f.r7 << 32);
      }
    } else if (f.kind == 2) {
      r.pos = (int32_t)f.r1; r.l_nc = f.r2; r.nm = f.r3 & 0xFFFFu;
      r.misc = ((f.r3 >> 16) & 0xFFu) | (((f.r3 >> 24) & 0xFFu) << 8) | kKindGen | (((f.r8 >> 16) & 0xFFu) << 24);
      r.a = f.r4;
      l = (int)(f.r2 & 0xFFFFu);
      so = (unsigned long long)f.r5 | ((unsigned long long)(f.r8 & 0xFFu) << 32);
      qo = (unsigned long long)f.r6 | ((unsigned long long)((f.r8 >> 8) & 0xFFu) << 32);
      r.co_lo = f.r7;
    }
    if (r.misc != 0u && q0 < l) {
      const uint8_t* qp = p.qual + qo + (size_t)q0;
      const uint8_t* sp = p.seq4 + so + (size_t)(q0 >> 1);
      const u32x4_a1 qa = *reinterpret_cast<const u32x4_a1*>(qp);
      const u32x4_a1 qb = *reinterpret_cast<const u32x4_a1*>(qp + 16);
      const u32x4_a1 sv = *reinterpret_cast<const u32x4_a1*>(sp);
      d.q[0] = qa.x; d.q[1] = qa.y; d.q[2] = qa.z; d.q[3] = qa.w;
      d.q[4] = qb.x; d.q[5] = qb.y; d.q[6] = qb.z; d.q[7] = qb.w;
      d.s[0] = sv.x; d.s[1] = sv.y; d.s[2] = sv.z; d.s[3] = sv.w;
    }
  };

  Tile tile = load_tile(c_tiles, w);
  Stream st = load_stream(w);
  auto n_iters = [&](const Stream& s) -> int { return (s.total + rpw - 1) / rpw; };
  int it_hi = n_iters(st);
  int v0 = wave * rpw + g;
  Raw raw_n = fetch_raw(st, v0 + vstep);
  Rd rd_cur;
  Dat dat_cur;
  {
    const Raw raw_c = fetch_raw(st, v0);
    settle(raw_c, rd_cur, dat_cur);
  }
  __syncthreads();   // LDS zeroed, tables in place

  unsigned long long acc_cov = 0ull, acc_depth = 0ull;
  int t = w;
  for (;;) {
    const int tile_len = tile.len;
    const int tile_start = tile.start;
    uint32_t w_aligned = 0, w_mapped = 0;
    constexpr int REF_IT = TILE / (4 * kPileupBlock);
    uint32_t refw[REF_IT];
    if (p.out_allele) {
      const uint8_t* ref = p.ref + tile.site_base;
#pragma unroll
      for (int it = 0; it < REF_IT; ++it) {
        const int i = 4 * (tid + it * kPileupBlock);
        if (i + 4 <= tile_len) refw[it] = *reinterpret_cast<const u32_a1*>(ref + i);
      }
    }

    int vpos = v0;
    // the column / base prefetch runs across the tile boundary (as in pileup_tiles.hip): a wave's last two iterations fetch
    // the columns of its first two iterations of the NEXT tile
    const int n_w = it_hi > wave ? (it_hi - wave + NWAVES - 1) / NWAVES : 0;
    const bool xt = w_next < w_end && n_w >= 2;
    Stream xs = st;
    if (xt) xs = load_stream(w_next);
    for (int it = wave; it < it_hi; it += NWAVES, vpos += vstep) {
      Rd rd_n;
      Dat dat_n;
      settle(raw_n, rd_n, dat_n);       // (the columns of the next iteration have arrived: its bases are requested ...)
      if (xt && it + 2 * NWAVES >= it_hi) raw_n = fetch_raw(xs, (it + NWAVES < it_hi) ? v0 : v0 + vstep);   // ... then the
      else raw_n = fetch_raw(st, vpos + 2 * vstep);                                                          // columns after it

      // ================= process (rd_cur, dat_cur) =========================================================================
      const uint32_t kind = (rd_cur.misc >> 16) & 3u;
      const unsigned long long m_any = __ballot(kind != 0u);
      if (m_any != 0ull) {
        const bool is_gen = kind == 2u;
        const int l = (int)(rd_cur.l_nc & 0xFFFFu);
        const int pos = rd_cur.pos;
        const int nb = l - q0 < LB ? (l - q0 < 0 ? 0 : l - q0) : LB;     // bases of the read in this lane
        const bool has = kind != 0u && nb > 0;
        // ---- sum of the read's quality bytes (np.mean(aln.query_qualities), clipped bases included) -----------------
        uint32_t part = 0;
        if (has) {
          if (nb == LB) {
#pragma unroll
            for (int k = 0; k < 7; ++k) part = __builtin_amdgcn_sad_u8(dat_cur.q[k], 0u, part);
            part = __builtin_amdgcn_sad_u8(LB == 32 ? dat_cur.q[7] : (dat_cur.q[7] & 0x0000FFFFu), 0u, part);
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) part = __builtin_amdgcn_sad_u8(dat_cur.q[k] & low_bytes_mask(nb - 4 * k), 0u, part);
          }
          if (c == 0) part |= ((dat_cur.q[0] & 0xFFu) == 0xFFu) ? 0x80000000u : 0u;   // QUAL absent (BAM: first byte 0xFF)
        }
        uint32_t qsum = 0;
        for (int cc = 0; cc < lpr; ++cc) qsum += __shfl(part, g * lpr + cc);
        const bool t_noqual = (qsum >> 31) != 0u;
        qsum &= 0x7FFFFFFFu;

        // ---- keep_read (midas/run/snps.py:141-162), every test evaluated, the reference's order decides -------------------
        int align_len, lead;
        if (!is_gen) {
          lead = (int)(rd_cur.a & 1023u);
          align_len = (int)((rd_cur.a >> kInfoAlenShift) & 2047u);
        } else {
          align_len = (int)(rd_cur.a & 0xFFFFu);
          lead = (int)(rd_cur.a >> 16);
        }
        const uint32_t gflags = (rd_cur.misc >> 8) & 0xFFu;
        const int mapq = (int)(rd_cur.misc & 0xFFu);
        const int min_match = s_tables[align_len < p.table_len ? align_len : 0];
        const int min_align = s_tables[p.table_len + (l < p.table_len ? l : 0)];
        const bool t_noseq = l == 0;
        const bool t_nonm = is_gen && (gflags & kGenNoNm) != 0u;
        const bool t_zero = align_len == 0;
        const bool t_pid = align_len - (int)rd_cur.nm < min_match;                                  // pid < mapid
        const bool t_drop = ((int)qsum < rq * l) | (mapq < p.mapq_min) | (align_len < min_align);   // readq, mapq, aln_cov
        const bool t_over = is_gen && (gflags & kGenOverrun) != 0u;
        uint32_t err = t_over ? (uint32_t)E_CIGAR_OVERRUN : 0u;
        err = t_drop ? 0u : err;
        err = t_noqual ? (uint32_t)E_NO_QUAL : err;
        err = t_pid ? 0u : err;
        err = t_zero ? (uint32_t)E_ZERO_ALIGN : err;
        err = t_nonm ? (uint32_t)E_NO_NM : err;
        err = t_noseq ? (uint32_t)E_NO_SEQ : err;
        err = kind != 0u ? err : 0u;
        const bool keep = kind != 0u && !(t_noseq | t_nonm | t_zero | t_pid | t_noqual | t_drop | t_over);

        // owner tile of a read = the tile holding its (clamped) start: it alone counts the read in the stats
        int cpos = pos < 0 ? 0 : pos;
        cpos = cpos > tile.contig_len - 1 ? tile.contig_len - 1 : cpos;
        const bool owner = kind != 0u && cpos >= tile_start && cpos < tile_start + tile_len;
        const int rel = pos - tile_start;                                  // pos >= -2^31, tile_start >= 0: may wrap for
        int rrel = (rel > (1 << 25) || rel < -(1 << 30)) ? (1 << 25) : rel;   // absurd positions -> parked far right

        // ---- per-base threshold bytes and counter offsets from the 4-bit codes ------------------------------------------
        uint32_t th[8], cd[8];
        bool walking = keep && has;
        if (walking) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const uint32_t x = dat_cur.s[s];
            const uint32_t se = (((x >> 4) & 0x0F0F0F0Fu) + 0x07070707u) ^ 0x08080808u;   // even bases (high nibbles)
            const uint32_t so = ((x & 0x0F0F0F0Fu) + 0x07070707u) ^ 0x08080808u;          // odd bases
            th[2 * s] = __builtin_amdgcn_perm(th_hi, th_lo, se);
            th[2 * s + 1] = __builtin_amdgcn_perm(th_hi, th_lo, so);
            cd[2 * s] = __builtin_amdgcn_perm(cd_hi, cd_lo, se);
            cd[2 * s + 1] = __builtin_amdgcn_perm(cd_hi, cd_lo, so);
          }
        }
        uint32_t qv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) qv[k] = BQ0 ? 0x01010101u : dat_cur.q[k];

        // ---- the read's match segments, one at a time: bases [jlo, jhi) of the lane, the first of them at site loc0 --------
        int k = 0, qpos = 0, jlo = 0, jhi = 0, loc0 = 0;
        const int nc = (int)(rd_cur.l_nc >> 16);
        // a general read's CIGAR is not prefetched (it would cost eight registers of the double-buffered bases): only the
        // iterations at the end of a tile's stream come here
        const uint32_t* cig = p.cigar + ((size_t)rd_cur.co_lo | ((size_t)(rd_cur.misc >> 24) << 32));
        uint32_t cg0 = 0u, cg1 = 0u, cg2 = 0u, cg3 = 0u;
        if (is_gen && nc > 0) {
          const u32x4_a4 cv = *reinterpret_cast<const u32x4_a4*>(cig);   // (may overhang into the array's slack)
          cg0 = cv.x; cg1 = cv.y; cg2 = cv.z; cg3 = cv.w;
        }
        const int q1 = q0 + nb;
        auto next_segment = [&]() -> bool {
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
            if (m || op == OP_I || op == OP_S) { qpos += len; qpos = qpos > (1 << 29) ? (1 << 29) : qpos; }
            if (m || op == OP_D || op == OP_N) { rrel += len; rrel = rrel > (1 << 29) ? (1 << 29) : rrel; }
            if (found) return true;   // H, P and anything else: no effect
          }
          return false;
        };
        if (walking) {
          if (!is_gen) {       // class 0: the one segment is query [lead, lead + align_len) at sites pos ...
            const int lo = lead > q0 ? lead : q0;
            const int hi = lead + align_len < q1 ? lead + align_len : q1;
            jlo = lo - q0; jhi = hi - q0; loc0 = rrel + (q0 - lead);
            walking = lo < hi;
          } else {
            walking = next_segment();
          }
        }
        const uint32_t one = 1u;
        while (__ballot(walking) != 0ull) {
          // bases of the lane that belong to this segment AND lie inside the tile: [lo, hi)
          const int lo = jlo > -loc0 ? jlo : -loc0;
          const int hi = jhi < tile_len - loc0 ? jhi : tile_len - loc0;
          const bool go = walking && lo < hi;
          const uint32_t abase = ((uint32_t)loc0 << 4) + lds_base;
          if (__ballot(go && (lo > 0 || hi < LB)) == 0ull) {
            if (go) tally_lane<LB>(qv, th, cd, abase, one);
          } else if (go) {
            // partial lanes: 0xFF into the threshold bytes outside [lo, hi) (a row of each table, LDS)
            const uint4* mh = reinterpret_cast<const uint4*>(s_mhi) + 2 * (hi > 32 ? 32 : hi);
            const uint4* ml = reinterpret_cast<const uint4*>(s_mlo) + 2 * (lo < 0 ? 0 : lo);
            const uint4 h0 = mh[0], h1 = mh[1], l0 = ml[0], l1 = ml[1];
            uint32_t tm[8];
            tm[0] = th[0] | h0.x | l0.x; tm[1] = th[1] | h0.y | l0.y; tm[2] = th[2] | h0.z | l0.z; tm[3] = th[3] | h0.w | l0.w;
            tm[4] = th[4] | h1.x | l1.x; tm[5] = th[5] | h1.y | l1.y; tm[6] = th[6] | h1.z | l1.z; tm[7] = th[7] | h1.w | l1.w;
            tally_lane<LB>(qv, tm, cd, abase, one);
          }
          walking = (walking && is_gen && k < nc) ? next_segment() : false;
        }

        // ---- per-species read counters: one ballot per wave ---------------------------------------------------------------
        const bool head = owner && c == 0;
        w_aligned += (uint32_t)__popcll(__ballot(head));
        w_mapped += (uint32_t)__popcll(__ballot(head && keep));
        if (head && err) {   // (the read's index: its stream position, or the first word of its descriptor)
          const uint32_t idx = is_gen ? p.gdesc[(size_t)(st.gb + (vpos - st.n0)) * kGenDescWords] : (uint32_t)(st.rb + vpos);
          atomicMin(p.err, ((unsigned long long)idx << 8) | err);
        }
      }

      rd_cur = rd_n;
      dat_cur = dat_n;
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
    const int nit_hi = n_iters(nst);
    Raw raw_c;
    if (more && !xt) {   // (with xt the pipeline already holds the next tile's first two iterations)
      raw_c = fetch_raw(nst, v0);
      raw_n = fetch_raw(nst, v0 + vstep);
    }
    lds_barrier();       // every tally of this tile is in LDS
    if (more && !xt) settle(raw_c, rd_cur, dat_cur);
    uint32_t ticket = 0;
    if (dynamic && more && tid == 0) ticket = atomicAdd(&sched[32 * sched_group], 1u);

    // ---- emit the tile: counts[site][A,C,G,T] (and re-zero LDS), covered / total-depth partials ---------------------------
    {
      uint4* out = reinterpret_cast<uint4*>(p.out_counts) + tile.site_base;
      uint4* lds4 = reinterpret_cast<uint4*>(lds);
#pragma unroll
      for (int it = 0; it < OUT_IT; ++it) {
        const int i = tid + it * kPileupBlock;
        if (i < tile_len) {
          const uint4 v = lds4[i];
          lds4[i] = make_uint4(0u, 0u, 0u, 0u);
          u32x4_a8 nv; nv.x = v.x; nv.y = v.y; nv.z = v.z; nv.w = v.w;
          __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_a8*>(out + i));
          const uint32_t d = v.x + v.y + v.z + v.w;
          acc_cov += d > 0u ? 1ull : 0ull;
          acc_depth += d;
        }
      }
    }
    if (p.out_allele) {
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
        if (acc_cov) atomicAdd(&s_stats[MIDAS_STAT_COVERED], acc_cov);
        if (acc_depth) atomicAdd(&s_stats[MIDAS_STAT_DEPTH], acc_depth);
      }
      acc_cov = 0ull;
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
    it_hi = nit_hi;
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
