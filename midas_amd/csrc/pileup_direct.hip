// gfx950 (CDNA4) pileup kernel of the MIDAS SNP path that reads the BAM's own bytes: per read ONE 16-byte record (pos, l_seq,
// n_cigar, NM, mapq, payload offset -- layout.h DirectRec) and its CIGAR / 4-bit SEQ / QUAL bytes as BAM lays them out, one
// run per read -- nothing decoded, sorted or decided beforehand, ONE visit per read.  Integer counting: no MFMA.
//
// Reference semantics implemented here (citations into /root/reference):
//   keep_read                       midas/run/snps.py:141-162  (query_alignment_sequence :145, np.mean(query_qualities) :151)
//   count_coverage call site        midas/run/snps.py:194-199  ([EXT] pysam: get_aligned_pairs(matches_only), qual >= quality_threshold,
//                                   only 'A','C','G','T' counted)
//   depth / covered / total_depth   midas/run/snps.py:204-213
//   str(rec.seq).upper()            midas/run/snps.py:62
// The reference filters and counts a read in one pass (keep_read inside count_coverage's iterator); so does this kernel.
//
// Work decomposition: as in pileup_tiles.hip -- tiles of <= 2048 sites, a persistent 256-thread workgroup per item (four per CU), tallies
// in LDS as [site][A,C,G,T] u32, one coalesced write-out per tile, items handed out by per-XCD counters.
//
// Stream.  The ranges pass (index_direct.hip, 4 bytes per read) left, per tile, the run [tbegin, tend) of read indices that
// can touch the tile -- the input is position-sorted, so that is a contiguous run of the read arrays.  It is dealt to the
// workgroup's waves as wave-iterations of floor(64 / lanes per read) reads.
//
// Lane mapping.  A lane owns LB (30 or 32) consecutive bases of a read's STORED query: two 16-byte loads of QUAL, one of
// 4-bit SEQ (LB is even, so a lane's bases start on a byte).  A read of l_seq bases takes ceil(l_seq / LB) adjacent lanes
// (5 for 150 bp).  Loads are issued two iterations (the read's record: one dwordx4) and one iteration (bases + the first four
// CIGAR ops: four dwordx4 off ONE scalar base per wave-iteration -- the payload of the iteration's first read -- plus a 32-bit
// lane offset) ahead of their use; none of them sits in a branch.
//
// Per read.  The lanes of a read decide in registers what its CIGAR is: ONE or TWO gap-free match runs (direct_common.h
// ReadShape: clips, at most one insertion / deletion / skip -- what an aligner writes for nearly every read) are tallied
// straight from the shape, the lane that holds the indel in a second masked pass; anything else (several indels, pads, odd
// clips, no NM / SEQ, a start off the contig) is walked op by op where it lies, the slow path.
//
// Per base, from the raw bytes:  the 4-bit codes of eight bases (one dword) are split into their even and odd nibbles
// (two masks), mapped to v_perm_b32 selectors by `(n + 7) ^ 8` -- A, C, G, T (1, 2, 4, 8) land on table slots 0, 1, 3, 7,
// every other code on a slot or a selector constant that yields 0xFF -- and looked up twice: a THRESHOLD byte (baseq - 1
// for A/C/G/T, 0xFF for anything else) and the byte offset of the base's counter.  Then per base one SDWA compare
// `qual.byte > threshold.byte` into a lane mask (the byte selects of the two operands are independent, so the even / odd
// order of the looked-up bytes costs nothing), one SDWA OR forming the LDS address, one returnless ds_add under the mask.
// Clipping (soft clips, segment borders, tile edges, the read's tail) zeroes the 4-bit codes of the bases outside (code 0 is
// no base: its threshold byte is 0xFF): two rows of a nibble-mask table from LDS per partial pass.
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

// (developer listings: hipcc -S -DMIDAS_ISA_MARKS puts `; MARK name` lines into the assembly, tools/isa/segments.py counts between them)
#ifdef MIDAS_ISA_MARKS
#define MIDAS_MARK(name) asm volatile("; MARK " name)
#else
#define MIDAS_MARK(name) do { } while (0)
#endif

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

// The per-tile stream: the reads rb .. rb + n0 of the read arrays, as `total` wave-iterations.
struct Stream { int rb, n0, total; };

// Workgroup shape (developer sweeps: tools/build_variant.sh x -DMIDAS_DIRECT_BLOCK=384).
#ifndef MIDAS_DIRECT_BLOCK
#define MIDAS_DIRECT_BLOCK 256
#endif
constexpr int kDirectBlock = MIDAS_DIRECT_BLOCK;
static_assert(kDirectBlock % 64 == 0 && kDirectBlock >= 128 && kDirectBlock <= 1024, "whole wavefronts");
constexpr int kDirectWavesPerSimd = (kWorkgroupsPerCU * kDirectBlock / 64 + 3) / 4;      // four workgroups per CU


template <int LB, bool BQ0, int OV = kDirectOverhang>
__global__ __launch_bounds__(kDirectBlock, kDirectWavesPerSimd) void pileup_direct_kernel(DirectParams p) {
  constexpr int TILE = kTileSites;
  constexpr int NWAVES = kDirectBlock / 64;
  constexpr int OUT_IT = (TILE + kDirectBlock - 1) / kDirectBlock;
  // OV: sites behind the tile's last that the tallies also hold (see "chunks" below); kDirectOverhangLong for batches of longer reads
  static_assert(OV <= TILE, "the overhang is moved by the write-out's first rounds");
  __shared__ __attribute__((aligned(16))) uint32_t lds[4 * (TILE + OV)];
  __shared__ __attribute__((aligned(16))) uint32_t s_khi[33 * 4];   // [h][w]: 0xF in the nibbles of the bases j <  h of a lane's four SEQ words
  __shared__ __attribute__((aligned(16))) uint32_t s_klo[33 * 4];   // [l][w]: 0xF in the nibbles of the bases j >= l
  __shared__ uint32_t s_qsum[NWAVES * 64];                           // per wave and read slot: sum of a read's quality bytes
  __shared__ unsigned long long s_stats[MIDAS_STATS];
  __shared__ uint32_t s_next_ticket;
  // [min_match table_len][min_align table_len]; 16-bit entries in the long-overhang instantiation (its reads are <= 288 bases, the
  // thresholds at most that): the kilobyte this saves is what lets FOUR workgroups of it share a CU's 160 KiB
  using table_t = typename std::conditional<(OV > kDirectOverhang), int16_t, int32_t>::type;
  extern __shared__ __attribute__((aligned(16))) uint8_t s_tables_raw[];
  table_t* const s_tables = reinterpret_cast<table_t*>(s_tables_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Work items.  The tiles [0, n_chunked_tiles) are dealt in CHUNKS of chunk_tiles consecutive tiles, the rest one by one (the
  // chip's last tens of microseconds need the fine grain).  Inside a chunk a read is visited ONCE, by the tile it starts in:
  // what it adds behind that tile's last site lands in the OV sites the tallies hold beyond the tile and moves to the front
  // when the workgroup goes on to the next tile (same contig).  Only a chunk's first tile streams the reads that reach in
  // from the tile before it, as every tile did before (7.5 % of the reads seen twice at 2048 sites and 150 bp; a quarter of
  // that with chunks of four).  The host asks for chunks when the reads are position-sorted; a read that spans more than OV sites
  // (a long deletion, an N skip) makes the chunks it touches fall back to tile-by-tile (chunk_ok), not the batch.
  const int K = p.chunk_tiles > 1 ? p.chunk_tiles : 1;
  const int T4 = K > 1 ? p.n_chunked_tiles : 0;
  const int n4 = T4 / K;
  const int w_end = n4 + (p.n_tiles - T4);              // items
  auto item_first = [&](int i) -> int { return i < n4 ? i * K : T4 + (i - n4); };
  auto item_end = [&](int i) -> int { return i < n4 ? i * K + K : T4 + (i - n4) + 1; };
#ifdef MIDAS_DIRECT_STATIC
  const bool dynamic = false;       // (developer variant: tiles dealt round robin)
#else
  const bool dynamic = (gridDim.x % kSchedGroups) == 0;
#endif
  const int sched_group = (int)(blockIdx.x % kSchedGroups);
  uint32_t* const sched = p.sched;
  if ((int)blockIdx.x >= w_end) return;
  int c_first = item_first((int)blockIdx.x), c_end = item_end((int)blockIdx.x);      // the chunk in work: tiles [c_first, c_end)
  int i_next = (int)blockIdx.x + (int)gridDim.x;                                      // the item after it (>= w_end: none)
  i_next = i_next < w_end ? i_next : w_end;
#if MIDAS_SNPS_DEBUG_BITS & 256
  unsigned long long pr_cols = 0, pr_bases = 0, pr_work = 0, pr_sync = 0, pr_out = 0, pr_iters = 0;
  const unsigned long long pr_t0 = __builtin_readcyclecounter();
#define PROBE_NOW() __builtin_readcyclecounter()
#else
#define PROBE_NOW() 0ull
#endif

  {
    uint4* z = reinterpret_cast<uint4*>(lds);
    for (int i = tid; i < TILE + OV; i += kDirectBlock) z[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < p.table_len; i += kDirectBlock) {
      s_tables[i] = (table_t)p.filt->min_match[i];
      s_tables[p.table_len + i] = (table_t)p.filt->min_align[i];
    }
    for (int i = tid; i < 33 * 4; i += kDirectBlock) {
      const int h = i >> 2, wd = i & 3;
      uint32_t kh = 0, kl = 0;
      for (int k = 0; k < 8; ++k) {
        const int j = 8 * wd + k;                              // base k of word wd: byte k / 2, the HIGH nibble when k is even
        const uint32_t nib = 0xFu << (8 * (k >> 1) + ((k & 1) ? 0 : 4));
        if (j < h) kh |= nib;
        if (j >= h) kl |= nib;
      }
      s_khi[i] = kh;
      s_klo[i] = kl;
    }
    s_qsum[tid] = 0u;
    if (tid < MIDAS_STATS) s_stats[tid] = 0ull;
  }

  const int lpr = p.lanes_per_read;
  const int rpw = p.reads_per_wave;
  const int g = lane / lpr;
  const int c = lane - g * lpr;
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
  // (cin: the tile continues a chunk in its contig -- the reads that reach in from the tile before were tallied by that tile)
  auto load_stream = [&](int tt, bool cin) -> Stream {
    Stream s;
    const uint32_t b = cin ? c_te[tt - 1] : c_tb[tt], e = c_te[tt];
    s.rb = e > b ? (int)b : 0;
    s.n0 = e > b ? (int)(e - b) : 0;
    s.total = (s.n0 + rpw - 1) / rpw;
    return s;
  };
  // reads of a stream's wave-iteration `it` (wave-uniform): the lanes with g below it hold one
  auto reads_in = [&](const Stream& st, int it) -> int {
    const long long left = (long long)st.n0 - (long long)it * rpw;
    return left <= 0 ? 0 : (left < rpw ? (int)left : rpw);
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
  // (sparse: a pass with few lanes, e.g. the one lane of a read that holds its indel -- a group of eight bases none of the
  // wave's lanes has a base in is skipped)
  auto tally_range = [&](bool go, int lo, int hi, int loc0, const uint32_t (&qv)[8], const uint32_t (&sq)[4], auto sparse_tag) {
    constexpr bool SPARSE = decltype(sparse_tag)::value;
    const uint32_t abase = ((uint32_t)loc0 << 4) + lds_base;
    const bool masked = !(kDebug & 32) && __ballot(go && (lo > 0 || hi < LB)) != 0ull;   // partial lanes: the codes outside become 0
    unsigned long long gmask[4];
    if (SPARSE) {
#pragma unroll
      for (int S = 0; S < 4; ++S) gmask[S] = __ballot(go && lo < 8 * S + 8 && hi > 8 * S);
    }
    if (!go) return;                                                    // outside [lo, hi) (a row of each table, LDS)
    uint4 keep = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    if (masked) {
      const uint4 kh = *reinterpret_cast<const uint4*>(s_khi + 4 * (hi > 32 ? 32 : hi));
      const uint4 kl = *reinterpret_cast<const uint4*>(s_klo + 4 * (lo < 0 ? 0 : lo));
      keep.x = kh.x & kl.x; keep.y = kh.y & kl.y; keep.z = kh.z & kl.z; keep.w = kh.w & kl.w;
    }
    auto group = [&](auto sidx, auto off, auto nbases) {
      constexpr int S = decltype(sidx)::value;
      if (SPARSE && gmask[S] == 0ull) return;
      const uint32_t x = masked ? (sq[S] & (S == 0 ? keep.x : (S == 1 ? keep.y : (S == 2 ? keep.z : keep.w)))) : sq[S];
      const uint32_t se = (((x >> 4) & 0x0F0F0F0Fu) + 0x07070707u) ^ 0x08080808u;   // even bases (high nibbles)
      const uint32_t so = ((x & 0x0F0F0F0Fu) + 0x07070707u) ^ 0x08080808u;          // odd bases
      const uint32_t te = __builtin_amdgcn_perm(th_hi, th_lo, se), to = __builtin_amdgcn_perm(th_hi, th_lo, so);
      const uint32_t ce = __builtin_amdgcn_perm(cd_hi, cd_lo, se), co = __builtin_amdgcn_perm(cd_hi, cd_lo, so);
      tally_group<decltype(off)::value, decltype(nbases)::value>(qv[2 * S], qv[2 * S + 1], te, to, ce, co, abase, one);
    };
    using std::integral_constant;
    group(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, 8>{});
    group(integral_constant<int, 1>{}, integral_constant<int, 128>{}, integral_constant<int, 8>{});
    group(integral_constant<int, 2>{}, integral_constant<int, 256>{}, integral_constant<int, 8>{});
    group(integral_constant<int, 3>{}, integral_constant<int, 384>{}, integral_constant<int, (LB == 32 ? 8 : 6)>{});
  };

  // ---- stage F: the record of this lane's read in wave-iteration `it` of a tile's stream: ONE dwordx4.  Raw loads: nothing is
  // computed from them here, and NO load sits in a branch -- a lane without a read fetches read 0's record, a lane without
  // bases the first bytes of the iteration's payload -- so that the compiler can count the loads in flight (a load in a branch
  // makes it wait for every outstanding load, vmcnt(0), before the first use of any of them: the prefetch of the next
  // iteration's bases would be waited for at once).
  struct Raw { uint32_t pos, l_nc, nmq, off; };
  const uint4* const recs = reinterpret_cast<const uint4*>(p.rec);
  auto fetch = [&](const Stream& st, int it) -> Raw {
    const uint32_t v = (uint32_t)(it * rpw) + (uint32_t)g;       // (a stream holds fewer than 2^31 reads, it * rpw <= n0 + 63)
    const uint32_t r = v < (uint32_t)st.n0 ? (uint32_t)st.rb + v : 0u;
    const uint4 x = recs[r];
    Raw f;
    f.pos = x.x; f.l_nc = x.y; f.nmq = x.z; f.off = x.w;
    return f;
  };
  // ---- stage D: what the read's processing needs of its record, and the lane's bases (two 16-byte loads of QUAL, one of
  // SEQ) + the read's first four CIGAR ops (one 16-byte load; the payload has slack behind its last read).  The four loads go
  // off ONE scalar base -- the payload of the iteration's first read (lane 0's) -- plus a 32-bit lane offset: the reads of an
  // iteration are neighbours in the payload.
  //   nmq: NM (16 bits, 0xFFFF = no NM tag) | mapq << 16 | kGenIdle << 24 (a lane without a read)      l_nc: l_seq | n_cigar << 16
  struct Rd { uint32_t pos, nmq, l_nc; };
  struct Dat { uint32_t q[8]; uint32_t s[4]; uint32_t cg[4]; };
  auto settle = [&](const Raw& f, int n_reads_it, Rd& r, Dat& d) {
    const bool act = g < n_reads_it;
    const uint32_t l_nc = act ? f.l_nc : 0u;
    r.pos = f.pos;
    r.l_nc = l_nc;
    r.nmq = act ? f.nmq : ((uint32_t)kGenIdle << 24);
    const uint32_t off0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)f.off);        // (lane 0: the iteration's first read, or read 0)
    const uint8_t* const base = p.payload + ((unsigned long long)off0 << 3);
    const uint32_t l = l_nc & 0xFFFFu, nc4 = (l_nc >> 16) << 2;
    const uint32_t rel = (f.off - off0) << 3;                          // (an iteration's reads lie within a few hundred KB)
    const bool has = (uint32_t)q0 < l && !(kDebug & 128);
    const uint32_t vc = act && !(kDebug & 128) ? rel : 0u;
    const uint32_t vs = has ? rel + nc4 + (uint32_t)(q0 >> 1) : 0u;
    const uint32_t vq = has ? rel + nc4 + ((l + 1u) >> 1) + (uint32_t)q0 : 0u;
    const u32x4_a1 qa = *reinterpret_cast<const u32x4_a1*>(base + (size_t)vq);
    const u32x4_a1 qb = *reinterpret_cast<const u32x4_a1*>(base + (size_t)vq + 16);
    const u32x4_a1 sv = *reinterpret_cast<const u32x4_a1*>(base + (size_t)vs);
    const u32x4_a4 cv = *reinterpret_cast<const u32x4_a4*>(base + (size_t)vc);
    d.q[0] = qa.x; d.q[1] = qa.y; d.q[2] = qa.z; d.q[3] = qa.w;
    d.q[4] = qb.x; d.q[5] = qb.y; d.q[6] = qb.z; d.q[7] = qb.w;
    d.s[0] = sv.x; d.s[1] = sv.y; d.s[2] = sv.z; d.s[3] = sv.w;
    d.cg[0] = cv.x; d.cg[1] = cv.y; d.cg[2] = cv.z; d.cg[3] = cv.w;
  };

  Tile tile = load_tile(c_tiles, c_first);
  Stream st = load_stream(c_first, false);
  // The pipeline of a wave, in two register sets that swap roles every iteration (the loop below is unrolled by two: a copy
  // at its back edge would have to WAIT for the loads it copies -- the next iteration's bases, requested a moment ago):
  //   set A / B   one holds the iteration being tallied, the other the next one's bases (in flight)
  //   raw X       the columns of the iteration after that (in flight); its bases are requested at the top of the next body
  // Between tiles the current iteration sits in set A.
  Raw rawX;
  int nrX = 0;                                  // reads of the iteration the columns belong to (wave-uniform)
  Rd rdA, rdB;
  Dat datA, datB;
  auto prime = [&](const Stream& s0) {
    const Raw r0 = fetch(s0, wave);
    rawX = fetch(s0, wave + NWAVES);
    nrX = reads_in(s0, wave + NWAVES);
    settle(r0, reads_in(s0, wave), rdA, datA);
  };
  prime(st);
  __syncthreads();   // LDS zeroed, tables in place

  uint32_t acc_cov = 0u;                 // (a thread's sites between two flushes: far below 2^32)
  unsigned long long acc_depth = 0ull;
  int t = c_first;
  for (;;) {
    const int tile_len = tile.len;
    const int tile_start = tile.start;
    // the tile after this one: the chunk's next, or the next item's first
    const bool in_chunk = t + 1 < c_end;
    const bool more = in_chunk || i_next < w_end;
    const int wn = in_chunk ? t + 1 : (more ? item_first(i_next) : t);
    // (a chunk that holds -- or is reached by -- a read spanning more than the overhang is piled up tile by tile: chunk_ok)
    const bool carry = in_chunk && (!p.chunk_ok || p.chunk_ok[c_first / K] != 0);
    const bool cout = carry && (int32_t)c_tiles[8 * (size_t)wn] == tile.contig;      // its first OV sites are tallied here
    // (the contig's last tile may be shorter than the overhang: nothing is tallied behind the contig's end)
    const int next_len = (int32_t)c_tiles[8 * (size_t)wn + 2];
    const int ext_len = tile_len + (cout ? (next_len < OV ? next_len : OV) : 0);
    const int it_hi = st.total;
    uint32_t w_aligned = 0, w_mapped = 0;
    constexpr int REF_IT = (TILE + 4 * kDirectBlock - 1) / (4 * kDirectBlock);

    // the column / base prefetch runs across the tile boundary (as in pileup_tiles.hip): a wave's last two iterations fetch
    // the columns of its first two iterations of the NEXT tile
    const int n_w = it_hi > wave ? (it_hi - wave + NWAVES - 1) / NWAVES : 0;
    const bool xt = more && n_w >= 2;
    Stream xs = st;
    if (xt) xs = load_stream(wn, cout);
    // one wave-iteration: `it` of the tile's stream, the wave's k_it-th; tallies (rd_cur, dat_cur), requests the bases of the
    // next iteration into (rd_n, dat_n) from the columns rawX and then the columns of the one after it into rawX
    auto iteration = [&](int it, int k_it, Rd& rd_cur, Dat& dat_cur, Rd& rd_n, Dat& dat_n) {
      const unsigned long long pt0 = PROBE_NOW();
      MIDAS_MARK("settle");
      settle(rawX, nrX, rd_n, dat_n);          // (the columns of the next iteration have arrived: its bases are requested ...)
      {                                        // ... then the columns of the one after it
        const int kf = k_it + 2;               // the wave's iteration (of this tile, or counted on into the next) to fetch for
        const bool over = xt && kf >= n_w;
        Stream fs;
        fs.rb = over ? xs.rb : st.rb; fs.n0 = over ? xs.n0 : st.n0; fs.total = over ? xs.total : st.total;
        const int fi = wave + (over ? kf - n_w : kf) * NWAVES;
        rawX = fetch(fs, fi);
        nrX = reads_in(fs, fi);
      }
#if MIDAS_SNPS_DEBUG_BITS & 256
      asm volatile("" :: "v"(rd_n.pos), "v"(rd_n.l_nc));
      const unsigned long long pt1 = PROBE_NOW();
      asm volatile("" :: "v"(dat_cur.q[0]), "v"(dat_cur.q[7]), "v"(dat_cur.s[3]), "v"(dat_cur.cg[3]));
      const unsigned long long pt2 = PROBE_NOW();
      pr_cols += pt1 - pt0; pr_bases += pt2 - pt1; pr_iters += 1;
#endif

      if (kDebug & 4) {          // (developer timing variant: the stream of loads only)
        asm volatile("" :: "v"(dat_cur.q[0]), "v"(dat_cur.q[7]), "v"(dat_cur.s[0]), "v"(dat_cur.s[3]), "v"(dat_cur.cg[0]), "v"(rd_cur.pos));
        return;
      }
      MIDAS_MARK("qsum");
      const int pos = (int)rd_cur.pos;
      const int l = (int)(rd_cur.l_nc & 0xFFFFu);
      const uint32_t nc = rd_cur.l_nc >> 16;
      const int nb = l - q0 < LB ? (l - q0 < 0 ? 0 : l - q0) : LB;     // bases of the read in this lane
      const bool has = nb > 0;
      // (developer timing variant, bit 512: ALL per-read work off -- no quality sum, no CIGAR shape, no filter tables: every read is
      // one match run of its length and kept; the tallies and the stream stay.  What a scheme that does the per-read work once
      // per read instead of on each of its lanes could save AT MOST: profiles/r06_kernel_experiments.txt section 3)
      uint32_t qsum = (kDebug & (64 | 512)) ? 0x00FFFFFFu : read_sum(has ? lane_qsum(dat_cur.q, nb) : 0u);
      const bool t_noqual = (qsum >> 31) != 0u;
      qsum &= 0x7FFFFFFFu;
      const uint32_t nm16 = rd_cur.nmq & 0xFFFFu;
      const int nm = (int)nm16, mapq = (int)((rd_cur.nmq >> 16) & 0xFFu);
      const bool act = !((rd_cur.nmq >> 24) & kGenIdle);
      uint32_t qv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) qv[k] = BQ0 ? 0x01010101u : dat_cur.q[k];

      // ---- the read's shape, in registers: one match op of the read's length settles most reads; anything else takes the
      // four-op grammar (wave-uniform branch) ------------------------------------------------------------------------------
      MIDAS_MARK("shape");
      ReadShape sh;
      sh.lead = 0u; sh.m1 = (uint32_t)l; sh.ins = 0u; sh.del = 0u; sh.alen = (uint32_t)l;
      bool shaped = nc == 1u && op_is_match(dat_cur.cg[0] & 15u) && (dat_cur.cg[0] >> 4) == (uint32_t)l && l >= 1;
      if (kDebug & 512) shaped = true;
      if (__ballot(act && !shaped) != 0ull) {
        ReadShape s2;
        const bool ok = decode_shape(dat_cur.cg[0], dat_cur.cg[1], dat_cur.cg[2], dat_cur.cg[3], nc, (uint32_t)l, &s2);
        if (!shaped) { sh = s2; shaped = ok; }
      }
      // one or two match runs, NM present, the start inside the contig: the fast path.  Everything else is walked.
      const bool fast = act && shaped && nm16 != 0xFFFFu && pos >= 0 && pos < tile.contig_len;
      const bool slow = act && !fast;

      // ======================= fast: one or two gap-free match runs ==========================================================
      MIDAS_MARK("filter");
      const int lead = (int)sh.lead, align_len = (int)sh.alen;
      bool keep, owner;
      uint32_t err;
      {
        // ---- keep_read (midas/run/snps.py:141-162): such a read has SEQ, NM and a non-empty aligned part -----------------------
        const int min_match = (kDebug & 512) ? 0 : s_tables[align_len < p.table_len ? align_len : 0];
        const int min_align = (kDebug & 512) ? 0 : s_tables[p.table_len + (l < p.table_len ? l : 0)];
        const bool t_pid = align_len - nm < min_match;                                                       // pid < mapid
        const bool t_drop = ((int)qsum < rq * l) | (mapq < p.mapq_min) | (align_len < min_align);             // readq, mapq, aln_cov
        err = (fast && !t_pid && t_noqual) ? (uint32_t)E_NO_QUAL : 0u;
        keep = fast && !(t_pid | t_noqual | t_drop);
        const int rel = pos - tile_start;            // 0 <= pos < contig length: no wrap
        owner = fast && rel >= 0 && rel < tile_len;
        // run A: query [lead, lead + m1) at sites pos ...; run B: query [lead + m1 + ins, lead + alen) at pos + m1 + del ...
        // -- this lane's part of each, clipped to the tile
        const int qa1 = lead + (int)sh.m1, qb0 = qa1 + (int)sh.ins, qb1 = lead + align_len;
        const int loc_a = rel + (q0 - lead);
        const int loc_b = loc_a + (int)sh.del - (int)sh.ins;
        int lo_a = lead - q0, hi_a = qa1 - q0, lo_b = qb0 - q0, hi_b = qb1 - q0;
        lo_a = lo_a > -loc_a ? lo_a : -loc_a;
        lo_a = lo_a > 0 ? lo_a : 0;
        hi_a = hi_a < ext_len - loc_a ? hi_a : ext_len - loc_a;
        hi_a = hi_a < nb ? hi_a : nb;
        lo_b = lo_b > -loc_b ? lo_b : -loc_b;
        lo_b = lo_b > 0 ? lo_b : 0;
        hi_b = hi_b < ext_len - loc_b ? hi_b : ext_len - loc_b;
        hi_b = hi_b < nb ? hi_b : nb;
        const bool go_a = keep && lo_a < hi_a, go_b = keep && lo_b < hi_b;
        // first pass: every lane its run (the lane that holds the indel: the part in front of it); second pass, only when a
        // lane of the wave has bases on both sides of an indel: the part behind it
        const bool go1 = go_a | go_b;
        MIDAS_MARK("pass1");
        if (!(kDebug & 1) && __ballot(go1) != 0ull)
          tally_range(go1, go_a ? lo_a : lo_b, go_a ? hi_a : hi_b, go_a ? loc_a : loc_b, qv, dat_cur.s, std::false_type{});
        const bool go2 = go_a & go_b;
        MIDAS_MARK("pass2");
        if (!(kDebug & (1 | 16)) && __ballot(go2) != 0ull) tally_range(go2, lo_b, hi_b, loc_b, qv, dat_cur.s, std::true_type{});
      }
      // ======================= slow: walked op by op ============================================================================
      MIDAS_MARK("slowgate");
      if (__ballot(slow) != 0ull) {
        const uint32_t idx = (uint32_t)(st.rb + it * rpw + g);
        CigarView cg;
        cg.c0 = dat_cur.cg[0]; cg.c1 = dat_cur.cg[1]; cg.c2 = dat_cur.cg[2]; cg.c3 = dat_cur.cg[3];
        cg.p = reinterpret_cast<const uint32_t*>(p.payload);
        if (slow && nc > 4u) cg.p = reinterpret_cast<const uint32_t*>(p.payload + ((unsigned long long)p.rec[idx].off8 << 3));   // (only a CIGAR of more than four ops is read again)
        const uint32_t ncs = slow ? nc : 0u;
        // [EXT] pysam query_alignment_start / _end -> len(aln.query_alignment_sequence) (midas/run/snps.py:145)
        long long qs = 0, qe = 0;
        query_bounds(cg, ncs, l, &qs, &qe);
        long long al = qe - qs;
        al = al < 0 ? 0 : (al > 2047 ? 2047 : al);      // (l_seq <= 1024)
        const int align_g = (int)al;
        // the one case in which count_coverage raises IndexError for a kept read: a match op maps a query position >= l_seq
        // onto a site inside the contig
        bool t_over = false;
        {
          long long qpos = 0, rpos = pos;
          const long long clen = tile.contig_len;
          for_each_op(cg, ncs, [&](uint32_t, uint32_t v) {
            const uint32_t op = v & 15u;
            const long long len = (long long)(v >> 4);
            if (op_is_match(op)) {
              if (qpos + len > (long long)l) {
                const long long qs2 = qpos > (long long)l ? qpos : (long long)l;
                const long long rs = rpos + (qs2 - qpos), re = rpos + len;
                if (rs < clen && re > 0) t_over = true;
              }
              qpos += len;
              rpos += len;
            } else if (op == OP_I || op == OP_S || (op == OP_P && p.pad_advances)) {
              qpos += len;
            } else if (op == OP_D || op == OP_N) {
              rpos += len;
            }
            return true;
          });
        }
        // ---- keep_read, every test evaluated, the reference's order decides which outcome wins ---------------------------
        const int min_match = s_tables[align_g < p.table_len ? align_g : 0];
        const int min_align = s_tables[p.table_len + (l < p.table_len ? l : 0)];
        const bool t_noseq = l == 0;
        const bool t_nonm = nm16 == 0xFFFFu;
        const bool t_zero = align_g == 0;
        const bool t_pid = align_g - nm < min_match;
        const bool t_drop = ((int)qsum < rq * l) | (mapq < p.mapq_min) | (align_g < min_align);
        uint32_t e = t_over ? (uint32_t)E_CIGAR_OVERRUN : 0u;
        e = t_drop ? 0u : e;
        e = t_noqual ? (uint32_t)E_NO_QUAL : e;
        e = t_pid ? 0u : e;
        e = t_zero ? (uint32_t)E_ZERO_ALIGN : e;
        e = t_nonm ? (uint32_t)E_NO_NM : e;
        e = t_noseq ? (uint32_t)E_NO_SEQ : e;
        const bool keep_s = slow && !(t_noseq | t_nonm | t_zero | t_pid | t_noqual | t_drop | t_over);
        // owner tile of a read = the tile holding its (clamped) start: it alone counts the read in the stats
        int cpos = pos < 0 ? 0 : pos;
        cpos = cpos > tile.contig_len - 1 ? tile.contig_len - 1 : cpos;
        const bool owner_s = slow && cpos >= tile_start && cpos < tile_start + tile_len && !(tile.halo && pos < 0);
        if (slow) { err = e; keep = keep_s; owner = owner_s; }
        const int rel = pos - tile_start;                                     // may wrap for absurd positions:
        int rrel = (rel > (1 << 25) || rel < -(1 << 30)) ? (1 << 25) : rel;   // those are parked far right
        // ---- CIGAR walk ([EXT] get_aligned_pairs(matches_only=True)): one match segment at a time ------------------------
        // 32-bit saturating positions: a query position only matters below q1 <= 1024 and a tile-relative reference
        // position only below 4096, and both only ever grow.
        uint32_t k = 0;
        int qpos = 0, jlo = 0, jhi = 0, loc0 = 0;
        const int q1 = q0 + nb;
        auto next_segment = [&]() -> bool {
          while (k < nc) {
            const uint32_t v = cg[k];
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
        bool walking = keep_s && has;
        if (walking) walking = next_segment();
        while (__ballot(walking) != 0ull) {
          const int lo = jlo > -loc0 ? jlo : -loc0;
          const int hi = jhi < ext_len - loc0 ? jhi : ext_len - loc0;
          if (!(kDebug & 1)) tally_range(walking && lo < hi, lo, hi, loc0, qv, dat_cur.s, std::true_type{});
          walking = (walking && k < nc) ? next_segment() : false;
        }
      }
      // ---- per-species read counters: one ballot per wave ---------------------------------------------------------------
      MIDAS_MARK("counters");
      const bool head = owner && c == 0;
      w_aligned += (uint32_t)__popcll(__ballot(head));
      w_mapped += (uint32_t)__popcll(__ballot(head && keep));
      // (a read of the piece in front, midas_snps_contigs.origin: its own piece reports what keep_read raises, this one the
      // overrun its walk runs into here)
      const bool walk_err = slow && tile.halo && pos < 0 && c == 0 && err == (uint32_t)E_CIGAR_OVERRUN;
      if ((head && err) || walk_err) {
        const uint32_t idx = (uint32_t)(st.rb + it * rpw + g);
        atomicMin(p.err, ((unsigned long long)idx << 8) | err);
      }

      MIDAS_MARK("iterend");
#if MIDAS_SNPS_DEBUG_BITS & 256
      pr_work += PROBE_NOW() - pt2;
#endif
    };
    {
      int it = wave, k_it = 0;
      bool odd = false;
      while (it < it_hi) {
        iteration(it, k_it, rdA, datA, rdB, datB);
        it += NWAVES;
        ++k_it;
        if (it >= it_hi) { odd = true; break; }
        iteration(it, k_it, rdB, datB, rdA, datA);
        it += NWAVES;
        ++k_it;
      }
      if (odd) { rdA = rdB; datA = datB; }      // (once per tile, not per iteration)
    }

    // the tile's reference letters: requested here, behind the stream loop (two registers less in it), they arrive while
    // the workgroup waits for its last wave and writes the counts out
    uint32_t refw[REF_IT];
    if (p.out_allele) {
      const uint8_t* ref = p.ref + tile.site_base;
#pragma unroll
      for (int it = 0; it < REF_IT; ++it) {
        const int i = 4 * (tid + it * kDirectBlock);
        if (i + 4 <= tile_len) refw[it] = *reinterpret_cast<const u32_a1*>(ref + i);
      }
    }
    if (lane == 0) {
      if (w_aligned) atomicAdd(&s_stats[MIDAS_STAT_ALIGNED], (unsigned long long)w_aligned);
      if (w_mapped) atomicAdd(&s_stats[MIDAS_STAT_MAPPED], (unsigned long long)w_mapped);
    }
    // ---- next tile ---------------------------------------------------------------------------------------------------------
    const int tn = wn;
    const Tile ntile = load_tile(c_tiles, tn);
    const Stream nst = load_stream(tn, cout);
    const bool take = more && !in_chunk;          // the next tile opens the next item: draw the one after it
    const unsigned long long ps0 = PROBE_NOW();
    if (more && !xt) {   // (with xt the pipeline already holds the next tile's first iterations)
      // (the columns are requested in front of the barrier, the bases behind it)
      const Raw r0 = fetch(nst, wave);
      rawX = fetch(nst, wave + NWAVES);
      nrX = reads_in(nst, wave + NWAVES);
      lds_barrier();       // every tally of this tile is in LDS
      settle(r0, reads_in(nst, wave), rdA, datA);
    } else {
      lds_barrier();       // every tally of this tile is in LDS
    }
    const unsigned long long ps1 = PROBE_NOW();
    uint32_t ticket = 0;
    if (dynamic && take && tid == 0) ticket = atomicAdd(&sched[32 * sched_group], 1u);

    // ---- emit the tile: counts[site][A,C,G,T] (and re-zero LDS), covered / total-depth partials ---------------------------
    {
      uint4* out = reinterpret_cast<uint4*>(p.out_counts) + ((kDebug & 8) ? 0 : tile.site_base);
      uint4* lds4 = reinterpret_cast<uint4*>(lds);
      const int lim = (kDebug & 2) ? 0 : tile_len;
#pragma unroll
      for (int it = 0; it < OUT_IT; ++it) {
        const int i = tid + it * kDirectBlock;
        if (i < lim) {
          const uint4 v = lds4[i];
          if (cout && i < OV) {                 // what this tile's reads added behind it opens the next tile
            lds4[i] = lds4[TILE + i];
            lds4[TILE + i] = make_uint4(0u, 0u, 0u, 0u);
          } else {
            lds4[i] = make_uint4(0u, 0u, 0u, 0u);
          }
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
        const int i = 4 * (tid + it * kDirectBlock);
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
    if (dynamic && take && tid == 0) s_next_ticket = ticket;
    lds_barrier();       // tallies re-zeroed, this tile's s_stats additions done
#if MIDAS_SNPS_DEBUG_BITS & 256
    pr_sync += ps1 - ps0;
    pr_out += PROBE_NOW() - ps1;
#else
    (void)ps0; (void)ps1;
#endif
    if (take) {
      const long long nn = dynamic ? 2ll * (long long)gridDim.x + (long long)kSchedGroups * s_next_ticket + sched_group
                                   : (long long)i_next + (long long)gridDim.x;
      c_first = item_first(i_next);
      c_end = item_end(i_next);
      i_next = __builtin_amdgcn_readfirstlane((int)(nn < (long long)w_end ? nn : (long long)w_end));
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
    t = tn;
    tile = ntile;
    st = nst;
  }
#if MIDAS_SNPS_DEBUG_BITS & 256
  if (lane == 0 && p.probe) {      // per wave: cycles waiting for columns / bases, working, at the barrier, writing out; iterations; all
    unsigned long long* o = p.probe + ((size_t)blockIdx.x * NWAVES + wave) * 8;
    uint32_t hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_id), "=s"(xcc_id));
    o[0] = pr_cols; o[1] = (unsigned long long)hw_id | ((unsigned long long)xcc_id << 32); o[2] = pr_work; o[3] = pr_sync; o[4] = pr_out; o[5] = pr_iters;
    o[6] = __builtin_readcyclecounter() - pr_t0; o[7] = 0ull;
  }
#endif
#undef PROBE_NOW
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
  const size_t dyn_lds = (size_t)p.table_len * 2 * (p.overhang > kDirectOverhang ? sizeof(int16_t) : sizeof(int32_t));
  const int k = p.chunk_tiles > 1 ? p.chunk_tiles : 1;
  const int n_items = k > 1 ? p.n_chunked_tiles / k + (p.n_tiles - p.n_chunked_tiles) : p.n_tiles;      // (every workgroup of the grid has work)
  const int grid = n_items < p.grid_blocks ? n_items : p.grid_blocks;
  const bool bq0 = p.baseq <= 0;
  if (p.overhang > kDirectOverhang) {      // (a batch of reads longer than the common overhang: the instantiation with the long one)
    if (lane_bases == 32) {
      if (bq0) hipLaunchKernelGGL((pileup_direct_kernel<32, true, kDirectOverhangLong>), dim3(grid), dim3(kDirectBlock), dyn_lds, stream, p);
      else hipLaunchKernelGGL((pileup_direct_kernel<32, false, kDirectOverhangLong>), dim3(grid), dim3(kDirectBlock), dyn_lds, stream, p);
    } else {
      if (bq0) hipLaunchKernelGGL((pileup_direct_kernel<30, true, kDirectOverhangLong>), dim3(grid), dim3(kDirectBlock), dyn_lds, stream, p);
      else hipLaunchKernelGGL((pileup_direct_kernel<30, false, kDirectOverhangLong>), dim3(grid), dim3(kDirectBlock), dyn_lds, stream, p);
    }
  } else if (lane_bases == 32) {
    if (bq0) hipLaunchKernelGGL((pileup_direct_kernel<32, true>), dim3(grid), dim3(kDirectBlock), dyn_lds, stream, p);
    else hipLaunchKernelGGL((pileup_direct_kernel<32, false>), dim3(grid), dim3(kDirectBlock), dyn_lds, stream, p);
  } else {
    if (bq0) hipLaunchKernelGGL((pileup_direct_kernel<30, true>), dim3(grid), dim3(kDirectBlock), dyn_lds, stream, p);
    else hipLaunchKernelGGL((pileup_direct_kernel<30, false>), dim3(grid), dim3(kDirectBlock), dyn_lds, stream, p);
  }
  return hipGetLastError();
}

}  // namespace midas
