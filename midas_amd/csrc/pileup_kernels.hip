// gfx950 (CDNA4) kernels of the MIDAS SNP pileup.  Integer counting, HBM-bound: no MFMA.
//
// Reference semantics implemented here (citations into /root/reference):
//   keep_read                       midas/run/snps.py:141-162
//   count_coverage call site        midas/run/snps.py:194-199  ([EXT] pysam: get_aligned_pairs(matches_only),
//                                   qual >= quality_threshold, only 'A','C','G','T' counted)
//   depth / covered / total_depth   midas/run/snps.py:204-213
//   str(rec.seq).upper()            midas/run/snps.py:62
//   index_bam                       midas/run/snps.py:130-137 (here: per-tile read ranges, built on device)
//
// Work decomposition.  The site space is cut into tiles of 4096 sites that never span contigs.  One
// 512-thread workgroup owns a tile: its tallies live in LDS as [site][A,C,G,T] u32, reads are streamed
// straight from the packed HBM arrays through a two-deep register prefetch pipeline, tallies are LDS
// atomics, and the tile is written out once, 16 B per site, fully coalesced, with the per-species
// counters reduced by wave shuffles on the way out.
//
// Lane mapping.  A lane owns kChunk = 32 consecutive bases of one read: two 16-byte loads of quals and
// one 16-byte load of 4-bit call codes.  A read of l_seq bases occupies ceil(l_seq/32) adjacent lanes
// (5 for 150 bp; `lanes_per_read` is fixed per batch from the longest read) and a wave works on
// floor(64 / lanes_per_read) reads at a time.  The read filter needs the quality sum of the whole
// read: a segmented shuffle reduction over the read's lanes gives it without re-reading anything.
//
// The kernel is VALU-issue bound, not HBM bound, until the per-base instruction count is tiny, so the
// per-base work is arranged as: (1) SWAR, four bases per instruction -- everything that decides
// WHETHER a base counts (not A/C/G/T, read tail, CIGAR segment, tile edge) is folded into the
// quality byte itself (a base that must not count gets quality 0); (2) per base -- one byte compare
// against baseq, one OR that forms the LDS address (site << 4 | call code), one predicated ds_add.
#include "kernels.h"

namespace midas {

namespace {

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

__device__ __forceinline__ bool consumes_both(uint32_t op) { return op == OP_M || op == OP_EQ || op == OP_X; }

// Unpacked view of a 16-byte ReadRec held as a uint4.
__device__ __forceinline__ int rec_pos(const uint4& r) { return (int)r.x; }
__device__ __forceinline__ uint32_t rec_off8(const uint4& r) { return r.y; }
__device__ __forceinline__ int rec_l(const uint4& r) { return (int)(r.z & 0xFFFFu); }
__device__ __forceinline__ int rec_n(const uint4& r) { return (int)(r.z >> 16); }
__device__ __forceinline__ uint32_t rec_nm(const uint4& r) { return r.w & 0xFFFFu; }
__device__ __forceinline__ int rec_mapq(const uint4& r) { return (int)((r.w >> 16) & 0xFFu); }
__device__ __forceinline__ uint32_t rec_flags(const uint4& r) { return r.w >> 24; }

// ------------------------------------------------------------------------------------------------
// Index kernel: one thread per read.  Finds the reference span (from the record alone for the
// common single-match CIGAR, else by walking the CIGAR) and records, for every tile the read
// overlaps, the lowest and highest read index seen.  The pileup kernel scans exactly that range per
// tile, so a long deletion in one read widens the scan of the tiles it really crosses and of no
// other.  Sortedness of the input only affects how tight these ranges are.  Block 0 also resets the
// per-species counters and the error word for this run (the tile ranges reset themselves: each
// pileup workgroup zeroes its own entry after reading it).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kIndexBlock) void index_reads_kernel(IndexParams p) {
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < p.n_stat_words; i += kIndexBlock) p.stats[i] = 0ull;
    if (threadIdx.x == 0) *p.err = kNoError;
  }
  const int i = blockIdx.x * kIndexBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = i < p.n_reads;
  int gt0 = -1, gt1 = -1;
  if (valid) {
    int lo = 0, hi = p.n_contigs;  // read_begin[lo] <= i < read_begin[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (p.contig_read_begin[mid] <= i) lo = mid; else hi = mid;
    }
    const uint4 r = reinterpret_cast<const uint4*>(p.rec)[i];
    long long reflen = rec_l(r);
    if (!(rec_flags(r) & kRecSimple)) {
      const uint32_t* cig = reinterpret_cast<const uint32_t*>(p.blob + (size_t)rec_off8(r) * 8 +
                                                              blob_cigar_off((uint32_t)rec_l(r)));
      reflen = 0;
      const int n = rec_n(r);
      for (int k = 0; k < n; ++k) {
        const uint32_t v = cig[k];
        const uint32_t op = v & 15u;
        if (consumes_both(op) || op == OP_D || op == OP_N) reflen += (long long)(v >> 4);
      }
    }
    const long long clen = p.contig_len[lo];
    long long p0 = rec_pos(r);
    p0 = p0 < 0 ? 0 : (p0 > clen - 1 ? clen - 1 : p0);
    long long p1 = (long long)rec_pos(r) + (reflen > 0 ? reflen : 1) - 1;
    p1 = p1 < p0 ? p0 : (p1 > clen - 1 ? clen - 1 : p1);
    const int tb = p.contig_tile_base[lo];
    gt0 = tb + (int)(p0 >> kTileShift);
    gt1 = tb + (int)(p1 >> kTileShift);
  }
  // Reads are (normally) sorted, so a wave mostly sees runs of one tile: only the first lane of a run
  // publishes the low bound and only the last one the high bound.
  const int prev = __shfl_up(gt0, 1);
  const int next = __shfl_down(gt0, 1);
  if (valid) {
    const uint32_t inv = (uint32_t)(p.n_reads - i);
    if (lane == 0 || prev != gt0) atomicMax(&p.rbinv[gt0], inv);
    if (lane == 63 || next != gt0) atomicMax(&p.rend[gt0], (uint32_t)(i + 1));
    for (int t = gt0 + 1; t <= gt1; ++t) {
      atomicMax(&p.rbinv[t], inv);
      atomicMax(&p.rend[t], (uint32_t)(i + 1));
    }
  }
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

template <int TILE_SHIFT>
__global__ __launch_bounds__(kPileupBlock, 4) void pileup_tiles_kernel(PileupParams p) {
  constexpr int TILE = 1 << TILE_SHIFT;
  constexpr int NW = kChunk / 4;   // quality words per lane
  __shared__ __attribute__((aligned(16))) uint32_t lds[4 * TILE];
  __shared__ unsigned long long s_stats[MIDAS_STATS];
  __shared__ int32_t s_range[2];
  extern __shared__ __attribute__((aligned(16))) int32_t s_tables[];   // [min_match table_len][min_align table_len]

  // Consecutive tiles share their straddling reads: keep neighbours on one XCD (block b runs on XCD b % 8).
  const int t = (int)(blockIdx.x & 7u) * p.tiles_per_xcd + (int)(blockIdx.x >> 3);
  if (t >= p.n_tiles) return;
  const Tile tile = p.tiles[t];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  {
    uint4* z = reinterpret_cast<uint4*>(lds);
    for (int i = tid; i < TILE; i += kPileupBlock) z[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < p.table_len; i += kPileupBlock) {
      s_tables[i] = p.filt->min_match[i];
      s_tables[p.table_len + i] = p.filt->min_align[i];
    }
    if (tid < MIDAS_STATS) s_stats[tid] = 0ull;
    if (tid == 0) {
      const uint32_t rbinv = p.rbinv[t];
      const int re0 = (int)p.rend[t];
      s_range[0] = rbinv ? p.n_reads - (int)rbinv : re0;
      s_range[1] = re0;
      p.rbinv[t] = 0u;   // leave the index clean for the next run
      p.rend[t] = 0u;
    }
  }
  __syncthreads();
  const int rb = s_range[0];
  const int re = s_range[1];

  const int lpr = p.lanes_per_read;
  const int rpw = p.reads_per_wave;
  const int g = lane / lpr;
  const int c = lane - g * lpr;
  const bool lane_used = g < rpw;
  const int q0 = c * kChunk;
  const int stride = (kPileupBlock / 64) * rpw;
  const uint4* recs = reinterpret_cast<const uint4*>(p.rec);
  const int tile_len = tile.len;
  const int tile_start = tile.start;
  const int bq = p.baseq < 1 ? 1 : p.baseq;   // baseq <= 0 counts every base: validity bytes become 0xFF >= 1
  const bool count_all = p.baseq < 1;
  char* const lds_bytes = reinterpret_cast<char*>(lds);
  uint32_t w_aligned = 0, w_mapped = 0;

  // ---- two-deep prefetch: records two iterations ahead, payload one iteration ahead -------------
  auto fetch_rec = [&](int b) -> uint4 {
    const int r = b + g;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (lane_used && r < re) v = recs[r];
    return v;
  };
  struct Payload {
    uint32_t qw[NW];  // 32 quality bytes (zero beyond the end of the read: the blob is padded)
    uint32_t sw[4];   // 32 call codes
    uint32_t cg[4];   // first four CIGAR ops (non-simple reads only)
    uint32_t cl;      // last CIGAR op
  };
  auto fetch_payload = [&](const uint4& rv, int b, Payload& d) {
    const int r = b + g;
    const bool act = lane_used && r < re;
    const int l = rec_l(rv);
    const int n = rec_n(rv);
    const uint8_t* bp = p.blob + (size_t)rec_off8(rv) * 8;
#pragma unroll
    for (int w = 0; w < NW; ++w) d.qw[w] = 0u;
    d.sw[0] = d.sw[1] = d.sw[2] = d.sw[3] = 0u;
    d.cg[0] = d.cg[1] = d.cg[2] = d.cg[3] = 0u;
    d.cl = 0u;
    if (act && q0 < l) {
      const u32x4_a8 qa = *reinterpret_cast<const u32x4_a8*>(bp + q0);
      const u32x4_a8 qb = *reinterpret_cast<const u32x4_a8*>(bp + q0 + 16);
      const u32x4_a8 sv = *reinterpret_cast<const u32x4_a8*>(bp + blob_seq_off((uint32_t)l) + (q0 >> 1));
      d.qw[0] = qa.x; d.qw[1] = qa.y; d.qw[2] = qa.z; d.qw[3] = qa.w;
      d.qw[4] = qb.x; d.qw[5] = qb.y; d.qw[6] = qb.z; d.qw[7] = qb.w;
      d.sw[0] = sv.x; d.sw[1] = sv.y; d.sw[2] = sv.z; d.sw[3] = sv.w;
    }
    if (act && !(rec_flags(rv) & kRecSimple) && n > 0) {
      const uint32_t* cig = reinterpret_cast<const uint32_t*>(bp + blob_cigar_off((uint32_t)l));
      const u32x4_a4 cv = *reinterpret_cast<const u32x4_a4*>(cig);   // may overhang into padding / next blob
      d.cg[0] = cv.x; d.cg[1] = cv.y; d.cg[2] = cv.z; d.cg[3] = cv.w;
      if (n > 4) d.cl = cig[n - 1];   // n <= 4: the last op is one of cg[0..3]; picked at use, never here (a use would drain the prefetch)
    }
  };

  int base = rb + wave * rpw;
  uint4 rec_cur = fetch_rec(base);
  uint4 rec_nxt = fetch_rec(base + stride);
  Payload cur;
  fetch_payload(rec_cur, base, cur);

  for (; base < re; base += stride) {
    const uint4 rec_nn = fetch_rec(base + 2 * stride);
    Payload nxt;
    fetch_payload(rec_nxt, base + stride, nxt);

    // ================= process (rec_cur, cur) =====================================================
    const int r = base + g;
    bool act = lane_used && r < re;
    const int l = rec_l(rec_cur);
    const int n = rec_n(rec_cur);
    const int pos = rec_pos(rec_cur);
    const uint32_t flags = rec_flags(rec_cur);
    const bool simple = (flags & kRecSimple) != 0u;
    // owner tile of a read = the tile holding its (clamped) start: it alone counts the read in the stats
    int cpos = pos < 0 ? 0 : pos;
    cpos = cpos > tile.contig_len - 1 ? tile.contig_len - 1 : cpos;
    const bool owner = act && cpos >= tile_start && cpos < tile_start + tile_len;
    // position of the read relative to the tile; a read that can never reach the tile is parked far right
    // (reference positions only grow along a CIGAR, so "far right" stays far right)
    const long long rel64 = (long long)pos - (long long)tile_start;
    int rrel = (rel64 > (1LL << 25) || rel64 < -(1LL << 30)) ? (1 << 25) : (int)rel64;
    // reads that start in an earlier tile and provably end before this one: nothing to do here
    if (act && !owner && n == 1 && rrel + l <= 0) act = false;
    const bool has = act && q0 < l;

    // ---- soft-clip trimming ([EXT] pysam getQueryStart / getQueryEnd) -----------------------------
    int k0 = 0, lead_s = 0, trail_s = 0;
    const uint32_t* cig = nullptr;
    if (act && !simple) {
      cig = reinterpret_cast<const uint32_t*>(p.blob + (size_t)rec_off8(rec_cur) * 8 + blob_cigar_off((uint32_t)l));
      if (!(flags & kRecClipGeneric)) {
        if (n > 0 && (cur.cg[0] & 15u) == OP_S) { lead_s = (int)(cur.cg[0] >> 4); k0 = 1; }
        const uint32_t last = n > 4 ? cur.cl : (n == 2 ? cur.cg[1] : (n == 3 ? cur.cg[2] : cur.cg[3]));
        if (n > 1 && (last & 15u) == OP_S) trail_s = (int)(last >> 4);
      } else {
        while (k0 < n) {
          const uint32_t v = cig[k0];
          const uint32_t op = v & 15u;
          if (op == OP_H) { ++k0; }
          else if (op == OP_S) { lead_s += (int)(v >> 4); ++k0; }
          else break;
        }
        for (int k = n - 1; k >= 1; --k) {   // index 0 is never inspected by pysam's backward walk
          const uint32_t v = cig[k];
          const uint32_t op = v & 15u;
          if (op == OP_H) continue;
          if (op == OP_S) trail_s += (int)(v >> 4); else break;
        }
      }
    }
    int align_len = (l - trail_s) - lead_s;
    align_len = align_len < 0 ? 0 : align_len;
    // exact integer form of the two fp64 ratio tests (tables built by the host with the reference's expressions)
    const int min_match = s_tables[align_len < p.table_len ? align_len : 0];
    const int min_align = s_tables[p.table_len + (l < p.table_len ? l : 0)];

    // ---- quality sum of the whole read: per-lane partial, then a segmented reduction ------------
    const int nvalid = has ? (l - q0 < kChunk ? l - q0 : kChunk) : 0;
    uint32_t part = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) part = __builtin_amdgcn_sad_u8(cur.qw[w], 0u, part);
    for (int d = 1; d < lpr; d <<= 1) {
      const uint32_t o = __shfl_down(part, d);
      if (c + d < lpr) part += o;
    }
    const int qsum = (int)__shfl(part, lane - c);

    // ---- keep_read (midas/run/snps.py:141-162), same order of evaluation ------------------------
    bool keep = false;
    uint32_t err = 0;
    if (act) {
      if (l == 0) err = E_NO_SEQ;
      else if (rec_nm(rec_cur) == kNmAbsent) err = E_NO_NM;
      else if (align_len == 0) err = E_ZERO_ALIGN;
      else if (align_len - (int)rec_nm(rec_cur) < min_match) keep = false;            // pid < mapid
      else if (flags & kRecQualAbsent) err = E_NO_QUAL;
      // np.mean(q) < readq  <=>  sum(q) < readq * n exactly (integers; the quotient is >= 2^-16 away from readq)
      else if ((long long)qsum < (long long)p.readq * (long long)l) keep = false;
      else if (rec_mapq(rec_cur) < p.mapq) keep = false;
      else if (align_len < min_align) keep = false;                                     // aln_cov
      else if (flags & kRecOverrun) err = E_CIGAR_OVERRUN;   // kept, and its CIGAR reaches past SEQ inside the contig
      else keep = true;
    }

    // ---- per-base call codes and validity, four bases per instruction ------------------------------
    uint32_t qv[NW];   // quality byte if the base may count (is A/C/G/T, inside the read), else 0
    uint32_t cd[NW];   // byte offset of the base's counter inside its site (call code & 0xC)
    bool walking = keep && has;
    if (walking) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const uint32_t x16 = cur.sw[w >> 1] >> (16 * (w & 1));               // call bytes 2w, 2w+1: bases 4w .. 4w+3
        const uint32_t t4 = __builtin_amdgcn_perm(0u, x16, 0x01010000u);     // [b0, b0, b1, b1]
        const uint32_t nib = ((t4 >> 4) & 0x000F000Fu) | (t4 & 0x0F000F00u);  // one call code per byte
        const uint32_t inv = nib & 0x02020202u;                              // not A/C/G/T
        const uint32_t inv_ff = (inv << 7) - (inv >> 1);                     // 0xFF in every such byte
        const uint32_t q = count_all ? low_bytes_mask(nvalid - 4 * w) : cur.qw[w];
        qv[w] = q & ~inv_ff;
        cd[w] = nib & 0x0C0C0C0Cu;
      }
    } else {
#pragma unroll
      for (int w = 0; w < NW; ++w) { qv[w] = 0u; cd[w] = 0u; }
    }

    // ---- CIGAR walk ([EXT] get_aligned_pairs(matches_only=True)): one match segment at a time -------
    // 32-bit saturating positions: a query position only matters below q1 <= 1024 and a tile-relative
    // reference position only below 4096, and both only ever grow.
    int k = k0;
    int qpos = lead_s;
    const int q1 = q0 + nvalid;
    int jlo = 0, jhi = 0, loc0 = 0;
    auto next_segment = [&]() -> bool {
      while (k < n) {
        const uint32_t v = k < 4 ? (k == 0 ? cur.cg[0] : (k == 1 ? cur.cg[1] : (k == 2 ? cur.cg[2] : cur.cg[3]))) : cig[k];
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
      if (simple) {
        jlo = 0; jhi = nvalid; loc0 = rrel + q0; k = n;
      } else {
        walking = next_segment();
      }
    }
    while (walking) {
      // bases of the chunk that belong to this segment AND lie inside the tile: [lo, hi)
      uint32_t q4[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) q4[w] = qv[w];
      const int lo = jlo > -loc0 ? jlo : -loc0;
      const int hi = jhi < tile_len - loc0 ? jhi : tile_len - loc0;
      if (lo > 0 || hi < nvalid) {    // partial chunk (segment border or tile edge): zero the bytes outside
        const uint32_t below_hi = hi >= 32 ? 0xFFFFFFFFu : (hi <= 0 ? 0u : ((1u << hi) - 1u));
        const uint32_t below_lo = lo >= 32 ? 0xFFFFFFFFu : (lo <= 0 ? 0u : ((1u << lo) - 1u));
        const uint32_t jm = below_hi & ~below_lo;
#pragma unroll
        for (int w = 0; w < NW; ++w) q4[w] &= bits_to_bytes((jm >> (4 * w)) & 0xFu);
      }
      const uint32_t abase = (uint32_t)loc0 << 4;
#pragma unroll
      for (int j = 0; j < kChunk; ++j) {
        const uint32_t q = (q4[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
        const uint32_t code = (cd[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
        if (q >= (uint32_t)bq) atomicAdd(reinterpret_cast<uint32_t*>(lds_bytes + ((abase | code) + 16u * j)), 1u);
      }
      walking = (k < n) ? next_segment() : false;
    }

    // ---- per-species read counters: one ballot per wave -------------------------------------------
    const bool head = owner && c == 0;
    const unsigned long long m_al = __ballot(head);
    const unsigned long long m_mp = __ballot(head && keep);
    w_aligned += (uint32_t)__popcll(m_al);
    w_mapped += (uint32_t)__popcll(m_mp);
    if (head && err) atomicMin(p.err, ((unsigned long long)(uint32_t)r << 8) | err);

    rec_cur = rec_nxt;
    rec_nxt = rec_nn;
    cur = nxt;
  }

  if (lane == 0) {
    if (w_aligned) atomicAdd(&s_stats[MIDAS_STAT_ALIGNED], (unsigned long long)w_aligned);
    if (w_mapped) atomicAdd(&s_stats[MIDAS_STAT_MAPPED], (unsigned long long)w_mapped);
  }
  __syncthreads();

  // ---- emit the tile: counts[site][A,C,G,T], upper-cased ref allele, covered/total-depth partials -
  unsigned long long covered = 0, depth_sum = 0;
  uint4* out = reinterpret_cast<uint4*>(p.out_counts) + tile.site_base;
  const uint4* lds4 = reinterpret_cast<const uint4*>(lds);
  for (int i = tid; i < tile_len; i += kPileupBlock) {
    const uint4 v = lds4[i];
    out[i] = v;
    if (p.out_allele) {
      uint32_t ch = p.ref[tile.site_base + i];
      if (ch >= 'a' && ch <= 'z') ch -= 32u;
      p.out_allele[tile.site_base + i] = (uint8_t)ch;
    }
    const uint32_t d = v.x + v.y + v.z + v.w;
    covered += d > 0u ? 1ull : 0ull;
    depth_sum += d;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    covered += __shfl_down(covered, d);
    depth_sum += __shfl_down(depth_sum, d);
  }
  if (lane == 0) {
    if (covered) atomicAdd(&s_stats[MIDAS_STAT_COVERED], covered);
    if (depth_sum) atomicAdd(&s_stats[MIDAS_STAT_DEPTH], depth_sum);
  }
  __syncthreads();
  if (tid < MIDAS_STATS && s_stats[tid]) atomicAdd(&p.stats[(size_t)tile.species * MIDAS_STATS + tid], s_stats[tid]);
}

}  // namespace

hipError_t launch_index_reads(const IndexParams& p, hipStream_t stream) {
  // always launched (even with no reads): block 0 resets the counters and the error word
  const int grid = p.n_reads > 0 ? (p.n_reads + kIndexBlock - 1) / kIndexBlock : 1;
  hipLaunchKernelGGL(index_reads_kernel, dim3(grid), dim3(kIndexBlock), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_pileup_tiles(const PileupParams& p, hipStream_t stream) {
  if (p.n_tiles <= 0) return hipSuccess;
  const int grid = p.tiles_per_xcd * 8;
  const size_t dyn_lds = (size_t)p.table_len * 2 * sizeof(int32_t);
  hipLaunchKernelGGL(pileup_tiles_kernel<kTileShift>, dim3(grid), dim3(kPileupBlock), dyn_lds, stream, p);
  return hipGetLastError();
}

}  // namespace midas
