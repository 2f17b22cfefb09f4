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
// 512-thread workgroup owns a tile: its A/C/G/T tallies live in LDS as four planes (plane-major, so the
// 16 consecutive sites a lane touches fall into 16 different banks), reads are streamed straight from
// the packed HBM arrays, tallies are LDS atomics, and the tile is written out once, 16 B per site,
// fully coalesced, with the per-species counters reduced by wave shuffles on the way out.
//
// Lane mapping.  A lane owns 16 consecutive bases of one read: one 16-byte load of quals and one
// 8-byte load of packed bases.  A read of l_seq bases therefore occupies ceil(l_seq/16) adjacent
// lanes (10 for 150 bp; `lanes_per_read` is fixed per batch from the longest read), and a wave works
// on floor(64 / lanes_per_read) reads at a time.  The read filter needs the quality sum of the whole
// read: a segmented shuffle reduction over the read's lanes gives it without re-reading anything.
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
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ bool consumes_both(uint32_t op) { return op == OP_M || op == OP_EQ || op == OP_X; }

// ------------------------------------------------------------------------------------------------
// Index kernel: one thread per read.  Walks the CIGAR for the reference span and records, for every
// tile the read overlaps, the lowest and highest read index seen.  The pileup kernel scans exactly
// [lowest, highest] per tile, so a long deletion in one read widens the scan of the tiles it really
// crosses and of no other.  Sortedness of the input only affects how tight these ranges are.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kIndexBlock) void index_reads_kernel(IndexParams p) {
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
    const ReadRec r = p.rec[i];
    const uint32_t* cig =
        reinterpret_cast<const uint32_t*>(p.blob + (size_t)r.blob_off8 * 8 + blob_cigar_off(r.l_seq));
    long long reflen = 0;
    for (int k = 0; k < (int)r.n_cigar; ++k) {
      const uint32_t v = cig[k];
      const uint32_t op = v & 15u;
      if (consumes_both(op) || op == OP_D || op == OP_N) reflen += (long long)(v >> 4);
    }
    const long long clen = p.contig_len[lo];
    long long p0 = r.pos;
    p0 = p0 < 0 ? 0 : (p0 > clen - 1 ? clen - 1 : p0);
    long long p1 = (long long)r.pos + (reflen > 0 ? reflen : 1) - 1;
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

// ------------------------------------------------------------------------------------------------
// Tally 16 bases held in registers into the tile's LDS planes.
//   qw[4]  : 16 quality bytes        sw[2] : 16 packed 4-bit base codes (first base in the high nibble)
//   [jlo,jhi) : which of the 16 belong to the current CIGAR match segment
//   local0 : tile-relative site of base 0 of the chunk (may be negative / past the tile)
// ------------------------------------------------------------------------------------------------
template <int TILE>
__device__ __forceinline__ void tally16(uint32_t* cnt, const uint32_t (&qw)[4], const uint32_t (&sw)[2],
                                        int jlo, int jhi, int local0, int tile_len, int baseq) {
  const uint32_t span = (uint32_t)(jhi - jlo);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t q = (qw[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
    const uint32_t nb = (sw[j >> 3] >> (4 * ((j & 7) ^ 1))) & 0xFu;
    const int loc = local0 + j;
    const bool onehot = nb != 0u && (nb & (nb - 1u)) == 0u;  // exactly A(1) C(2) G(4) T(8)
    const bool ok = (uint32_t)(j - jlo) < span && (uint32_t)loc < (uint32_t)tile_len && (int)q >= baseq && onehot;
    if (ok) {
      const int plane = __builtin_ctz(nb);
      atomicAdd(&cnt[plane * TILE + loc], 1u);
    }
  }
}

template <int TILE_SHIFT>
__global__ __launch_bounds__(kPileupBlock) void pileup_tiles_kernel(PileupParams p) {
  constexpr int TILE = 1 << TILE_SHIFT;
  __shared__ __attribute__((aligned(16))) uint32_t cnt[4 * TILE];
  __shared__ unsigned long long s_stats[MIDAS_STATS];

  // Consecutive tiles share their straddling reads: keep neighbours on one XCD (block b runs on XCD b % 8).
  const int t = (int)(blockIdx.x & 7u) * p.tiles_per_xcd + (int)(blockIdx.x >> 3);
  if (t >= p.n_tiles) return;
  const Tile tile = p.tiles[t];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  {
    uint4* z = reinterpret_cast<uint4*>(cnt);
    for (int i = tid; i < TILE; i += kPileupBlock) z[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < MIDAS_STATS) s_stats[tid] = 0ull;
  }
  const uint32_t rbinv = p.rbinv[t];
  const int re = (int)p.rend[t];
  const int rb = rbinv ? p.n_reads - (int)rbinv : re;
  __syncthreads();

  const int lpr = p.lanes_per_read;
  const int rpw = p.reads_per_wave;
  const int g = lane / lpr;
  const int c = lane - g * lpr;
  const bool lane_used = g < rpw;
  const int q0 = c * 16;
  uint32_t w_aligned = 0, w_mapped = 0;

  for (int base = rb + wave * rpw; base < re; base += (kPileupBlock / 64) * rpw) {
    const int r = base + g;
    bool act = lane_used && r < re;
    ReadRec rr;
    if (act) {
      rr = p.rec[r];
    } else {
      rr.pos = 0; rr.blob_off8 = 0; rr.l_seq = 0; rr.n_cigar = 0; rr.nm = 0; rr.mapq = 0; rr.flags = 0;
    }
    const int l = rr.l_seq;
    const int n = rr.n_cigar;
    // owner tile of a read = the tile holding its (clamped) start: it alone counts the read in the stats
    int cpos = rr.pos < 0 ? 0 : rr.pos;
    cpos = cpos > tile.contig_len - 1 ? tile.contig_len - 1 : cpos;
    const bool owner = act && cpos >= tile.start && cpos < tile.start + tile.len;
    // reads that start in an earlier tile and provably end before this one: nothing to do here
    if (act && !owner && n == 1 && (long long)rr.pos + l <= (long long)tile.start) act = false;

    const uint8_t* bp = p.blob + (size_t)rr.blob_off8 * 8;
    const uint32_t* cig = reinterpret_cast<const uint32_t*>(bp + blob_cigar_off((uint32_t)l));
    const bool has = act && q0 < l;
    uint32_t qw[4] = {0u, 0u, 0u, 0u};
    uint32_t sw[2] = {0u, 0u};
    if (has) {
      const u32x4_a8 qv = *reinterpret_cast<const u32x4_a8*>(bp + q0);
      const u32x2_a4 sv = *reinterpret_cast<const u32x2_a4*>(bp + blob_seq_off((uint32_t)l) + (q0 >> 1));
      qw[0] = qv.x; qw[1] = qv.y; qw[2] = qv.z; qw[3] = qv.w;
      sw[0] = sv.x; sw[1] = sv.y;
    }

    // ---- soft-clip trimming ([EXT] pysam getQueryStart / getQueryEnd) -------------------------
    int k0 = 0, lead_s = 0, trail_s = 0;
    if (act) {
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
    int align_len = (l - trail_s) - lead_s;
    align_len = align_len < 0 ? 0 : align_len;

    // ---- quality sum of the whole read: per-lane partial, then a segmented reduction ------------
    int part = 0;
    if (has) {
      const int nvalid = l - q0;  // > 0
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int nb = nvalid - 4 * w;
        const uint32_t m = nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
        part = (int)__builtin_amdgcn_sad_u8(qw[w] & m, 0u, (uint32_t)part);
      }
    }
    for (int d = 1; d < lpr; d <<= 1) {
      const int o = __shfl_down(part, d);
      if (c + d < lpr) part += o;
    }
    const int qsum = __shfl(part, lane - c);

    // ---- keep_read (midas/run/snps.py:141-162), same order of evaluation ------------------------
    bool keep = false;
    uint32_t err = 0;
    if (act) {
      if (l == 0) err = E_NO_SEQ;
      else if (rr.nm == kNmAbsent) err = E_NO_NM;
      else if (align_len == 0) err = E_ZERO_ALIGN;
      else {
        const double pid = (double)(100LL * (long long)(align_len - (int)rr.nm)) / (double)align_len;
        if (pid < p.mapid) keep = false;
        else if (rr.flags & kRecQualAbsent) err = E_NO_QUAL;
        // np.mean(q) < readq  <=>  sum(q) < readq * n exactly (integers; quotient is >= 2^-16 away from readq)
        else if ((long long)qsum < (long long)p.readq * (long long)l) keep = false;
        else if ((int)rr.mapq < p.mapq) keep = false;
        else if ((double)align_len / (double)l < p.aln_cov) keep = false;
        else keep = true;
      }
    }

    // ---- CIGAR walk + tallies ([EXT] get_aligned_pairs(matches_only=True)) -------------------------
    if (keep && has) {
      long long qpos = lead_s;
      long long rpos = rr.pos;
      const int q1 = (q0 + 16 < l) ? q0 + 16 : l;
      for (int k = k0; k < n; ++k) {
        const uint32_t v = cig[k];
        const uint32_t op = v & 15u;
        const long long len = (long long)(v >> 4);
        if (consumes_both(op)) {
          const long long lo = qpos > q0 ? qpos : q0;
          const long long hi = (qpos + len) < q1 ? (qpos + len) : q1;
          if (lo < hi) {
            long long local0 = rpos + ((long long)q0 - qpos) - (long long)tile.start;
            local0 = local0 < -(1LL << 30) ? -(1LL << 30) : (local0 > (1LL << 30) ? (1LL << 30) : local0);
            tally16<TILE>(cnt, qw, sw, (int)(lo - q0), (int)(hi - q0), (int)local0, tile.len, p.baseq);
          }
          if (qpos + len > l) {
            // query positions >= l_seq: pysam indexes past the end iff their refpos is inside the contig
            const long long qs = qpos > l ? qpos : l;
            const long long rs = rpos + (qs - qpos), rend = rpos + len;
            if (rs < (long long)tile.contig_len && rend > 0) err = E_CIGAR_OVERRUN;
          }
          qpos += len;
          rpos += len;
        } else if (op == OP_I || op == OP_S) {
          qpos += len;
        } else if (op == OP_D || op == OP_N) {
          rpos += len;
        }  // H, P and anything else: no effect
      }
    }

    // ---- per-species read counters: one ballot per wave -------------------------------------------
    const bool head = owner && c == 0;
    const unsigned long long m_al = __ballot(head);
    const unsigned long long m_mp = __ballot(head && keep);
    w_aligned += (uint32_t)__popcll(m_al);
    w_mapped += (uint32_t)__popcll(m_mp);
    if (head && err) atomicMin(p.err, ((unsigned long long)(uint32_t)r << 8) | err);
  }

  if (lane == 0) {
    if (w_aligned) atomicAdd(&s_stats[MIDAS_STAT_ALIGNED], (unsigned long long)w_aligned);
    if (w_mapped) atomicAdd(&s_stats[MIDAS_STAT_MAPPED], (unsigned long long)w_mapped);
  }
  __syncthreads();

  // ---- emit the tile: counts[site][A,C,G,T], upper-cased ref allele, covered/total-depth partials -
  unsigned long long covered = 0, depth_sum = 0;
  uint4* out = reinterpret_cast<uint4*>(p.out_counts) + tile.site_base;
  for (int i = tid; i < tile.len; i += kPileupBlock) {
    const uint32_t a = cnt[i], cc = cnt[TILE + i], gg = cnt[2 * TILE + i], tt = cnt[3 * TILE + i];
    out[i] = make_uint4(a, cc, gg, tt);
    if (p.out_allele) {
      uint32_t ch = p.ref[tile.site_base + i];
      if (ch >= 'a' && ch <= 'z') ch -= 32u;
      p.out_allele[tile.site_base + i] = (uint8_t)ch;
    }
    const uint32_t d = a + cc + gg + tt;
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
  if (p.n_reads <= 0) return hipSuccess;
  const int grid = (p.n_reads + kIndexBlock - 1) / kIndexBlock;
  hipLaunchKernelGGL(index_reads_kernel, dim3(grid), dim3(kIndexBlock), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_pileup_tiles(const PileupParams& p, hipStream_t stream) {
  if (p.n_tiles <= 0) return hipSuccess;
  const int grid = p.tiles_per_xcd * 8;
  hipLaunchKernelGGL(pileup_tiles_kernel<kTileShift>, dim3(grid), dim3(kPileupBlock), 0, stream, p);
  return hipGetLastError();
}

}  // namespace midas
