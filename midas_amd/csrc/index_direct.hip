// gfx950 index pass of the direct pileup path: the device-side `samtools index` for reads that stay where the BAM decoder
// put them (reference: midas/run/snps.py:130-137 index_bam; pysam's fetch(contig, 0, length) then walks the index).
//
// The pileup kernel (pileup_direct.hip) visits a read ONCE: it fetches the read's columns itself, decides in registers what
// kind of CIGAR it has and tallies it.  All it needs beforehand is, per tile, the run of read indices that can touch the tile:
//
//   direct_ranges_kernel   every pass (inside the timed step): one thread per read, FOUR BYTES per read -- pos[i] and
//                          pos[i - 1].  Read i compares the tile of its (clamped) start and the tile of `start + reach`
//                          with those of read i - 1 and, where they differ, writes i into the tiles in between:
//                            tend[t]   = first read that starts behind tile t
//                            tbegin[t] = first read whose start + reach gets to tile t
//                          (reach = the longest reference span of any read of the batch), so [tbegin, tend) holds every
//                          read that can touch t.  No atomics on position-sorted input; for input that is not sorted the
//                          lowest / highest index per tile by atomicMin / atomicMax (exact in any order, only wider).
//   direct_facts_kernel    ONCE per batch (batch_create; the batch never changes afterwards): validates every read's CSR
//                          extents (status + lowest read, as the packer does), adds up the algorithmic bytes, finds the
//                          longest read, the longest reference span, whether every contig's reads are in position order,
//                          and how many reads are not of the one- or two-segment forms the pileup kernel settles in registers.
// No sort, no payload, no per-read record: the pileup kernel reads pos / l_seq / NM / mapq / the CSR offsets / SEQ / QUAL / CIGAR
// where they are.
#include "direct_common.h"

namespace midas {

using namespace dev;
using namespace direct;

namespace {

constexpr int kIdxBlock = 256;
#ifndef MIDAS_IDX_RUN
#define MIDAS_IDX_RUN 2048
#endif
constexpr int kIdxRun = MIDAS_IDX_RUN;        // reads per workgroup

// contig of read i: the last contig whose first read is <= i (empty contigs share their begin with the next one)
__device__ __forceinline__ int contig_of_read(const DirectIndexParams& p, int i) {
  int lo = 0, hi = p.n_contigs;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (p.contig_read_begin[mid] > i) hi = mid; else lo = mid + 1;
  }
  int c = lo - 1;
  c = c < 0 ? 0 : c;
  return c > p.n_contigs - 1 ? p.n_contigs - 1 : c;
}
struct ContigCursor {
  int c, begin, next_begin, tile_base, tile_end;
  long long clen;
  __device__ __forceinline__ void fetch(const DirectIndexParams& p) {
    begin = p.contig_read_begin[c];
    next_begin = p.contig_read_begin[c + 1];
    clen = p.contig_len[c];
    tile_base = p.contig_tile_base[c];
    tile_end = p.contig_tile_base[c + 1];
  }
  __device__ __forceinline__ void advance(const DirectIndexParams& p, int i) {
    if (i < next_begin || c + 1 >= p.n_contigs) return;
    while (c + 1 < p.n_contigs && i >= p.contig_read_begin[c + 1]) ++c;
    fetch(p);
  }
};

__device__ __forceinline__ unsigned long long block_sum(unsigned long long v, unsigned long long* lds4) {
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}
__device__ __forceinline__ unsigned long long block_max(unsigned long long v, unsigned long long* lds4) {
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_down(v, d);
    v = o > v ? o : v;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  const unsigned long long a = lds4[0] > lds4[1] ? lds4[0] : lds4[1], b = lds4[2] > lds4[3] ? lds4[2] : lds4[3];
  return a > b ? a : b;
}

// ---- per pass: the tile ranges, from the positions alone ---------------------------------------------------------------
template <bool SORTED>
__global__ __launch_bounds__(kIdxBlock) void direct_ranges_kernel(DirectIndexParams p) {
  __shared__ int s_c0;
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < p.n_stat_words; i += kIdxBlock) p.stats[i] = 0ull;
    if (threadIdx.x == 0 && p.err) *p.err = kNoError;
  }
  for (int i = blockIdx.x * kIdxBlock + threadIdx.x; i < p.n_tiles; i += gridDim.x * kIdxBlock) {
    p.tbegin_next[i] = 0xFFFFFFFFu;
    p.tend_next[i] = 0u;
  }
  const long long lo = (long long)blockIdx.x * kIdxRun;
  const long long hi = lo + kIdxRun < (long long)p.n_reads ? lo + kIdxRun : (long long)p.n_reads;
  if (lo >= hi) return;
  // the contig of the workgroup's first read, as the facts pass left it (the batch does not change: a binary search of nine
  // dependent loads per workgroup and pass was most of this kernel's 28 us); the threads walk on from there
  (void)s_c0;
  ContigCursor cur;
  {   // (the contig's row of the tables with it: one load, in flight together with the positions below)
    const DirectBlockCursor bc = p.block_contig[blockIdx.x];
    cur.c = bc.c; cur.begin = bc.begin; cur.next_begin = bc.next_begin; cur.tile_base = bc.tile_base; cur.tile_end = bc.tile_end;
    cur.clen = bc.clen;
  }
  const int lane = threadIdx.x & 63;
  constexpr int U = kIdxRun / kIdxBlock;
  int32_t pos[U], before[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {      // (all the loads first: one trip to memory)
    const long long ii = lo + (long long)u * kIdxBlock + threadIdx.x;
    const long long ic = ii < hi ? ii : hi - 1;
    pos[u] = p.pos[ic];
    before[u] = p.pos[ic > 0 ? ic - 1 : 0];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long ii = lo + (long long)u * kIdxBlock + threadIdx.x;
    const bool valid = ii < hi;
    uint32_t key = 0xFFFFFFFEu;
    int ka = 0, kb = 0;
    if (valid) {
      const int i = (int)ii;
      cur.advance(p, i);
      // the tiles of the read's (clamped) start and of start + reach, and those of the read before it
      const long long last = cur.clen - 1;
      long long pc = pos[u];
      pc = pc < 0 ? 0 : (pc > last ? last : pc);
      const long long pr = pc + p.reach < last ? pc + p.reach : last;
      ka = cur.tile_base + (int)(pc >> p.tile_shift);
      kb = cur.tile_base + (int)(pr >> p.tile_shift);
      if (SORTED) {
        const bool first_of_contig = i == cur.begin;
        long long pb = before[u];
        pb = pb < 0 ? 0 : (pb > last ? last : pb);
        const long long pbr = pb + p.reach < last ? pb + p.reach : last;
        const int ka0 = first_of_contig ? cur.tile_base : cur.tile_base + (int)(pb >> p.tile_shift);
        const int kb0 = first_of_contig ? cur.tile_base - 1 : cur.tile_base + (int)(pbr >> p.tile_shift);
        for (int t = ka0; t < ka; ++t) p.tend[t] = (uint32_t)i;            // the first read that starts behind tile t
        for (int t = kb0 + 1; t <= kb; ++t) p.tbegin[t] = (uint32_t)i;     // the first read that can reach tile t
        if (i == cur.next_begin - 1)                                        // the contig's last read: every tile from its own on ends here
          for (int t = ka; t < cur.tile_end; ++t) p.tend[t] = (uint32_t)i + 1u;
      } else {
        key = (uint32_t)ka;
      }
    }
    if (!SORTED) {
      // consecutive reads mostly share a tile: the first read of a run publishes the low bound, the last one the high
      // bound (read indices grow with the lane whatever the positions do: unsorted input makes more runs, never a wrong bound)
      const uint32_t key_before = __shfl_up(key, 1), key_after = __shfl_down(key, 1);
      if (valid) {
        const uint32_t i = (uint32_t)ii;
        if (lane == 0 || key_before != key) atomicMin(&p.tbegin[ka], i);
        if (lane == 63 || key_after != key) atomicMax(&p.tend[ka], i + 1u);
        for (int t = ka + 1; t <= kb; ++t) {       // the tiles it can reach into
          atomicMin(&p.tbegin[t], i);
          atomicMax(&p.tend[t], i + 1u);
        }
      }
    }
  }
}

// ---- once per batch: validation and the batch's numbers ------------------------------------------------------------------
struct Fields {
  long long so, so1, qo, qo1, co, co1;
  int32_t pos, l, nm;
};
__device__ __forceinline__ Fields load_fields(const DirectIndexParams& p, long long i) {
  Fields f;
  f.so = p.seq_off[i]; f.so1 = p.seq_off[i + 1];
  f.qo = p.qual_off[i]; f.qo1 = p.qual_off[i + 1];
  f.co = p.cigar_off[i]; f.co1 = p.cigar_off[i + 1];
  f.pos = p.pos[i]; f.l = p.l_seq[i]; f.nm = p.nm[i];
  return f;
}
// the layout checks every read passes before anything is read through its offsets (same rules as the packer)
__device__ __forceinline__ bool bad_layout(const DirectIndexParams& p, const Fields& f) {
  const long long l = f.l;
  return l < 0 || f.co1 - f.co < 0 || f.co < 0 || f.so < 0 || f.qo < 0 || f.so1 - f.so < (l + 1) / 2 || f.qo1 - f.qo < l ||
         f.so1 > p.seq_bytes || f.qo1 > p.qual_bytes || f.co1 > p.n_cigar;
}

__global__ __launch_bounds__(kIdxBlock) void direct_facts_kernel(DirectIndexParams p) {
  __shared__ unsigned long long red[4];
  __shared__ int s_c0;
  const long long lo = (long long)blockIdx.x * kIdxRun;
  const long long hi = lo + kIdxRun < (long long)p.n_reads ? lo + kIdxRun : (long long)p.n_reads;
  if (threadIdx.x == 0 && lo < hi) {
    s_c0 = contig_of_read(p, (int)lo);
    ContigCursor c0;
    c0.c = s_c0;
    c0.fetch(p);
    DirectBlockCursor bc;
    bc.c = c0.c; bc.begin = c0.begin; bc.next_begin = c0.next_begin; bc.tile_base = c0.tile_base; bc.tile_end = c0.tile_end; bc.pad = 0;
    bc.clen = c0.clen;
    p.block_contig[blockIdx.x] = bc;
  }
  __syncthreads();
  unsigned long long alg = 0, status = kNoError;
  uint32_t maxl = 0, maxspan = 0, unsorted = 0, general = 0, n_long = 0, n_out = 0, maxcommon = 0;
  ContigCursor cur;
  if (lo < hi) {
    cur.c = s_c0;
    cur.fetch(p);
  }
  for (long long ii = lo + threadIdx.x; ii < hi; ii += kIdxBlock) {
    const int i = (int)ii;
    cur.advance(p, i);
    const Fields f = load_fields(p, ii);
    const long long l = f.l, nc = f.co1 - f.co;
    {       // position order inside the contig (clamped starts, as the ranges kernel compares them)
      const long long last = cur.clen - 1;
      long long pc = f.pos, pb = p.pos[i > 0 ? i - 1 : 0];
      pc = pc < 0 ? 0 : (pc > last ? last : pc);
      pb = pb < 0 ? 0 : (pb > last ? last : pb);
      if (i != cur.begin && pb > pc) unsorted = 1u;
    }
    if (bad_layout(p, f)) {
      const unsigned long long s = ((unsigned long long)i << 8) | kPackBadLayout;
      status = s < status ? s : status;
      continue;
    }
    alg += (unsigned long long)((l + 1) / 2 + l + 4 * nc + 16);
    maxl = (uint32_t)l > maxl ? (uint32_t)l : maxl;
    if (l > kMaxLSeq || nc > kMaxField16 || f.nm > kMaxField16) {      // beyond the fast paths: the batch takes the long path (pileup_long.hip)
      n_long += 1u;
      continue;
    }
    CigarView cg;
    cg.load(p.rec ? reinterpret_cast<const uint32_t*>(p.payload + ((unsigned long long)p.rec[i].off8 << 3)) : p.cigar + f.co);
    // reference span: what the ranges kernel must reach over (sites from the read's start to its last aligned base)
    unsigned long long span = 0;
    for_each_op(cg, (uint32_t)nc, [&](uint32_t, uint32_t v) {
      const uint32_t op = v & 15u;
      if (op_is_match(op) || op == OP_D || op == OP_N) span += (unsigned long long)(v >> 4);
      return true;
    });
    span = span > 0x3FFFFFFFull ? 0x3FFFFFFFull : span;
    maxspan = (uint32_t)span > maxspan ? (uint32_t)span : maxspan;
    if (span > (unsigned long long)p.overhang) {
      // an outlier (a long deletion, an N skip): listed with the tiles behind its own that it reaches, so that the ranges pass
      // can keep to the common span; where it leaves its tile's overhang -- or reaches into tiles beyond the next -- the chunks
      // of those tiles are dealt tile by tile (tile_flag)
      n_out += 1u;
      const long long last = cur.clen - 1;
      long long pc = f.pos;
      pc = pc < 0 ? 0 : (pc > last ? last : pc);
      long long e = (long long)f.pos + (long long)span - 1;
      e = e > last ? last : e;
      if (e >= pc) {
        const uint32_t ka = (uint32_t)cur.tile_base + (uint32_t)(pc >> p.tile_shift), kb = (uint32_t)cur.tile_base + (uint32_t)(e >> p.tile_shift);
        if (kb > ka) {
          const uint32_t slot = atomicAdd(p.n_outliers_listed, 1u);
          if (slot < p.outlier_cap) p.outliers[slot] = DirectOutlier{(uint32_t)i, ka + 1u, kb, 0u};
        }
        const long long tile_start = (pc >> p.tile_shift) << p.tile_shift;
        if (e - tile_start >= (long long)(1 << p.tile_shift) + p.overhang)
          for (uint32_t t = ka; t <= kb; ++t) p.tile_flag[t] = 1;
      }
    } else {
      maxcommon = (uint32_t)span > maxcommon ? (uint32_t)span : maxcommon;
    }
    ReadShape sh;
    const bool fast = (uint32_t)nc <= 4u && decode_shape(cg.c0, cg.c1, cg.c2, cg.c3, (uint32_t)nc, (uint32_t)l, &sh) && f.nm >= 0 &&
                      f.pos >= 0 && (long long)f.pos < cur.clen;
    general += fast ? 0u : 1u;
  }
  alg = block_sum(alg, red);
  const unsigned long long ngen = block_sum((unsigned long long)general, red);
  const unsigned long long nlong = block_sum((unsigned long long)n_long, red);
  const unsigned long long bmax = block_max((unsigned long long)maxl, red);
  const unsigned long long bspan = block_max((unsigned long long)maxspan, red);
  const unsigned long long bcommon = block_max((unsigned long long)maxcommon, red);
  const unsigned long long nout = block_sum((unsigned long long)n_out, red);
  const unsigned long long any_unsorted = block_max((unsigned long long)unsorted, red);
  const unsigned long long worst = ~block_max(~status, red);       // (the lowest status word)
  if (threadIdx.x == 0) {
    DirectFacts* f = p.facts + (blockIdx.x % kDirectFactSlots);
    if (alg) atomicAdd(&f->alg_bytes, alg);
    if (ngen) atomicAdd(&f->n_general, (uint32_t)ngen);
    if (nlong) atomicAdd(&f->n_long, (uint32_t)nlong);
    if (bmax) atomicMax(&f->max_l, (uint32_t)bmax);
    if (bspan) atomicMax(&f->max_span, (uint32_t)bspan);
    if (bcommon) atomicMax(&f->max_span_common, (uint32_t)bcommon);
    if (nout) atomicAdd(&f->n_outliers, (uint32_t)nout);
    if (any_unsorted) atomicOr(&f->unsorted, 1u);
    if (worst != kNoError) atomicMin(&f->status, worst);
  }
}

// ---- once per batch: the direct layout (layout.h DirectRec + payload) ------------------------------------------------------
// A read's 41 bytes of columns in seven arrays become ONE 16-byte record and its CIGAR / SEQ / QUAL bytes ONE run of the
// payload (BAM's own order), so that the pileup kernel finds everything a lane needs behind one dwordx4 load.  Every read was
// validated by the facts pass before this runs.  Three launches: payload units per workgroup's reads, their scan (one
// workgroup), records + gather.
__device__ __forceinline__ uint32_t read_units(const DirectLayoutParams& p, long long i) {
  return direct_payload_units((uint32_t)p.l_seq[i], (uint32_t)(p.cigar_off[i + 1] - p.cigar_off[i]));
}

__global__ __launch_bounds__(kIdxBlock) void direct_layout_units_kernel(DirectLayoutParams p) {
  __shared__ unsigned long long red[4];
  const long long lo = (long long)blockIdx.x * kIdxRun;
  const long long hi = lo + kIdxRun < p.n_reads ? lo + kIdxRun : p.n_reads;
  unsigned long long u = 0;
  for (long long i = lo + threadIdx.x; i < hi; i += kIdxBlock) u += read_units(p, i);
  u = block_sum(u, red);
  if (threadIdx.x == 0) p.block_units[blockIdx.x] = u;
}

// exclusive scan of the workgroups' units, in place; [n_blocks] = all of them (one workgroup: a few thousand entries)
__global__ __launch_bounds__(kIdxBlock) void direct_layout_scan_kernel(unsigned long long* v, int n_blocks) {
  __shared__ unsigned long long part[kIdxBlock];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0ull;
  __syncthreads();
  for (int base = 0; base < n_blocks; base += kIdxBlock) {
    const int i = base + (int)threadIdx.x;
    const unsigned long long x = i < n_blocks ? v[i] : 0ull;
    part[threadIdx.x] = x;
    __syncthreads();
    for (int d = 1; d < kIdxBlock; d <<= 1) {       // (Hillis-Steele: 8 rounds of 256)
      const unsigned long long y = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0ull;
      __syncthreads();
      part[threadIdx.x] += y;
      __syncthreads();
    }
    const unsigned long long c = carry;
    if (i < n_blocks) v[i] = c + part[threadIdx.x] - x;
    __syncthreads();
    if (threadIdx.x == kIdxBlock - 1) carry = c + part[kIdxBlock - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) v[n_blocks] = carry;
}

typedef uint32_t lay_u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
// `len` bytes src -> dst by the 16 lanes of a group: whole 16-byte chunks by one (unaligned) load and store, the tail bytewise
__device__ __forceinline__ void group_copy(uint8_t* dst, const uint8_t* src, uint32_t len, int gl) {
  for (uint32_t k = (uint32_t)gl * 16u; k < len; k += 256u) {
    if (k + 16u <= len) {
      *reinterpret_cast<lay_u32x4_a1*>(dst + k) = *reinterpret_cast<const lay_u32x4_a1*>(src + k);
    } else {
      for (uint32_t j = k; j < len; ++j) dst[j] = src[j];
    }
  }
}

__global__ __launch_bounds__(kIdxBlock) void direct_layout_fill_kernel(DirectLayoutParams p) {
  constexpr int U = kIdxRun / kIdxBlock;
  __shared__ uint32_t s_off[kIdxRun];            // a read's first unit, relative to the workgroup's
  __shared__ uint32_t s_part[kIdxBlock];
  const long long lo = (long long)blockIdx.x * kIdxRun;
  const long long hi = lo + kIdxRun < p.n_reads ? lo + kIdxRun : p.n_reads;
  const unsigned long long base = p.block_units[blockIdx.x];
  // thread t owns the reads lo + t * U ... (read order = payload order)
  uint32_t units[U], sum = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long i = lo + (long long)threadIdx.x * U + u;
    units[u] = i < hi ? read_units(p, i) : 0u;
    sum += units[u];
  }
  s_part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < kIdxBlock; d <<= 1) {
    const uint32_t y = (int)threadIdx.x >= d ? s_part[threadIdx.x - d] : 0u;
    __syncthreads();
    s_part[threadIdx.x] += y;
    __syncthreads();
  }
  uint32_t run = s_part[threadIdx.x] - sum;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long i = lo + (long long)threadIdx.x * U + u;
    s_off[threadIdx.x * U + u] = run;
    if (i < hi) {
      const int32_t nm = p.nm[i];
      DirectRec r;
      r.pos = p.pos[i];
      r.l_nc = (uint32_t)p.l_seq[i] | ((uint32_t)(p.cigar_off[i + 1] - p.cigar_off[i]) << 16);
      r.nmq = (nm < 0 ? (uint32_t)kNmAbsent : (uint32_t)nm) | ((uint32_t)p.mapq[i] << 16);
      r.off8 = (uint32_t)(base + run);
      p.rec[i] = r;
    }
    run += units[u];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {      // the sentinel: where the payload ends
    DirectRec r;
    r.pos = 0; r.l_nc = 0u; r.nmq = 0u; r.off8 = (uint32_t)(base + s_part[kIdxBlock - 1]);
    p.rec[p.n_reads] = r;
  }
  __syncthreads();
  // the bytes: a group of 16 lanes per read
  const int grp = (int)(threadIdx.x >> 4), gl = (int)(threadIdx.x & 15);
  for (long long i = lo + grp; i < hi; i += kIdxBlock / 16) {
    const uint32_t l = (uint32_t)p.l_seq[i];
    const long long co = p.cigar_off[i];
    const uint32_t nc4 = 4u * (uint32_t)(p.cigar_off[i + 1] - co), sl = (l + 1u) >> 1;
    uint8_t* dst = p.payload + 8ull * (base + s_off[(int)(i - lo)]);
    group_copy(dst, reinterpret_cast<const uint8_t*>(p.cigar + co), nc4, gl);
    group_copy(dst + nc4, p.seq4 + p.seq_off[i], sl, gl);
    group_copy(dst + nc4 + sl, p.qual + p.qual_off[i], l, gl);
    const uint32_t used = nc4 + sl + l, room = (used + 7u) & ~7u;
    if (gl == 0)
      for (uint32_t j = used; j < room; ++j) dst[j] = 0;
  }
}

// every pass, behind the ranges kernel, when the batch has outliers: a tile an outlier reaches begins its stream no later than
// at that read
__global__ __launch_bounds__(256) void direct_outliers_kernel(const DirectOutlier* list, uint32_t n, uint32_t* tbegin) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= n) return;
  const DirectOutlier o = list[j];
  for (uint32_t t = o.t_first; t <= o.t_last; ++t) atomicMin(&tbegin[t], o.read);
}

}  // namespace

hipError_t launch_direct_outliers(const DirectOutlier* list, uint32_t n, uint32_t* tbegin, hipStream_t s) {
  if (n == 0u) return hipSuccess;
  hipLaunchKernelGGL(direct_outliers_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, list, n, tbegin);
  return hipGetLastError();
}

int direct_index_blocks(int64_t n_reads) { return n_reads > 0 ? (int)((n_reads + kIdxRun - 1) / kIdxRun) : 1; }

hipError_t launch_direct_facts(const DirectIndexParams& p, hipStream_t s) {
  hipLaunchKernelGGL(direct_facts_kernel, dim3(direct_index_blocks(p.n_reads)), dim3(kIdxBlock), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_direct_layout_sizes(const DirectLayoutParams& p, hipStream_t s) {
  const int nb = direct_index_blocks(p.n_reads);
  hipLaunchKernelGGL(direct_layout_units_kernel, dim3(nb), dim3(kIdxBlock), 0, s, p);
  hipLaunchKernelGGL(direct_layout_scan_kernel, dim3(1), dim3(kIdxBlock), 0, s, p.block_units, nb);
  return hipGetLastError();
}

hipError_t launch_direct_layout_fill(const DirectLayoutParams& p, hipStream_t s) {
  hipLaunchKernelGGL(direct_layout_fill_kernel, dim3(direct_index_blocks(p.n_reads)), dim3(kIdxBlock), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_direct_ranges(const DirectIndexParams& p, hipStream_t s) {
  // always launched (even with no reads): block 0 resets the counters and the error word, every block its share of the
  // other parity's ranges
  const int grid = direct_index_blocks(p.n_reads);
  if (p.sorted) hipLaunchKernelGGL(direct_ranges_kernel<true>, dim3(grid), dim3(kIdxBlock), 0, s, p);
  else hipLaunchKernelGGL(direct_ranges_kernel<false>, dim3(grid), dim3(kIdxBlock), 0, s, p);
  return hipGetLastError();
}

}  // namespace midas
