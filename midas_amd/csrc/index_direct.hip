// gfx950 index pass of the direct pileup path: the device-side `samtools index` for reads that stay where the BAM decoder
// put them (reference: midas/run/snps.py:130-137 index_bam; pysam's fetch(contig, 0, length) then walks the index).
//
// Three launches on one stream, every pass (they are inside the timed step):
//   direct_classify_kernel  one thread per read: validates the read's CSR extents (status + lowest read, as the packer
//                           does), decides its class (kernels.h), writes the one-word `info` of a class-0 read (leading
//                           clip, aligned length, trailing clip -- what query_alignment_sequence, midas/run/snps.py:145,
//                           needs of the CIGAR), publishes the lowest / highest class-0 read index touching every tile,
//                           counts the (general read, tile) entries per tile and appends the general reads to a list
//   direct_scan_kernel      one workgroup: entry offsets per tile (exclusive scan), the pass's totals
//   direct_fill_kernel      one thread per general read: its 48-byte descriptor into every tile it touches (the tile of its
//                           clamped start, which counts it, and every tile holding one of its aligned bases), with the
//                           clip lengths by pysam's rules and the IndexError condition of count_coverage precomputed
// No sort, no payload: the pileup kernel reads SEQ / QUAL / CIGAR where they are.
#include "direct_common.h"

namespace midas {

using namespace dev;
using namespace direct;

namespace {

constexpr int kClsBlock = 256;
constexpr int kClsRun = 4096;        // reads per workgroup (the list of its not-so-simple reads lives in LDS)

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
  int c, next_begin, tile_base;
  long long clen;
  __device__ __forceinline__ void fetch(const DirectIndexParams& p) {
    next_begin = p.contig_read_begin[c + 1];
    clen = p.contig_len[c];
    tile_base = p.contig_tile_base[c];
  }
  __device__ __forceinline__ void seek(const DirectIndexParams& p, int i) { c = contig_of_read(p, i); fetch(p); }
  __device__ __forceinline__ void advance(const DirectIndexParams& p, int i) {
    if (i < next_begin || c + 1 >= p.n_contigs) return;
    while (c + 1 < p.n_contigs && i >= p.contig_read_begin[c + 1]) ++c;
    fetch(p);
  }
};

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

// Class 0 or not: `H* S? (M|=|X)+ S? H*`, every length >= 1, the query length adding up, NM present, the read's start inside
// its contig.  On success *info = leading clip | aligned length << 10 | trailing clip << 21.
__device__ bool class0_info(const Fields& f, const CigarView& cg, long long clen, uint32_t* info) {
  const uint32_t nc = (uint32_t)(f.co1 - f.co);
  if (f.l < 1 || f.l > kMaxLSeq || f.nm < 0 || f.pos < 0 || !((long long)f.pos < clen) || nc == 0u) return false;
  uint32_t k = 0;
  while (k < nc && (cg[k] & 15u) == OP_H) { if ((cg[k] >> 4) == 0u) return false; ++k; }
  uint32_t lead = 0, trail = 0;
  if (k < nc && (cg[k] & 15u) == OP_S) { lead = cg[k] >> 4; if (lead == 0u) return false; ++k; }
  uint32_t e = nc;
  while (e > k && (cg[e - 1] & 15u) == OP_H) { if ((cg[e - 1] >> 4) == 0u) return false; --e; }
  if (e > k && (cg[e - 1] & 15u) == OP_S) { trail = cg[e - 1] >> 4; if (trail == 0u) return false; --e; }
  if (e <= k) return false;
  unsigned long long m = 0;
  for (uint32_t i = k; i < e; ++i) {
    const uint32_t v = cg[i];
    if (!op_is_match(v & 15u) || (v >> 4) == 0u) return false;
    m += v >> 4;
  }
  if ((unsigned long long)lead + m + trail != (unsigned long long)f.l) return false;
  *info = lead | ((uint32_t)m << kInfoAlenShift) | (trail << kInfoTrailShift);
  return true;
}

// The tiles a general read is entered in, in increasing order: the tile of its clamped start (it counts the read and
// reports its errors), then every tile that holds one of its aligned bases inside the contig.
template <class F>
__device__ void general_tiles(long long pos, uint32_t nc, const CigarView& cg, long long clen, int tile_shift, int tile_base, F emit) {
  long long pc = pos < 0 ? 0 : pos;
  pc = pc > clen - 1 ? clen - 1 : pc;
  long long last = pc >> tile_shift;
  emit(tile_base + (int)last);
  long long r = pos;
  for (uint32_t k = 0; k < nc; ++k) {
    const uint32_t v = cg[k], op = v & 15u;
    const long long len = (long long)(v >> 4);
    if (op_is_match(op)) {
      const long long a = r < 0 ? 0 : r, b = r + len < clen ? r + len : clen;
      if (a < b) {
        const long long tz = (b - 1) >> tile_shift;
        for (long long t = ((a >> tile_shift) > last ? (a >> tile_shift) : last + 1); t <= tz; ++t) emit(tile_base + (int)t);
        last = tz > last ? tz : last;
      }
      r += len;
    } else if (op == OP_D || op == OP_N) {
      r += len;
    }
  }
}

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

// append `value` of the lanes with `pred` to a list whose cursor is *count (one atomic per wave)
__device__ __forceinline__ void wave_append(bool pred, uint32_t value, uint32_t* count, uint32_t* list) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0ull) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (pred) list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = value;
}

// ---- 1. classify -------------------------------------------------------------------------------------------------------
// Two phases, like the packer's per-read kernels: phase A settles in a few dozen instructions the reads whose CIGAR is one
// match op of the read's length (most of what an end-to-end aligner writes) -- `info`, and the tile bounds published once
// per run of consecutive reads in the same tile; everything else goes onto the workgroup's list (LDS) and is taken by phase
// B with all lanes busy on the CIGAR grammar.
__global__ __launch_bounds__(kClsBlock) void direct_classify_kernel(DirectIndexParams p) {
  __shared__ unsigned long long red[4];
  __shared__ uint32_t s_later[kClsRun];
  __shared__ uint32_t s_nlater;
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < p.n_stat_words; i += kClsBlock) p.stats[i] = 0ull;
    if (threadIdx.x == 0) *p.err = kNoError;
  }
  for (int i = blockIdx.x * kClsBlock + threadIdx.x; i < p.n_tiles; i += gridDim.x * kClsBlock) {
    p.tbegin_next[i] = 0xFFFFFFFFu;
    p.tend_next[i] = 0u;
  }
  if (threadIdx.x == 0) s_nlater = 0u;
  __syncthreads();
  const long long lo = (long long)blockIdx.x * kClsRun;
  const long long hi = lo + kClsRun < (long long)p.n_reads ? lo + kClsRun : (long long)p.n_reads;
  const int lane = threadIdx.x & 63;
  unsigned long long alg = 0, entries = 0;
  uint32_t maxl = 0;
  ContigCursor cur;
  if (lo + threadIdx.x < hi) cur.seek(p, (int)(lo + threadIdx.x));
  for (long long base = lo; base < hi; base += kClsBlock) {       // phase A
    const long long ii = base + threadIdx.x;
    const bool valid = ii < hi;
    bool quick = false;
    uint32_t key = 0xFFFFFFFEu;
    int t1 = -1;
    if (valid) {
      const int i = (int)ii;
      const Fields f = load_fields(p, ii);
      cur.advance(p, i);
      const bool bad = bad_layout(p, f);
      const uint32_t c0 = bad ? 0u : p.cigar[f.co < p.n_cigar ? f.co : 0];
      const uint32_t l = (uint32_t)f.l;
      quick = !bad && f.co1 - f.co == 1 && l - 1u < (uint32_t)kMaxLSeq && f.nm >= 0 && f.nm <= kMaxField16 && f.pos >= 0 &&
              (long long)f.pos < cur.clen && op_is_match(c0 & 15u) && (c0 >> 4) == l;
      if (quick) {
        p.info[i] = l << kInfoAlenShift;
        const uint32_t start = (uint32_t)f.pos, room = (uint32_t)cur.clen - start;
        const uint32_t ln = l < room ? l : room;
        key = (uint32_t)cur.tile_base + (start >> p.tile_shift);
        const int te = cur.tile_base + (int)((start + ln - 1u) >> p.tile_shift);
        t1 = te != (int)key ? te : -1;
        alg += (unsigned long long)((l + 1u) / 2u + l + 4u + 16u);
        maxl = l > maxl ? l : maxl;
      }
    }
    // reads are position-sorted, so consecutive reads mostly share a tile: the first read of a run publishes the low bound,
    // the last one the high bound (read indices grow with the lane whatever the positions do: unsorted input makes more
    // runs, never a wrong bound)
    const uint32_t key_before = __shfl_up(key, 1), key_after = __shfl_down(key, 1);
    if (quick) {
      const uint32_t i = (uint32_t)ii;
      if (lane == 0 || key_before != key) atomicMin(&p.tbegin[key], i);
      if (lane == 63 || key_after != key) atomicMax(&p.tend[key], i + 1u);
      if (t1 >= 0) {       // it reaches into the next tile (at most one: a read is no longer than a tile)
        atomicMin(&p.tbegin[t1], i);
        atomicMax(&p.tend[t1], i + 1u);
      }
    }
    wave_append(valid && !quick, (uint32_t)ii, &s_nlater, s_later);
  }
  __syncthreads();
  const uint32_t n_later = s_nlater;
  for (uint32_t k0 = 0; k0 < n_later; k0 += kClsBlock) {      // phase B
    const uint32_t k = k0 + threadIdx.x;
    bool general = false;
    uint32_t gi = 0;
    if (k < n_later) {
      const int i = (int)s_later[k];
      gi = (uint32_t)i;
      const Fields f = load_fields(p, i);
      const long long l = f.l, nc = f.co1 - f.co;
      if (bad_layout(p, f)) {
        atomicMin(&p.facts->status, ((unsigned long long)i << 8) | kPackBadLayout);
        p.info[i] = kInfoGeneral;     // (the run fails: nothing reads it)
      } else if (l > kMaxLSeq || nc > kMaxField16 || f.nm > kMaxField16) {
        atomicMin(&p.facts->status, ((unsigned long long)i << 8) | kPackUnsupported);
        p.info[i] = kInfoGeneral;
      } else {
        CigarView cg;
        cg.load(p.cigar + f.co);
        ContigCursor at;
        at.seek(p, i);
        uint32_t info = 0;
        alg += (unsigned long long)((l + 1) / 2 + l + 4 * nc + 16);
        maxl = (uint32_t)l > maxl ? (uint32_t)l : maxl;
        if (class0_info(f, cg, at.clen, &info)) {
          p.info[i] = info;
          const uint32_t alen = (info >> kInfoAlenShift) & 2047u;
          const uint32_t start = (uint32_t)f.pos, room = (uint32_t)at.clen - start;
          const uint32_t ln = alen < room ? alen : room;
          const int ta = at.tile_base + (int)(start >> p.tile_shift), tz = at.tile_base + (int)((start + ln - 1u) >> p.tile_shift);
          atomicMin(&p.tbegin[ta], (uint32_t)i);
          atomicMax(&p.tend[ta], (uint32_t)i + 1u);
          if (tz != ta) {
            atomicMin(&p.tbegin[tz], (uint32_t)i);
            atomicMax(&p.tend[tz], (uint32_t)i + 1u);
          }
        } else {
          p.info[i] = kInfoGeneral;
          general = true;
          unsigned long long n = 0;
          general_tiles(f.pos, (uint32_t)nc, cg, at.clen, p.tile_shift, at.tile_base, [&](int t) { atomicAdd(&p.gcount[t], 1u); ++n; });
          entries += n;
        }
      }
    }
    wave_append(general, gi, &p.facts->n_general, p.gen_reads);
  }
  alg = block_sum(alg, red);
  entries = block_sum(entries, red);
  const unsigned long long bmax = block_max((unsigned long long)maxl, red);
  if (threadIdx.x == 0) {
    DirectFacts* f = p.facts + (blockIdx.x % kDirectFactSlots);
    if (alg) atomicAdd(&f->alg_bytes, alg);
    if (entries) atomicAdd(&f->n_entries, entries);
    if (bmax) atomicMax(&f->max_l, (uint32_t)bmax);
  }
}

// ---- 2. scan: entry offsets per tile, totals of the pass ----------------------------------------------------------------
constexpr int kScanBlock = 1024;
__global__ __launch_bounds__(kScanBlock) void direct_scan_kernel(DirectIndexParams p) {
  __shared__ unsigned long long s_part[kScanBlock / 64];
  __shared__ unsigned long long s_base[kScanBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = p.n_tiles + 1;
  const int per = (n + kScanBlock - 1) / kScanBlock;
  const int a = tid * per, b = a + per < n ? a + per : n;
  unsigned long long sum = 0;
  for (int i = a; i < b; ++i) sum += p.gcount[i];
  unsigned long long incl = sum;                      // inclusive scan over the wave
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  if (tid == 0) {
    unsigned long long acc = 0;
    for (int w = 0; w < kScanBlock / 64; ++w) { s_base[w] = acc; acc += s_part[w]; }
  }
  __syncthreads();
  unsigned long long run = s_base[wave] + incl - sum;
  for (int i = a; i < b; ++i) {
    p.goff[i] = (uint32_t)run;
    run += p.gcount[i];
  }
  // the pass's totals: the slots added up, then cleared for the next pass
  if (tid < kDirectFactSlots) {
    DirectFacts* f = p.facts + tid;
    unsigned long long alg = f->alg_bytes, ent = f->n_entries, mx = f->max_l;
    for (int d = 32; d >= 1; d >>= 1) {
      alg += __shfl_down(alg, d);
      ent += __shfl_down(ent, d);
      const unsigned long long o = __shfl_down(mx, d);
      mx = o > mx ? o : mx;
    }
    if (tid == 0) {
      p.totals->status = f->status;
      p.totals->alg_bytes = alg;
      p.totals->n_entries = ent;
      p.totals->n_general = f->n_general;
      p.totals->max_l = (uint32_t)mx;
      f->status = kNoError;
      f->n_general = 0u;
    }
    f->alg_bytes = 0ull;
    f->n_entries = 0ull;
    f->max_l = 0u;
  }
}

// ---- 3. fill: the descriptors of the general reads, tile by tile -------------------------------------------------------
__global__ __launch_bounds__(kClsBlock) void direct_fill_kernel(DirectIndexParams p) {
  const uint32_t n_gen = p.totals->n_general;
  for (uint32_t k = blockIdx.x * kClsBlock + threadIdx.x; k < n_gen; k += gridDim.x * kClsBlock) {
    const int i = (int)p.gen_reads[k];
    const Fields f = load_fields(p, i);
    const uint32_t nc = (uint32_t)(f.co1 - f.co);
    CigarView cg;
    cg.load(p.cigar + f.co);
    const int c = contig_of_read(p, i);
    const long long clen = p.contig_len[c];
    GenDesc d;
    d.idx = (uint32_t)i;
    d.pos = f.pos;
    d.l = (uint32_t)f.l;
    d.nc = nc;
    d.nm16 = f.nm < 0 ? (uint32_t)kNmAbsent : (uint32_t)f.nm;
    d.mapq = p.mapq[i];
    d.so = (unsigned long long)f.so; d.qo = (unsigned long long)f.qo; d.co = (unsigned long long)f.co;
    // [EXT] pysam query_alignment_start / _end -> len(aln.query_alignment_sequence) (midas/run/snps.py:145)
    const long long qs = query_start(cg, nc), qe = query_end(cg, nc, f.l);
    long long al = qe - qs;
    al = al < 0 ? 0 : al;
    d.align_len = (uint32_t)(al > 0xFFFF ? 0xFFFF : al);
    d.lead = (uint32_t)(qs > 0xFFFF ? 0xFFFF : qs);
    // the one case in which count_coverage raises IndexError for a kept read: a match op maps a query position
    // >= l_seq onto a site inside the contig
    uint32_t flags = f.nm < 0 ? kGenNoNm : 0u;
    {
      long long qpos = 0, rpos = f.pos;
      for (uint32_t j = 0; j < nc; ++j) {
        const uint32_t v = cg[j], op = v & 15u;
        const long long len = (long long)(v >> 4);
        if (op_is_match(op)) {
          if (qpos + len > (long long)f.l) {
            const long long qs2 = qpos > (long long)f.l ? qpos : (long long)f.l;
            const long long rs = rpos + (qs2 - qpos), re = rpos + len;
            if (rs < clen && re > 0) flags |= kGenOverrun;
          }
          qpos += len;
          rpos += len;
        } else if (op == OP_I || op == OP_S) {
          qpos += len;
        } else if (op == OP_D || op == OP_N) {
          rpos += len;
        }
      }
    }
    d.flags = flags;
    general_tiles(f.pos, nc, cg, clen, p.tile_shift, p.contig_tile_base[c], [&](int t) {
      const uint32_t left = atomicSub(&p.gcount[t], 1u);        // counts the tile's entries back to zero: ready for the next pass
      const long long slot = (long long)p.goff[t] + (long long)left - 1;
      if (slot >= 0 && slot < p.gdesc_capacity) gdesc_store(p.gdesc + (size_t)slot * kGenDescWords, d);
    });
  }
}

}  // namespace

hipError_t launch_direct_index(const DirectIndexParams& p, hipStream_t s) {
  // always launched (even with no reads): block 0 resets the counters and the error word, the scan publishes the totals
  const int grid = p.n_reads > 0 ? (int)(((long long)p.n_reads + kClsRun - 1) / kClsRun) : 1;
  hipLaunchKernelGGL(direct_classify_kernel, dim3(grid), dim3(kClsBlock), 0, s, p);
  hipLaunchKernelGGL(direct_scan_kernel, dim3(1), dim3(kScanBlock), 0, s, p);
  hipLaunchKernelGGL(direct_fill_kernel, dim3(256), dim3(kClsBlock), 0, s, p);
  return hipGetLastError();
}

}  // namespace midas
