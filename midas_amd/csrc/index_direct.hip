// gfx950 index pass of the direct pileup path: the device-side `samtools index` for reads that stay where the BAM decoder
// put them (reference: midas/run/snps.py:130-137 index_bam; pysam's fetch(contig, 0, length) then walks the index).
//
// Three launches on one stream, every pass (they are inside the timed step):
//   direct_classify_kernel  one thread per read: validates the read's CSR extents (status + lowest read, as the packer
//                           does), decides its class (kernels.h), writes the 20-byte index record of a class-0 read (leading
//                           clip, aligned length, trailing clip -- what query_alignment_sequence, midas/run/snps.py:145,
//                           needs of the CIGAR --, position, NM, mapq, byte offsets of its SEQ and QUAL), publishes the lowest / highest class-0 read index touching every tile,
//                           counts the (general read, tile) entries per tile and appends the general reads to a list
//   direct_scan_kernel      one workgroup: entry offsets per tile (exclusive scan), the pass's totals
//   direct_fill_kernel      one thread per general read: its 32-byte descriptor into every tile it touches (the tile of its
//                           clamped start, which counts it, and every tile holding one of its aligned bases), with the
//                           clip lengths by pysam's rules and the IndexError condition of count_coverage precomputed
// No sort, no payload: the pileup kernel reads SEQ / QUAL / CIGAR where they are.
#include "direct_common.h"

namespace midas {

using namespace dev;
using namespace direct;

namespace {

constexpr int kClsBlock = 256;
constexpr int kClsU = 4;            // reads a thread has in flight in phase A
constexpr int kClsRun = 2048;        // reads per workgroup (the list of its not-so-simple reads lives in LDS)

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
// its contig.  On success *info = leading clip | aligned length << 10 | trailing clip << 21.  One forward pass:
// stage 0 leading hard clips, 1 behind the leading soft clip, 2 in the matches, 3 behind the trailing soft clip, 4 trailing
// hard clips.
__device__ bool class0_info(const Fields& f, const CigarView& cg, long long clen, uint32_t* info) {
  const uint32_t nc = (uint32_t)(f.co1 - f.co);
  if (f.l < 1 || f.l > kMaxLSeq || f.nm < 0 || f.pos < 0 || !((long long)f.pos < clen) || nc == 0u) return false;
  uint32_t stage = 0, lead = 0, trail = 0;
  unsigned long long m = 0;
  bool ok = true;
  for_each_op(cg, nc, [&](uint32_t, uint32_t v) {
    const uint32_t op = v & 15u, len = v >> 4;
    if (len == 0u) ok = false;
    else if (op == OP_H) { if (stage == 2u || stage == 3u) stage = 4u; else if (stage != 0u && stage != 4u) ok = false; }
    else if (op == OP_S) { if (stage == 0u) { lead = len; stage = 1u; } else if (stage == 2u) { trail = len; stage = 3u; } else ok = false; }
    else if (op_is_match(op)) { if (stage <= 2u) { m += len; stage = 2u; } else ok = false; }
    else ok = false;
    return ok;
  });
  if (!ok || stage < 2u || (unsigned long long)lead + m + trail != (unsigned long long)f.l) return false;
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
  for_each_op(cg, nc, [&](uint32_t, uint32_t v) {
    const uint32_t op = v & 15u;
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
    return true;
  });
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

// What the pileup kernel needs of a general read besides its columns: the clip lengths by pysam's rules and the one case in
// which count_coverage raises IndexError for a kept read (a match op maps a query position >= l_seq onto a site inside the contig).
__device__ void general_facts(const Fields& f, const CigarView& cg, uint32_t nc, long long clen, bool pad_advances, GenDesc* d) {
  // [EXT] pysam query_alignment_start / _end -> len(aln.query_alignment_sequence) (midas/run/snps.py:145)
  long long qs = 0, qe = 0;
  query_bounds(cg, nc, f.l, &qs, &qe);
  long long al = qe - qs;
  al = al < 0 ? 0 : al;
  d->align_len = (uint32_t)(al > 2047 ? 2047 : al);      // (l_seq <= 1024)
  d->lead = (uint32_t)(qs > 2047 ? 2047 : qs);
  uint32_t flags = f.nm < 0 ? kGenNoNm : 0u;
  long long qpos = 0, rpos = f.pos;
  for_each_op(cg, nc, [&](uint32_t, uint32_t v) {
    const uint32_t op = v & 15u;
    const long long len = (long long)(v >> 4);
    if (op_is_match(op)) {
      if (qpos + len > (long long)f.l) {
        const long long qs2 = qpos > (long long)f.l ? qpos : (long long)f.l;
        const long long rs = rpos + (qs2 - qpos), re = rpos + len;
        if (rs < clen && re > 0) flags |= kGenOverrun;
      }
      qpos += len;
      rpos += len;
    } else if (op == OP_I || op == OP_S || (op == OP_P && pad_advances)) {
      qpos += len;
    } else if (op == OP_D || op == OP_N) {
      rpos += len;
    }
    return true;
  });
  // `H* S? (M|=|X)+ (I|D|N) (M|=|X)+ S? H*` with the query adding up -- one indel, most of what is not class 0 in an aligner's
  // output: the two match runs are described in the descriptor itself (kGenInline), the pileup kernel never fetches the CIGAR.
  // (A gap below the tile length cannot jump a whole tile: such a read's tiles are consecutive, the fill kernel never walks it.)
  {
    uint32_t stage = 0, lead = 0, trail = 0, m1 = 0, m2 = 0, ins = 0, del = 0;
    bool ok = f.l >= 1 && f.l <= kMaxLSeq && nc >= 3u;
    for_each_op(cg, nc, [&](uint32_t, uint32_t v) {
      const uint32_t op = v & 15u, len = v >> 4;
      if (len == 0u) ok = false;
      else if (op == OP_H) { if (stage == 4u || stage == 5u) stage = 6u; else if (stage != 0u && stage != 6u) ok = false; }
      else if (op == OP_S) { if (stage == 0u) { lead = len; stage = 1u; } else if (stage == 4u) { trail = len; stage = 5u; } else ok = false; }
      else if (op_is_match(op)) {
        if (stage <= 2u) { m1 += len; stage = 2u; } else if (stage == 3u || stage == 4u) { m2 += len; stage = 4u; } else ok = false;
      } else if (op == OP_I || op == OP_D || op == OP_N) {
        if (stage == 2u) { if (op == OP_I) ins = len; else del = len; stage = 3u; } else ok = false;
      } else ok = false;
      if (m1 > 1023u || m2 > 1023u) ok = false;
      return ok;
    });
    if (ok && stage >= 4u && ins <= 1023u && del <= 4000u && lead + m1 + ins + m2 + trail == (uint32_t)f.l &&
        d->align_len == m1 + ins + m2 && d->lead == lead) {
      flags |= kGenInline;
      d->co = (unsigned long long)(m1 | (ins << 10) | (del << 20));      // (in place of the CIGAR offset)
    }
  }
  d->flags = flags;
}

// ---- 1. classify -------------------------------------------------------------------------------------------------------
// Two phases, like the packer's per-read kernels: phase A settles in a few dozen instructions the reads whose CIGAR is one
// match op of the read's length (most of what an end-to-end aligner writes); everything else goes onto the workgroup's
// list (LDS) and is taken by phase B with all lanes busy on the CIGAR grammar.
//
// The per-tile ranges.  SORTED (the batch's first pass found every contig's reads in position order): no atomics at all --
// read i compares the tile of its start and the tile of `start + reach` (reach = the longest read of the batch: no class-0
// read is longer) with those of read i - 1 and, where they differ, writes the index i into the tiles in between:
//   tend[t]   = first read that starts behind tile t          tbegin[t] = first read whose start + reach gets to tile t
// so [tbegin, tend) holds every class-0 read touching t (and the few that end just short of it).  !SORTED: the lowest / highest
// index of the class-0 reads touching a tile by atomicMin / atomicMax, one pair per run of consecutive reads in a tile --
// exact in any order, but a device-scope atomic is a trip to the memory side of the fabric: 1.3 M of them were 0.25 ms of
// this kernel's 0.35 on configs[2].
constexpr int kGenWin = 64;          // tiles (from the workgroup's first) whose general entries are counted in LDS first

template <bool SORTED>
__global__ __launch_bounds__(kClsBlock) void direct_classify_kernel(DirectIndexParams p) {
  __shared__ unsigned long long red[4];
  __shared__ uint32_t s_later[kClsRun];
  __shared__ uint32_t s_nlater, s_ngen, s_gen_base;
  __shared__ uint32_t s_hist[kGenWin];
  __shared__ int s_crange[2];
  __shared__ int s_tile0;
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < p.n_stat_words; i += kClsBlock) p.stats[i] = 0ull;
    if (threadIdx.x == 0) {
      *p.err = kNoError;
      idxrec_store_idle(p.rec, (size_t)p.n_reads, 0u);      // the sentinels: what a lane without a read / an entry fetches
      if (p.gdesc) gdesc_store_idle(p.gdesc + (size_t)p.gdesc_capacity * kGenDescWords);
    }
  }
  for (int i = blockIdx.x * kClsBlock + threadIdx.x; i < p.n_tiles; i += gridDim.x * kClsBlock) {
    p.tbegin_next[i] = 0xFFFFFFFFu;
    p.tend_next[i] = 0u;
  }
  const long long lo = (long long)blockIdx.x * kClsRun;
  const long long hi = lo + kClsRun < (long long)p.n_reads ? lo + kClsRun : (long long)p.n_reads;
  if (threadIdx.x == 0) { s_nlater = 0u; s_ngen = 0u; }
  if (threadIdx.x < kGenWin) s_hist[threadIdx.x] = 0u;
  // the contigs this workgroup's reads lie in (one binary search each, not one per read: nine dependent loads)
  if (threadIdx.x == 64 && lo < hi) s_crange[0] = contig_of_read(p, (int)lo);
  if (threadIdx.x == 128 && lo < hi) s_crange[1] = contig_of_read(p, (int)(hi - 1));
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned long long alg = 0, entries = 0;
  uint32_t maxl = 0, unsorted = 0;
  ContigCursor cur;
  if (lo + threadIdx.x < hi) {
    cur.c = s_crange[0];
    cur.fetch(p);
    cur.advance(p, (int)(lo + threadIdx.x));
  }
  if (threadIdx.x == 0 && lo < hi) {       // first tile of the workgroup's reads: the window of the LDS entry counts
    long long pc = p.pos[lo];
    pc = pc < 0 ? 0 : (pc > cur.clen - 1 ? cur.clen - 1 : pc);
    s_tile0 = cur.tile_base + (int)(pc >> p.tile_shift);
  }
  // phase A, four reads per thread at a time: all their columns are requested first, then the first CIGAR word of each,
  // then they are settled one after the other (written the obvious way a read cost two dependent trips to memory, and
  // those trips -- not bandwidth -- were the kernel's duration)
  for (long long base = lo; base < hi; base += kClsBlock * kClsU) {
    Fields f[kClsU];
    uint32_t c0[kClsU];
    int32_t pos_before[kClsU];
    uint32_t mapq[kClsU];
    bool bad[kClsU];
#pragma unroll
    for (int u = 0; u < kClsU; ++u) {
      const long long ii = base + (long long)u * kClsBlock + threadIdx.x;
      const long long ic = ii < hi ? ii : hi - 1;
      f[u] = load_fields(p, ic);
      pos_before[u] = p.pos[ic > 0 ? ic - 1 : 0];
      mapq[u] = p.mapq[ic];
    }
#pragma unroll
    for (int u = 0; u < kClsU; ++u) {
      bad[u] = bad_layout(p, f[u]);
      c0[u] = p.cigar[(!bad[u] && f[u].co < p.n_cigar) ? f[u].co : 0];
    }
#pragma unroll
    for (int u = 0; u < kClsU; ++u) {
      const long long ii = base + (long long)u * kClsBlock + threadIdx.x;
      const bool valid = ii < hi;
      bool quick = false;
      uint32_t key = 0xFFFFFFFEu;
      int t1 = -1;
      if (valid) {
        const int i = (int)ii;
        cur.advance(p, i);
        const uint32_t l = (uint32_t)f[u].l;
        quick = !bad[u] && f[u].co1 - f[u].co == 1 && l - 1u < (uint32_t)kMaxLSeq && idxrec_fits(f[u].nm, (unsigned long long)f[u].so) &&
                f[u].pos >= 0 && (long long)f[u].pos < cur.clen && op_is_match(c0[u] & 15u) && (c0[u] >> 4) == l;
        if (quick) {
          idxrec_store(p.rec, (size_t)i, l << kInfoAlenShift, f[u].pos, (uint32_t)f[u].nm, mapq[u], (unsigned long long)f[u].so,
                       (unsigned long long)f[u].qo);
          alg += (unsigned long long)((l + 1u) / 2u + l + 4u + 16u);
          maxl = l > maxl ? l : maxl;
        }
        // the tiles of the read's (clamped) start and of start + reach, and those of the read before it
        const long long last = cur.clen - 1;
        long long pc = f[u].pos;
        pc = pc < 0 ? 0 : (pc > last ? last : pc);
        const int ka = cur.tile_base + (int)(pc >> p.tile_shift);
        const bool first_of_contig = i == cur.begin;
        long long pb = pos_before[u];
        pb = pb < 0 ? 0 : (pb > last ? last : pb);
        if (!first_of_contig && pb > pc) unsorted = 1u;
        if (SORTED) {
          const long long pr = pc + p.reach < last ? pc + p.reach : last;
          const int kb = cur.tile_base + (int)(pr >> p.tile_shift);
          const long long pbr = pb + p.reach < last ? pb + p.reach : last;
          const int ka0 = first_of_contig ? cur.tile_base : cur.tile_base + (int)(pb >> p.tile_shift);
          const int kb0 = first_of_contig ? cur.tile_base - 1 : cur.tile_base + (int)(pbr >> p.tile_shift);
          for (int t = ka0; t < ka; ++t) p.tend[t] = (uint32_t)i;            // the first read that starts behind tile t
          for (int t = kb0 + 1; t <= kb; ++t) p.tbegin[t] = (uint32_t)i;     // the first read that can reach tile t
          if (i == cur.next_begin - 1)                                        // the contig's last read: every tile from its own on ends here
            for (int t = ka; t < cur.tile_end; ++t) p.tend[t] = (uint32_t)i + 1u;
        } else if (quick) {
          const uint32_t start = (uint32_t)f[u].pos, room = (uint32_t)cur.clen - start;
          const uint32_t ln = l < room ? l : room;
          key = (uint32_t)ka;
          const int te = cur.tile_base + (int)((start + ln - 1u) >> p.tile_shift);
          t1 = te != ka ? te : -1;
        }
      }
      if (!SORTED) {
        // consecutive reads mostly share a tile: the first read of a run publishes the low bound, the last one the high
        // bound (read indices grow with the lane whatever the positions do: unsorted input makes more runs, never a wrong bound)
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
      }
      wave_append(valid && !quick, (uint32_t)ii, &s_nlater, s_later);
    }
  }
  __syncthreads();
  const uint32_t n_later = s_nlater;
  const int tile0 = s_tile0;
  for (uint32_t k0 = 0; k0 < n_later; k0 += kClsBlock) {      // phase B
    const uint32_t k = k0 + threadIdx.x;
    bool general = false;
    uint32_t gi = 0;
    GenDesc gd{};
    int g_first = 0, g_span = 0, g_contig = 0;
    if (k < n_later) {
      const int i = (int)s_later[k];
      gi = (uint32_t)i;
      const Fields f = load_fields(p, i);
      const long long l = f.l, nc = f.co1 - f.co;
      if (bad_layout(p, f)) {
        atomicMin(&p.facts->status, ((unsigned long long)i << 8) | kPackBadLayout);
        idxrec_store_idle(p.rec, (size_t)i, 0u);     // (the run fails: nothing reads it)
      } else if (l > kMaxLSeq || nc > kMaxField16 || f.nm > kMaxField16) {
        atomicMin(&p.facts->status, ((unsigned long long)i << 8) | kPackUnsupported);
        idxrec_store_idle(p.rec, (size_t)i, 0u);
      } else {
        CigarView cg;
        cg.load(p.cigar + f.co);
        ContigCursor at;
        at.c = s_crange[0];
        while (at.c < s_crange[1] && i >= p.contig_read_begin[at.c + 1]) ++at.c;
        at.fetch(p);
        uint32_t info = 0;
        alg += (unsigned long long)((l + 1) / 2 + l + 4 * nc + 16);
        maxl = (uint32_t)l > maxl ? (uint32_t)l : maxl;
        if (idxrec_fits(f.nm, (unsigned long long)f.so) && class0_info(f, cg, at.clen, &info)) {
          idxrec_store(p.rec, (size_t)i, info, f.pos, (uint32_t)f.nm, p.mapq[i], (unsigned long long)f.so, (unsigned long long)f.qo);
          if (!SORTED) {
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
          }
        } else {
          idxrec_store_idle(p.rec, (size_t)i, (uint32_t)at.c);
          general = true;
          // the read's descriptor, complete but for the tiles it goes to: the fill kernel only places it
          gd.idx = (uint32_t)i; gd.pos = f.pos; gd.l = (uint32_t)l; gd.nc = (uint32_t)nc;
          gd.nm16 = f.nm < 0 ? (uint32_t)kNmAbsent : (uint32_t)f.nm;
          gd.mapq = p.mapq[i];
          gd.so = (unsigned long long)f.so; gd.qo = (unsigned long long)f.qo; gd.co = (unsigned long long)f.co;
          general_facts(f, cg, (uint32_t)nc, at.clen, p.pad_advances != 0, &gd);
          g_contig = at.c;
          unsigned long long n = 0;
          bool consecutive = true;
          // entries per tile: counted in LDS for the tiles near the workgroup's reads, one global atomic per tile afterwards
          general_tiles(f.pos, (uint32_t)nc, cg, at.clen, p.tile_shift, at.tile_base, [&](int t) {
            const unsigned w = (unsigned)(t - tile0);
            if (w < (unsigned)kGenWin) atomicAdd(&s_hist[w], 1u); else atomicAdd(&p.gcount[t], 1u);
            if (n == 0) g_first = t; else if (t != g_first + (int)n) consecutive = false;
            ++n;
          });
          g_span = (consecutive && n <= 255) ? (int)n : 0;      // 0: the fill kernel walks the CIGAR itself
          entries += n;
        }
      }
    }
    // the workgroup's general reads, compacted over the list that has been consumed up to here
    {
      const unsigned long long mask = __ballot(general);
      if (mask != 0ull) {
        const int leader = __ffsll((long long)mask) - 1;
        uint32_t b0 = 0;
        if (lane == leader) b0 = atomicAdd(&s_ngen, (uint32_t)__popcll(mask));
        b0 = __shfl(b0, leader);
        // (slot b0 + rank <= k0 + rank: entries at or below the ones being read in this round are only overwritten by the
        // waves that have already read theirs -- write after the round's reads)
        const uint32_t slot = b0 + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        __syncthreads();
        if (general) {
          s_later[slot] = gi;
          if (p.gen_body) {      // (not on the batch's first pass, which sizes the array)
            uint32_t* body = p.gen_body + ((size_t)p.gen_base[blockIdx.x] + slot) * kGenBodyWords;
            gdesc_store(body, gd);
            reinterpret_cast<uint4*>(body)[2] = make_uint4((uint32_t)g_first, (uint32_t)g_span, (uint32_t)g_contig, gi);
          }
        }
      } else {
        __syncthreads();
      }
    }
    __syncthreads();
  }
  {
    const uint32_t n_gen = s_ngen;
    // the workgroup's general reads go to ITS stretch of the list (the fill kernel takes it workgroup by workgroup: reads of
    // one neighbourhood, so that their tiles fall into one LDS window there too)
    if (threadIdx.x == 0) {
      s_gen_base = (uint32_t)lo;
      p.gen_count[blockIdx.x] = n_gen;
      if (n_gen) atomicAdd(&p.facts[blockIdx.x % kDirectFactSlots].n_general, n_gen);
    }
    if (threadIdx.x < kGenWin && s_hist[threadIdx.x] && tile0 + (int)threadIdx.x <= p.n_tiles)
      atomicAdd(&p.gcount[tile0 + threadIdx.x], s_hist[threadIdx.x]);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_gen; k += kClsBlock) p.gen_reads[s_gen_base + k] = s_later[k];
  }
  alg = block_sum(alg, red);
  entries = block_sum(entries, red);
  const unsigned long long bmax = block_max((unsigned long long)maxl, red);
  const unsigned long long any_unsorted = block_max((unsigned long long)unsorted, red);
  if (threadIdx.x == 0) {
    DirectFacts* f = p.facts + (blockIdx.x % kDirectFactSlots);
    if (alg) atomicAdd(&f->alg_bytes, alg);
    if (entries) atomicAdd(&f->n_entries, entries);
    if (bmax) atomicMax(&f->max_l, (uint32_t)bmax);
    if (any_unsorted) atomicOr(&f->unsorted, 1u);
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
  // (up to 32 elements per thread are held in registers: one trip to memory for the loads, one for the stores)
  constexpr int kKeep = 32;
  uint32_t v[kKeep];
  unsigned long long sum = 0;
  if (per <= kKeep) {
#pragma unroll
    for (int j = 0; j < kKeep; ++j) v[j] = (j < per && a + j < n) ? p.gcount[a + j] : 0u;
#pragma unroll
    for (int j = 0; j < kKeep; ++j) sum += v[j];
  } else {
#pragma unroll 8
    for (int i = a; i < b; ++i) sum += p.gcount[i];
  }
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
  if (per <= kKeep) {
#pragma unroll
    for (int j = 0; j < kKeep; ++j) {
      if (j < per && a + j < n) p.goff[a + j] = (uint32_t)run;
      run += v[j];
    }
  } else {
#pragma unroll 8
    for (int i = a; i < b; ++i) {
      p.goff[i] = (uint32_t)run;
      run += p.gcount[i];
    }
  }
  // the pass's totals: the slots added up, then cleared for the next pass
  if (tid < kDirectFactSlots) {
    DirectFacts* f = p.facts + tid;
    unsigned long long alg = f->alg_bytes, ent = f->n_entries, mx = f->max_l, uns = f->unsorted, ngen = f->n_general;
    for (int d = 32; d >= 1; d >>= 1) {
      alg += __shfl_down(alg, d);
      ent += __shfl_down(ent, d);
      ngen += __shfl_down(ngen, d);   // (every slot holds the general reads of the workgroups that use it)
      const unsigned long long o = __shfl_down(mx, d);
      mx = o > mx ? o : mx;
      uns |= __shfl_down(uns, d);
    }
    if (tid == 0) {
      p.totals->status = f->status;
      p.totals->alg_bytes = alg;
      p.totals->n_entries = ent;
      p.totals->n_general = (uint32_t)ngen;
      p.totals->max_l = (uint32_t)mx;
      p.totals->unsorted = (uint32_t)uns;
      f->status = kNoError;
    }
    f->n_general = 0u;
    f->alg_bytes = 0ull;
    f->n_entries = 0ull;
    f->max_l = 0u;
    f->unsorted = 0u;
  }
}

// ---- 3. fill: the descriptors of the general reads, tile by tile -------------------------------------------------------
// Workgroup b places the general reads of classify workgroup b (reads of one neighbourhood): their descriptors come ready
// from the classify kernel (gen_body: descriptor, first tile, number of consecutive tiles), so a read costs one coalesced
// fetch here; its tile entries are ranked in LDS, the slots of a tile reserved with ONE returning atomic per tile and
// workgroup.  (One thread per read fetching its columns again -- nine scattered loads -- and one returning atomic per entry
// took 0.12-0.16 ms on configs[2]: dependent trips to memory, not bytes.)
constexpr int kFillKeep = 4;         // tile entries of a read ranked through LDS (a read has one or two; more go the direct way)
__global__ __launch_bounds__(kClsBlock) void direct_fill_kernel(DirectIndexParams p) {
  __shared__ uint32_t s_cnt[kGenWin], s_base[kGenWin];
  __shared__ int s_tile0;
  const uint32_t n_gen = p.gen_count[blockIdx.x];
  const size_t body0 = p.gen_base[blockIdx.x];
  for (uint32_t chunk = 0; chunk < n_gen; chunk += kClsBlock) {
    const uint32_t k = chunk + threadIdx.x;
    const bool act = k < n_gen;
    if (threadIdx.x < kGenWin) s_cnt[threadIdx.x] = 0u;
    uint4 b0 = make_uint4(0u, 0u, 0u, 0u), b1 = b0, b2 = b0;
    if (act) {
      const uint4* body = reinterpret_cast<const uint4*>(p.gen_body + (body0 + k) * kGenBodyWords);
      b0 = body[0]; b1 = body[1]; b2 = body[2];
    }
    const int t_first = (int)b2.x, t_span = (int)b2.y;
    if (threadIdx.x == 0) s_tile0 = t_first;
    __syncthreads();
    const int tile0 = s_tile0;
    auto place = [&](long long slot) {
      if (slot >= 0 && slot < p.gdesc_capacity) {
        uint4* q = reinterpret_cast<uint4*>(p.gdesc + (size_t)slot * kGenDescWords);
        q[0] = b0; q[1] = b1;
        p.gidx[slot] = b2.w;
      }
    };
    auto direct = [&](int t) {
      const uint32_t left = atomicSub(&p.gcount[t], 1u);
      place((long long)p.goff[t] + (long long)left - 1);
    };
    // the first entries of the read: rank inside the workgroup (LDS); anything else takes its slot directly
    int kept_tile[kFillKeep];
    uint32_t kept_rank[kFillKeep];
    int n_kept = 0;
    if (act) {
      auto entry = [&](int t, int ord) {
        const unsigned w = (unsigned)(t - tile0);
        if (ord < kFillKeep && w < (unsigned)kGenWin) {
          kept_tile[n_kept] = t;
          kept_rank[n_kept] = atomicAdd(&s_cnt[w], 1u);
          ++n_kept;
        } else {
          direct(t);
        }
      };
      if (t_span > 0) {
        for (int j = 0; j < t_span; ++j) entry(t_first + j, j);
      } else {      // tiles that are not consecutive (a long skip): walk the CIGAR again
        const uint32_t nc = b1.w >> 16;
        const unsigned long long co = (unsigned long long)b1.z | ((unsigned long long)((b0.x >> 22) & 0xFFu) << 32);
        CigarView cg;
        cg.load(p.cigar + co);
        const int c = (int)b2.z;
        int ord = 0;
        general_tiles((long long)(int32_t)b0.y, nc, cg, p.contig_len[c], p.tile_shift, p.contig_tile_base[c], [&](int t) { entry(t, ord); ++ord; });
      }
    }
    __syncthreads();
    if (threadIdx.x < kGenWin && s_cnt[threadIdx.x]) {       // counts a tile's entries back towards zero: ready for the next pass
      const uint32_t n = s_cnt[threadIdx.x];
      s_base[threadIdx.x] = atomicSub(&p.gcount[tile0 + threadIdx.x], n) - n;
    }
    __syncthreads();
    for (int e = 0; e < n_kept; ++e) {
      const int t = kept_tile[e];
      place((long long)p.goff[t] + (long long)s_base[t - tile0] + (long long)kept_rank[e]);
    }
    __syncthreads();      // s_cnt / s_tile0 are rewritten by the next chunk
  }
}

}  // namespace

int direct_index_blocks(int64_t n_reads) { return n_reads > 0 ? (int)((n_reads + kClsRun - 1) / kClsRun) : 1; }

hipError_t launch_direct_index(const DirectIndexParams& p, hipStream_t s) {
  // always launched (even with no reads): block 0 resets the counters and the error word, the scan publishes the totals
  const int grid = p.n_reads > 0 ? (int)(((long long)p.n_reads + kClsRun - 1) / kClsRun) : 1;
  if (p.sorted) hipLaunchKernelGGL(direct_classify_kernel<true>, dim3(grid), dim3(kClsBlock), 0, s, p);
  else hipLaunchKernelGGL(direct_classify_kernel<false>, dim3(grid), dim3(kClsBlock), 0, s, p);
  hipLaunchKernelGGL(direct_scan_kernel, dim3(1), dim3(kScanBlock), 0, s, p);
  if (p.gen_body) {
    hipLaunchKernelGGL(direct_fill_kernel, dim3(grid), dim3(kClsBlock), 0, s, p);       // one workgroup per classify workgroup
  } else {      // the batch's first pass only sizes things: nothing to place, the entry counts go back to zero
    hipError_t e = hipMemsetAsync(p.gcount, 0, ((size_t)p.n_tiles + 1) * 4, s);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}

}  // namespace midas
