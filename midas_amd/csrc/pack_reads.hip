// gfx950 device packer of the MIDAS SNP pileup: BAM-native SoA resident in HBM -> the device layout of layout.h.
//
// What it stands for in the reference (citations into /root/reference):
//   midas/run/snps.py:145       aln.query_alignment_sequence      (clip structure of the CIGAR, [EXT] pysam)
//   midas/run/snps.py:151,154   np.mean(aln.query_qualities)      (whole-read quality reduction, one wave reduction here)
//   midas/run/snps.py:194-199   count_coverage -> get_aligned_pairs(matches_only=True): the CIGAR walk that turns a
//                               read into gap-free match segments; only 'A','C','G','T' are counted
//   midas/run/snps.py:130-137   samtools index (the records are laid out in tile order for the index kernel)
//
// The layout produced is bit for bit the one of the host packer (pack.cpp), which stays as the CPU test mirror:
// tests/test_gpu_pack.py compares records, payload, input-order map and index keys of both.
//
// Pipeline (all on one stream, no host round trip in between once the buffers exist):
//   pack_plan_kernel     1 thread / read   validate, CIGAR -> number of device records (match segments cut at tile
//                                          boundaries, or one record that keeps its CIGAR); algorithmic bytes, longest read
//   [scan]                                 first device record of every read (exclusive sum, device_sort.hip)
//   pack_keys_kernel     1 thread / read   per record: sort key (tile, class, bank phase), payload size, index key and a
//                                          16-byte descriptor (position, read, query offset, length, the filter's numbers)
//   [radix sort]                           stable sort of (key, record) -- input order survives inside a key (device_sort.hip)
//   pack_bounds_kernel   1 thread / record first sorted position of every key
//   pack_dest_kernel     1 thread / record device position of every record: a tile's segment records are dealt round-robin
//                                          over their eight bank phases (layout.h), everything else keeps the sorted order;
//                                          index keys and the input-order map land in device order
//   [scan]                                 payload offsets in device order
//   pack_scatter_kernel  5 lanes / record  sum(qual) of the record's read by wave reduction, 4-bit SEQ -> call codes (SWAR on
//                                          nibbles), N-mask folded into the quality bytes, record + payload written in place
// Everything that is per READ and branchy (CIGAR grammar, clipping, tile cuts) runs one thread per read, 64 reads per
// wave; the scatter kernel is uniform data movement, one lane per 31 bases.
//
// HBM roofline of the scatter kernel (the dominant one): it reads the raw read (ceil(l/2) + l + 4 n_cigar + 16 B of
// fixed fields) and writes 32 B per 31 bases + 16 B of record.

#include "device_common.h"

namespace midas {

using namespace dev;

namespace {

constexpr int kPlanBlock = 256;
constexpr int kScatterBlock = 256;

__device__ __forceinline__ bool op_is_match(uint32_t op) { return op == OP_M || op == OP_EQ || op == OP_X; }
__device__ __forceinline__ bool op_is_clip(uint32_t op) { return op == OP_S || op == OP_H; }

// contig of read i: the last contig whose first read is <= i (empty contigs share their begin with the next one)
__device__ __forceinline__ int contig_of_read(const PackParams& p, int i) {
  int lo = 0, hi = p.n_contigs;   // first index in (0, n_contigs] with read_begin[idx] > i
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (p.contig_read_begin[mid] > i) hi = mid; else lo = mid + 1;
  }
  int c = lo - 1;
  c = c < 0 ? 0 : c;
  return c > p.n_contigs - 1 ? p.n_contigs - 1 : c;
}

// A read's CIGAR: the first four ops arrive with one 16-byte load (nearly every CIGAR is that short), the rest on demand.
struct CigarView {
  uint32_t c0, c1, c2, c3;
  const uint32_t* p;
  __device__ __forceinline__ void load(const uint32_t* q) {   // (the array has 64 bytes of slack behind its last op)
    p = q;
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(q);
    c0 = v.x; c1 = v.y; c2 = v.z; c3 = v.w;
  }
  __device__ __forceinline__ uint32_t operator[](uint32_t k) const {
    return k < 4u ? (k < 2u ? (k == 0u ? c0 : c1) : (k == 2u ? c2 : c3)) : p[k];
  }
};

struct ReadView {
  long long pos;
  uint32_t l, nc;
  int nm;
  CigarView cg;
  long long clen;
  int tile_len, tile_shift;   // tile_len == 1 << tile_shift
};

// Number of device records of a read served as match segments (1..kMaxPieces), or 0 when it keeps its CIGAR.
// Same rules as pack.cpp segment_plan + piece_plan: `H* S? (M|=|X|I|D|N)+ S? H*`, every length >= 1, the query length
// adding up, NM present and small, at most kMaxSegments gap-free runs, the first of them with a base inside the contig,
// at most kMaxPieces pieces after cutting at the tile boundaries.
__device__ int plan_read(const ReadView& r, uint32_t* align_total) {
  if (r.l < 1u || r.l > (uint32_t)kMaxSegField || r.nm < 0 || r.nm > kMaxSegField || r.pos < 0 || r.nc == 0u) return 0;
  if (!(r.pos < r.clen)) return 0;
  if (r.nc == 1u) {      // "<l>M", most of what an end-to-end aligner writes: the same answer without the walk, in 32 bits
    const uint32_t op = r.cg.c0 & 15u, len = r.cg.c0 >> 4;
    if (op_is_match(op) && len == r.l) {
      const uint32_t start = (uint32_t)r.pos, room = (uint32_t)r.clen - start;      // 0 <= pos < clen < 2^31
      const uint32_t ln = len < room ? len : room;
      const int np = (int)(((start + ln - 1u) >> r.tile_shift) - (start >> r.tile_shift)) + 1;
      if (np > kMaxPieces) return 0;
      *align_total = r.l;
      return np;
    }
  }
  uint32_t k = 0;
  while (k < r.nc && (r.cg[k] & 15u) == OP_H) { if ((r.cg[k] >> 4) == 0u) return 0; ++k; }
  uint32_t lead = 0, trail = 0;
  if (k < r.nc && (r.cg[k] & 15u) == OP_S) { lead = r.cg[k] >> 4; if (lead == 0u) return 0; ++k; }
  uint32_t e = r.nc;
  while (e > k && (r.cg[e - 1] & 15u) == OP_H) { if ((r.cg[e - 1] >> 4) == 0u) return 0; --e; }
  if (e > k && (r.cg[e - 1] & 15u) == OP_S) { trail = r.cg[e - 1] >> 4; if (trail == 0u) return 0; --e; }
  if (e <= k) return 0;
  long long q = lead, rr = 0;
  int nsegs = 0, npieces = 0;
  bool prev_match = false, bad = false;
  long long seg_r = 0, seg_len = 0;
  auto flush = [&]() {   // pieces of the finished segment
    long long start = r.pos + seg_r, len = seg_len;
    if (start + len > r.clen) len = r.clen - start;
    if (len <= 0) { if (nsegs == 1) bad = true; return; }
    npieces += (int)(((start + len - 1) >> r.tile_shift) - (start >> r.tile_shift)) + 1;   // 0 <= start: shifts divide
  };
  for (uint32_t i = k; i < e; ++i) {
    const uint32_t op = r.cg[i] & 15u;
    const long long len = r.cg[i] >> 4;
    if (len == 0) return 0;
    if (op_is_match(op)) {
      if (prev_match) {
        seg_len += len;
      } else {
        if (nsegs > 0) flush();
        if (nsegs == kMaxSegments) return 0;
        ++nsegs;
        seg_r = rr;
        seg_len = len;
      }
      q += len;
      rr += len;
      prev_match = true;
    } else if (op == OP_I) {
      q += len;
      prev_match = false;
    } else if (op == OP_D || op == OP_N) {
      rr += len;
      prev_match = false;
    } else {
      return 0;
    }
    if (q > (long long)r.l || rr > 0x7FFFFFFFll) return 0;
  }
  if (nsegs == 0 || q + trail != (long long)r.l) return 0;
  flush();
  if (bad || npieces > kMaxPieces || npieces == 0) return 0;
  *align_total = r.l - lead - trail;
  return npieces;
}

// The pieces of a read plan_read accepted, one after the other: its match segments (adjacent match ops merged), clipped
// to the contig and cut at the tile boundaries.
struct PieceIter {
  CigarView cg;
  uint32_t k, e;
  long long q, rr;       // query / reference offset of the op at k
  long long sq, sr, sl;  // what is left of the current segment
  long long pos, clen;
  int tile_len;
  __device__ void init(const ReadView& r) {
    cg = r.cg; pos = r.pos; clen = r.clen; tile_len = r.tile_len;
    k = 0;
    while (k < r.nc && (cg[k] & 15u) == OP_H) ++k;
    q = 0;
    if (k < r.nc && (cg[k] & 15u) == OP_S) { q = cg[k] >> 4; ++k; }
    e = r.nc;
    while (e > k && (cg[e - 1] & 15u) == OP_H) --e;
    if (e > k && (cg[e - 1] & 15u) == OP_S) --e;
    rr = 0; sq = sr = sl = 0;
  }
  __device__ bool next(int* qoff, long long* roff, int* len) {
    for (;;) {
      if (sl > 0) {
        const long long start = pos + sr;
        const long long room = tile_len - (start & (long long)(tile_len - 1));   // tile_len is a power of two, start >= 0
        const long long take = sl < room ? sl : room;
        *qoff = (int)sq; *roff = sr; *len = (int)take;
        sq += take; sr += take; sl -= take;
        return true;
      }
      while (k < e && !op_is_match(cg[k] & 15u)) {
        const uint32_t op = cg[k] & 15u;
        const long long len = cg[k] >> 4;
        if (op == OP_I) q += len; else rr += len;   // D / N (nothing else can be here: plan_read accepted the read)
        ++k;
      }
      if (k >= e) return false;
      const long long q0 = q, r0 = rr;
      long long len = 0;
      while (k < e && op_is_match(cg[k] & 15u)) { len += cg[k] >> 4; ++k; }
      q += len; rr += len;
      const long long start = pos + r0;
      if (start + len > clen) len = clen - start;
      if (len <= 0) continue;
      sq = q0; sr = r0; sl = len;
    }
  }
};

// Facts about a record that keeps its CIGAR (layout.h kRec* bits): clip structure, the one condition under which the
// reference would raise IndexError for a kept read, and its reference length.  pack.cpp cigar_flags.
__device__ uint32_t general_flags(const ReadView& r, long long* reflen, bool pad_advances) {
  uint32_t f = 0;
  if (r.nc > 0u) {
    uint32_t lead = 0;
    while (lead < r.nc && op_is_clip(r.cg[lead] & 15u)) ++lead;
    uint32_t trail = 0;
    while (trail + 1 < r.nc && op_is_clip(r.cg[r.nc - 1 - trail] & 15u)) ++trail;
    const bool lead_plain = lead == 0u || (lead == 1u && (r.cg[0] & 15u) == OP_S);
    const bool trail_plain = trail == 0u || (trail == 1u && (r.cg[r.nc - 1] & 15u) == OP_S);
    if (!lead_plain || !trail_plain) f |= kRecClipGeneric;
  }
  long long qpos = 0, rpos = r.pos;
  for (uint32_t k = 0; k < r.nc; ++k) {
    const uint32_t op = r.cg[k] & 15u;
    const long long len = r.cg[k] >> 4;
    if (op_is_match(op)) {
      if (qpos + len > (long long)r.l) {
        const long long qs = qpos > (long long)r.l ? qpos : (long long)r.l;
        const long long rs = rpos + (qs - qpos), rend = rpos + len;
        if (rs < r.clen && rend > 0) f |= kRecOverrun;
      }
      qpos += len;
      rpos += len;
    } else if (op == OP_I || op == OP_S || (op == OP_P && pad_advances)) {
      qpos += len;
    } else if (op == OP_D || op == OP_N) {
      rpos += len;
    }
  }
  *reflen = rpos - r.pos;
  return f;
}

// Sort key and index key of a record (pack.cpp set_keys; index_reads.hip reads the index key).
struct RecKeys { uint32_t sort_key, tile_key; int tile, reach; };
__device__ __forceinline__ RecKeys record_keys(long long start, long long reflen, bool simple, long long clen, int tile_shift, int tile_base) {
  long long pc = start < 0 ? 0 : start;
  pc = pc > clen - 1 ? clen - 1 : pc;
  long long pe = start + (reflen > 0 ? reflen : 1) - 1;
  pe = pe < pc ? pc : (pe > clen - 1 ? clen - 1 : pe);
  const uint32_t t0 = (uint32_t)(pc >> tile_shift);         // 0 <= pc <= pe < clen < 2^31
  const long long reach = (pe >> tile_shift) - (long long)t0;
  const uint32_t cls = reach > 0 ? 2u : (simple ? 0u : 1u);
  RecKeys k;
  k.tile = tile_base + (int)t0;
  k.reach = (int)(reach > 31 ? 31 : reach);
  k.sort_key = (uint32_t)k.tile * (uint32_t)kPackBinsPerTile + (cls == 0u ? (uint32_t)(pc & 7) : 7u + cls);
  k.tile_key = ((uint32_t)k.tile << 7) | ((uint32_t)k.reach << 2) | cls;
  return k;
}

// Contig of the reads a thread walks in increasing order: one binary search at the start, then a forward walk (reads are
// grouped by contig, so the walk almost never moves) -- no search on the critical path of every read.
struct ContigCursor {
  int c;
  int next_begin;     // read_begin[c + 1]
  long long clen;
  int tile_base;
  __device__ __forceinline__ void seek(const PackParams& p, int i) {
    c = contig_of_read(p, i);
    fetch(p);
  }
  __device__ __forceinline__ void fetch(const PackParams& p) {
    next_begin = p.contig_read_begin[c + 1];
    clen = p.contig_len[c];
    tile_base = p.contig_tile_base[c];
  }
  __device__ __forceinline__ void advance(const PackParams& p, int i) {
    if (i < next_begin || c + 1 >= p.n_contigs) return;
    while (c + 1 < p.n_contigs && i >= p.contig_read_begin[c + 1]) ++c;
    fetch(p);
  }
};

// Sum over the block, result in thread 0 (4 waves: shuffles, then four words of LDS).
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

// What the plan / keys kernels read of read i, in two rounds of loads: everything that hangs on the index alone, then the
// CIGAR (whose offset came with the first round).  Both kernels run the rounds two and one read AHEAD of the read they
// are working on: written the obvious way (check, then load the next thing the checks let through) a read cost eight
// dependent trips to memory, and with ten reads per thread that latency -- not bandwidth, not arithmetic -- was the
// kernels' whole duration (0.30 and 0.56 ms for 0.45 + 0.66 GB of loads on configs[2]).
struct ReadFields {
  long long so, so1, qo, qo1, co, co1;
  int32_t pos, l, nm;
  uint32_t first, mapq, nseg;     // keys kernel only
};
template <bool KEYS>
__device__ __forceinline__ ReadFields load_fields(const PackParams& p, long long i) {
  ReadFields f;
  f.so = p.seq_off[i]; f.so1 = p.seq_off[i + 1];
  f.qo = p.qual_off[i]; f.qo1 = p.qual_off[i + 1];
  f.co = p.cigar_off[i]; f.co1 = p.cigar_off[i + 1];
  f.pos = p.pos[i]; f.l = p.l_seq[i]; f.nm = p.nm[i];
  f.first = 0u; f.mapq = 0u; f.nseg = 0u;
  if (KEYS) { f.first = p.first[i]; f.mapq = p.mapq[i]; f.nseg = p.nseg[i]; }
  return f;
}
// (an offset the layout check is about to refuse must not be followed: the load goes to the array's start instead)
__device__ __forceinline__ const uint32_t* cigar_at(const PackParams& p, const ReadFields& f) {
  return p.cigar + ((f.co >= 0 && f.co <= p.n_cigar) ? f.co : 0);
}
__device__ __forceinline__ void view_of(const PackParams& p, const ReadFields& f, const CigarView& cg, long long clen, ReadView* r) {
  r->pos = f.pos;
  r->l = (uint32_t)f.l;
  r->nm = f.nm;
  r->nc = (uint32_t)(f.co1 - f.co);
  r->cg = cg;
  r->clen = clen;
  r->tile_len = p.tile_len;
  r->tile_shift = p.tile_shift;
}

// ---- 1. validate + count --------------------------------------------------------------------------------------------
// Both per-read kernels (this one and pack_keys_kernel) run in two phases.  Phase A walks the workgroup's reads and
// settles, in a few dozen instructions, the ones whose CIGAR is a single match op of the read's length -- most of what an
// end-to-end aligner writes; every other read (clips, indels, anything malformed) is put on the workgroup's list.  Phase B
// takes the list with all lanes busy.  Why: with both kinds in one loop nearly every wave holds at least one read of the
// second kind (8 % of the reads of configs[2] -> 99.5 % of the waves), so every wave executed the whole CIGAR machinery
// every iteration under a mask of a few lanes -- PMC: 920 vector + 960 scalar instructions per read-iteration in the keys
// kernel, the kernels' whole duration was instruction issue.
__device__ __forceinline__ void later_push(bool pred, uint32_t value, uint32_t* s_count, uint32_t* list) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0ull) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(s_count, (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (pred) list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = value;
}
// the layout checks every read passes before anything is read through its offsets
__device__ __forceinline__ bool bad_layout(const PackParams& p, const ReadFields& f) {
  const long long l = f.l;
  return l < 0 || f.co1 - f.co < 0 || f.co < 0 || f.so < 0 || f.qo < 0 || f.so1 - f.so < (l + 1) / 2 || f.qo1 - f.qo < l ||
         f.so1 > p.seq_bytes || f.qo1 > p.qual_bytes || f.co1 > p.n_cigar;
}
// "<l>M" on a well-formed read inside its contig: the number of tile pieces, 0 = not that kind of read
__device__ __forceinline__ int single_match_pieces(const PackParams& p, const ReadFields& f, uint32_t cigar0, long long clen) {
  const uint32_t l = (uint32_t)f.l;
  if (f.co1 - f.co != 1 || l - 1u >= (uint32_t)kMaxSegField || (uint32_t)f.nm > (uint32_t)kMaxSegField || f.pos < 0 ||
      !((long long)f.pos < clen))
    return 0;
  if (!op_is_match(cigar0 & 15u) || (cigar0 >> 4) != l) return 0;
  const uint32_t start = (uint32_t)f.pos, room = (uint32_t)clen - start;
  const uint32_t ln = l < room ? l : room;
  const int np = (int)(((start + ln - 1u) >> p.tile_shift) - (start >> p.tile_shift)) + 1;
  return np <= kMaxPieces ? np : 0;
}

// Grid-stride over the reads: a few thousand workgroups, so that the batch-wide sums cost one atomic per workgroup
// (one per wave -- 170 k same-address atomics on configs[2] -- took 5.9 ms; the kernel itself takes ~0.1).
__global__ __launch_bounds__(kPlanBlock) void pack_plan_kernel(PackParams p) {
  __shared__ unsigned long long red[4];
  __shared__ uint32_t s_later;
  unsigned long long alg = 0, recs = 0;
  uint32_t maxl = 0;
  const long long per = ((long long)p.n_reads + gridDim.x - 1) / gridDim.x;   // a workgroup owns a contiguous run of reads
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < (long long)p.n_reads ? lo + per : (long long)p.n_reads;
  uint32_t* const later = p.first + (lo < hi ? lo : 0);    // the list lives where the scan will write afterwards
  if (threadIdx.x == 0) s_later = 0u;
  __syncthreads();
  ContigCursor cur;
  if (lo + threadIdx.x < hi) cur.seek(p, (int)(lo + threadIdx.x));
  long long ii = lo + threadIdx.x;
  ReadFields f0{}, f1{};
  CigarView c0{};
  // (loads ahead of the thread's last read go to the batch's last read instead of sitting in a branch of their own)
  const long long last = (long long)p.n_reads - 1;
  auto ahead = [&](long long k) { return k < hi ? k : last; };
  if (ii < hi) {
    f0 = load_fields<false>(p, ii);
    f1 = load_fields<false>(p, ahead(ii + kPlanBlock));
    c0.load(cigar_at(p, f0));
  }
  for (; ii < hi; ii += kPlanBlock) {      // phase A
    const ReadFields f2 = load_fields<false>(p, ahead(ii + 2 * kPlanBlock));      // two reads ahead: the fields
    CigarView c1;
    c1.load(cigar_at(p, f1));                                     // one read ahead: its CIGAR
    const int i = (int)ii;
    cur.advance(p, i);
    const int np = (bad_layout(p, f0) || f0.l > kMaxLSeq) ? 0 : single_match_pieces(p, f0, c0.c0, cur.clen);
    if (np > 0) {
      p.nseg[i] = (uint8_t)np;
      p.cnt[i] = (uint32_t)np;
      recs += (unsigned long long)np;
      alg += (unsigned long long)((f0.l + 1) / 2 + f0.l + 4 + 16);
      maxl = (uint32_t)f0.l > maxl ? (uint32_t)f0.l : maxl;
    }
    later_push(np == 0, (uint32_t)i, &s_later, later);
    f0 = f1; f1 = f2; c0 = c1;
  }
  __syncthreads();
  const uint32_t n_later = s_later;
  for (uint32_t k = threadIdx.x; k < n_later; k += kPlanBlock) {      // phase B
    const int i = (int)later[k];
    const ReadFields f = load_fields<false>(p, i);
    const long long l = f.l, nc = f.co1 - f.co;
    uint32_t cnt = 1;
    uint8_t ns = 0;
    if (bad_layout(p, f)) {
      atomicMin(&p.facts->status, ((unsigned long long)i << 8) | kPackBadLayout);
    } else if (l > kMaxLSeq || nc > kMaxField16 || f.nm > kMaxField16) {
      atomicMin(&p.facts->status, ((unsigned long long)i << 8) | kPackUnsupported);
    } else {
      CigarView cg;
      cg.load(p.cigar + f.co);
      ContigCursor at_read;
      at_read.seek(p, i);
      ReadView r;
      view_of(p, f, cg, at_read.clen, &r);
      uint32_t at = 0;
      ns = (uint8_t)plan_read(r, &at);
      cnt = ns ? ns : 1u;
      alg += (unsigned long long)((l + 1) / 2 + l + 4 * nc + 16);
      maxl = (uint32_t)l > maxl ? (uint32_t)l : maxl;
    }
    p.nseg[i] = ns;
    p.cnt[i] = cnt;
    recs += cnt;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) p.cnt[p.n_reads] = 0u;
  alg = block_sum(alg, red);
  recs = block_sum(recs, red);
  const unsigned long long bmax = block_max((unsigned long long)maxl, red);
  if (threadIdx.x == 0) {
    PackFacts* f = p.facts + (blockIdx.x % kPackFactSlots);
    if (alg) atomicAdd(&f->alg_bytes, alg);
    if (recs) atomicAdd(&f->n_records, recs);
    if (bmax) atomicMax(&f->max_l, (uint32_t)bmax);
  }
}

// Record descriptor, 32 bytes per device record in input order (d0 of all records, then d1 of all records): everything the scatter kernel needs to start loading
// the record's bytes at once -- no per-read lookups on its critical path.
//   d0.x  position of the record's first base on the contig        d0.y  index of its read
//   d0.z  bases in the record | n_cigar field << 16                 d0.w  nm field | mapq << 16 | kRec* flags << 24
//         (ReadRec words 2 and 3 as they will be stored, still without the read's mean quality / "QUAL absent")
//   d1.x / d1.y  low words of the read's byte offsets into qual / seq4
//   d1.z  bits 0-7 / 8-15 their bits 32-39, bits 16-26 the record's first base in the read (query offset)
//   d1.w  index key of the record (tile << 7 | reach << 2 | class)
struct Desc { uint4 d0, d1; };
__device__ __forceinline__ Desc make_desc(long long pos, int read, int len, int qoff, uint32_t flags, uint32_t n16, uint32_t nm16,
                                          uint32_t mapq, long long qual_off, long long seq_off, uint32_t tile_key) {
  Desc d;
  d.d0 = make_uint4((uint32_t)(int32_t)pos, (uint32_t)read, (uint32_t)len | ((n16 & 0xFFFFu) << 16),
                    (nm16 & 0xFFFFu) | (mapq << 16) | (flags << 24));
  d.d1 = make_uint4((uint32_t)qual_off, (uint32_t)seq_off,
                    (uint32_t)((qual_off >> 32) & 0xFF) | ((uint32_t)((seq_off >> 32) & 0xFF) << 8) | ((uint32_t)qoff << 16), tile_key);
  return d;
}

// ---- 2. per record: sort key, payload size, index key, descriptor ------------------------------------------------------
struct KeysOut {
  const PackParams& p;
  unsigned long long bytes_sum = 0;
  __device__ __forceinline__ void emit(int read, uint32_t mapq, long long qo, long long so, uint32_t j, const RecKeys& k, uint32_t bytes,
                                       long long pos, int len, int qoff, uint32_t flags, uint32_t n16, uint32_t nm16) {
    p.sort_key[j] = k.sort_key;
    p.sort_val[j] = j;
    p.bytes8[j] = bytes >> 3;
    const Desc d = make_desc(pos, read, len, qoff, flags, n16, nm16, mapq, qo, so, k.tile_key);
    p.desc[j] = d.d0;                                  // two arrays of 16-byte halves: neighbouring threads write
    p.desc[(size_t)p.n_records + j] = d.d1;            // neighbouring 16 bytes (interleaved halves: PMC 916 MB written for 510)
    bytes_sum += bytes;
    for (int t = 1; t <= k.reach; ++t)    // reads a later tile will see as well (hot-spot planning)
      if (k.tile + t < p.n_tiles) atomicAdd(&p.tile_extra[k.tile + t], 1u);
  }
};

__global__ __launch_bounds__(kPlanBlock) void pack_keys_kernel(PackParams p) {
  __shared__ unsigned long long red[4];
  __shared__ uint32_t s_later;
  KeysOut out{p};
  const long long per = ((long long)p.n_reads + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < (long long)p.n_reads ? lo + per : (long long)p.n_reads;
  uint32_t* const later = p.dest + (lo < hi ? lo : 0);     // (n_records >= n_reads entries, written by pack_dest_kernel afterwards)
  if (threadIdx.x == 0) s_later = 0u;
  __syncthreads();
  ContigCursor cur;
  if (lo + threadIdx.x < hi) cur.seek(p, (int)(lo + threadIdx.x));
  long long ii = lo + threadIdx.x;
  ReadFields f0{}, f1{};
  // (loads ahead of the thread's last read go to the batch's last read instead of sitting in a branch of their own)
  const long long last = (long long)p.n_reads - 1;
  auto ahead = [&](long long k) { return k < hi ? k : last; };
  if (ii < hi) {
    f0 = load_fields<true>(p, ii);
    f1 = load_fields<true>(p, ahead(ii + kPlanBlock));
  }
  for (; ii < hi; ii += kPlanBlock) {      // phase A: reads served as the pieces of one match op (pack_plan_kernel's short cut)
    const ReadFields f2 = load_fields<true>(p, ahead(ii + 2 * kPlanBlock));
    const int i = (int)ii;
    cur.advance(p, i);
    const int ns = (int)f0.nseg;
    const bool quick = ns > 0 && f0.co1 - f0.co == 1;
    if (quick) {
      const uint32_t l = (uint32_t)f0.l, at = l;
      uint32_t start = (uint32_t)f0.pos, left = (uint32_t)cur.clen - start, q = 0;
      left = l < left ? l : left;
      for (int s = 0; s < ns; ++s) {
        const uint32_t room = (uint32_t)p.tile_len - (start & (uint32_t)(p.tile_len - 1));
        const uint32_t take = left < room ? left : room;
        // a piece lies inside one tile and one contig: record_keys in 32 bits, reach 0, class 0
        RecKeys rk;
        rk.tile = cur.tile_base + (int)(start >> p.tile_shift);
        rk.reach = 0;
        rk.sort_key = (uint32_t)rk.tile * (uint32_t)kPackBinsPerTile + (start & 7u);
        rk.tile_key = (uint32_t)rk.tile << 7;
        out.emit(i, f0.mapq, f0.qo, f0.so, f0.first + (uint32_t)s, rk, blob_bytes(take, 0u, (uint32_t)p.lane_bases), (long long)start,
                 (int)take, (int)q, (uint32_t)kRecSimple, l | ((at >> 6) << 10) | ((s == 0 ? 1u : 0u) << 14),
                 (uint32_t)f0.nm | ((at & 63u) << 10));
        start += take; q += take; left -= take;
      }
    }
    later_push(!quick, (uint32_t)i, &s_later, later);
    f0 = f1; f1 = f2;
  }
  __syncthreads();
  const uint32_t n_later = s_later;
  for (uint32_t k = threadIdx.x; k < n_later; k += kPlanBlock) {      // phase B: everything else, all lanes busy
    const int i = (int)later[k];
    const ReadFields f = load_fields<true>(p, i);
    CigarView cg;
    cg.load(p.cigar + f.co);
    ContigCursor at_read;
    at_read.seek(p, i);
    ReadView r;
    view_of(p, f, cg, at_read.clen, &r);
    const int tb = at_read.tile_base;
    const uint32_t j0 = f.first;
    const int ns = (int)f.nseg;
    if (ns == 0) {
      long long reflen = 0;
      const uint32_t flags = general_flags(r, &reflen, p.pad_advances != 0);
      out.emit(i, f.mapq, f.qo, f.so, j0, record_keys(r.pos, reflen, false, r.clen, p.tile_shift, tb),
               blob_bytes(r.l, r.nc, (uint32_t)p.lane_bases), r.pos, (int)r.l, 0, flags, r.nc,
               r.nm < 0 ? (uint32_t)kNmAbsent : (uint32_t)r.nm);
    } else {
      uint32_t at = 0;
      (void)plan_read(r, &at);   // aligned length of the whole read
      PieceIter it;
      it.init(r);
      int qoff, len;
      long long roff;
      for (int s = 0; s < ns && it.next(&qoff, &roff, &len); ++s)
        // read-level numbers of the filter, in every segment (layout.h): l_seq, aligned length, NM, "first segment"
        out.emit(i, f.mapq, f.qo, f.so, j0 + (uint32_t)s, record_keys(r.pos + roff, len, true, r.clen, p.tile_shift, tb),
                 blob_bytes((uint32_t)len, 0u, (uint32_t)p.lane_bases), r.pos + roff, len, qoff, (uint32_t)kRecSimple,
                 r.l | ((at >> 6) << 10) | ((s == 0 ? 1u : 0u) << 14), (uint32_t)r.nm | ((at & 63u) << 10));
    }
  }
  const unsigned long long bytes_sum = block_sum(out.bytes_sum, red);
  if (threadIdx.x == 0 && bytes_sum) atomicAdd(&p.facts[blockIdx.x % kPackFactSlots].blob_bytes, bytes_sum);
}

// ---- 3. first sorted position of every key ---------------------------------------------------------------------------
__global__ __launch_bounds__(kPlanBlock) void pack_bounds_kernel(PackParams p) {
  const int d = blockIdx.x * kPlanBlock + threadIdx.x;
  const int m = p.n_records;
  const int nbins = p.n_tiles * kPackBinsPerTile;
  if (m == 0) {
    for (int b = d; b <= nbins; b += gridDim.x * kPlanBlock) p.bin_start[b] = 0u;
    return;
  }
  if (d >= m) return;
  const long long k = p.key_sorted[d];
  const long long kp = d > 0 ? (long long)p.key_sorted[d - 1] : -1ll;
  for (long long b = kp + 1; b <= k; ++b) p.bin_start[b] = (uint32_t)d;
  if (d == m - 1)
    for (long long b = k + 1; b <= nbins; ++b) p.bin_start[b] = (uint32_t)m;
}

// ---- 4. device position of every record -----------------------------------------------------------------------------
// A tile's segment records (bins 0..7 = first site modulo 8) are dealt round-robin over their phases, so that the reads
// one pileup wave tallies together start in different LDS bank groups (layout.h / pack.cpp): the record with rank r in
// phase ph lands behind the first min(count, r) records of every phase and behind the rank-r records of the phases
// before its own.
__global__ __launch_bounds__(kPlanBlock) void pack_dest_kernel(PackParams p) {
  const int d = blockIdx.x * kPlanBlock + threadIdx.x;
  if (d < p.n_tiles) p.tile_reads[d] = p.tile_extra[d] + (p.bin_start[(d + 1) * kPackBinsPerTile] - p.bin_start[d * kPackBinsPerTile]);
  if (d >= p.n_records) return;
  const uint32_t key = p.key_sorted[d];
  const uint32_t j = p.val_sorted[d];
  const uint32_t tile = key / (uint32_t)kPackBinsPerTile, sub = key - tile * (uint32_t)kPackBinsPerTile;
  uint32_t dpos = (uint32_t)d;
  if (sub < 8u) {
    const uint32_t* bs = p.bin_start + (size_t)tile * kPackBinsPerTile;
    const uint32_t rank = (uint32_t)d - bs[sub];
    uint32_t before = 0;
    uint32_t prev = bs[0];
#pragma unroll
    for (uint32_t ph = 0; ph < 8u; ++ph) {
      const uint32_t nxt = bs[ph + 1];
      const uint32_t cnt = nxt - prev;
      before += cnt < rank ? cnt : rank;
      before += (ph < sub && cnt > rank) ? 1u : 0u;
      prev = nxt;
    }
    dpos = bs[0] + before;
  }
  p.dest[j] = dpos;
  p.bytes8_dev[dpos] = p.bytes8[j];
  if (d == 0) p.bytes8_dev[p.n_records] = 0u;
}

// ---- 5. scatter: records + payload -----------------------------------------------------------------------------------
typedef uint32_t u32_a1 __attribute__((aligned(1)));

__device__ __forceinline__ uint32_t low_bytes_mask(int hi) {
  return hi >= 4 ? 0xFFFFFFFFu : (hi <= 0 ? 0u : ((1u << (8 * hi)) - 1u));
}

// Eight 4-bit BAM base codes ("=ACMGRSVTWYHKDBN") -> eight nibbles 4 | code (A, C, G, T = 0..3) for A (1), C (2), G (4),
// T (8) and 0 for anything else: bit 2 of a nibble says "counts", bits 0-1 are the counter of the base.
__device__ __forceinline__ uint32_t call_codes8(uint32_t x) {
  const uint32_t pair = (x & 0x55555555u) + ((x >> 1) & 0x55555555u);
  const uint32_t pop = (pair & 0x33333333u) + ((pair >> 2) & 0x33333333u);   // bits set per nibble
  const uint32_t z = pop ^ 0x11111111u;                                       // 0 where exactly one bit is set
  const uint32_t nz = (z | (z >> 1) | (z >> 2) | (z >> 3)) & 0x11111111u;
  const uint32_t valid = nz ^ 0x11111111u;                                    // bit 0 of the nibbles of A/C/G/T
  const uint32_t hi = ((x >> 3) | (x >> 2)) & 0x11111111u;                    // G or T
  const uint32_t lo = ((x >> 3) | (x >> 1)) & 0x11111111u;                    // C or T
  // 7 in the nibbles of A/C/G/T.  (valid << 3) - valid would do, but the compiler turns that into v_mul_lo_u32 by 7:
  // a quarter-rate instruction; two shift-ors stay full rate
  const uint32_t seven = valid | (valid << 1) | (valid << 2);
  return ((valid << 2) | (hi << 1) | lo) & seven;
}
// Two bytes (four nibbles: byte0.hi, byte0.lo, byte1.hi, byte1.lo in base order) -> four bytes, one nibble each.
// HALF 0: bytes 0,1 of x; HALF 1: bytes 2,3.
template <int HALF>
__device__ __forceinline__ uint32_t spread_nibbles(uint32_t x) {
  const uint32_t pp = __builtin_amdgcn_perm(x, x, HALF ? 0x03030202u : 0x01010000u);
  return ((pp >> 4) & 0x000F000Fu) | (pp & 0x0F000F00u);
}

// One group of `lanes_per_read` lanes per device record (input order), lane c = payload chunk c: 31 (32) bases.
// Uniform work: every per-read decision was taken by pack_keys_kernel and travels in the 32-byte descriptor, byte offsets of
// the read included, so a wave waits for two rounds of loads (descriptor, bytes).  One batch of records per wave and as
// many waves as there are batches: persistent waves with a software prefetch of the next descriptors measured 20 % slower
// (fewer waves in flight), and 2 / 4 / 8 / 16 batches per wave in a plain loop 8 / 12 / 10 / 19 % slower than one
// (1.32 ms -> 1.42 / 1.49 / 1.45 / 1.57): the hardware's own wave switching hides the latency better.
__global__ __launch_bounds__(kScatterBlock) void pack_scatter_kernel(PackParams p) {
  const int lane = threadIdx.x & 63;
  const int lpr = p.lanes_per_read, rpw = 64 / lpr;
  const int g = lane / lpr, c = lane - g * lpr;
  const uint32_t lb = (uint32_t)p.lane_bases;
  const long long m = p.n_records;
  if (blockIdx.x == 0 && threadIdx.x == 0) {   // the sentinel record: where the payload ends
    uint4 s = make_uint4(0u, p.off8[p.n_records], 0u, (uint32_t)kRecSentinel << 24);
    reinterpret_cast<uint4*>(p.rec)[p.n_records] = s;
  }
  const long long batch = (long long)blockIdx.x * (kScatterBlock / 64) + (threadIdx.x >> 6);
  if (batch * rpw >= m) return;
  struct Head { uint4 d0, d1; uint32_t dest; };
  Head cur;
  {
    const long long jj = batch * rpw + g;
    const size_t j = (g < rpw && jj < m) ? (size_t)jj : 0;
    cur.d0 = p.desc[j];
    cur.d1 = p.desc[(size_t)m + j];
    cur.dest = p.dest[j];
  }
  {
    const bool valid = g < rpw && batch * rpw + g < m;
    const uint32_t d = cur.dest;
    const uint32_t o8 = p.off8[d];
    const uint32_t flags = (cur.d0.w >> 24) & 0xFu;
    const bool simple = (flags & kRecSimple) != 0u;
    const int plen = valid ? (int)(cur.d0.z & 0xFFFFu) : 0;
    const int pq = (int)((cur.d1.z >> 16) & 0x7FFu);
    const int l = valid ? (simple ? (int)((cur.d0.z >> 16) & 0x3FFu) : plen) : 0;   // a segment carries its read's l_seq
    const uint8_t* qsrc = p.qual + ((size_t)cur.d1.x | ((size_t)(cur.d1.z & 0xFFu) << 32));
    const uint8_t* ssrc = p.seq4 + ((size_t)cur.d1.y | ((size_t)((cur.d1.z >> 8) & 0xFFu) << 32));
    const bool whole = pq == 0 && plen == l;   // the record is its whole read: the payload's quality bytes are the read's

    // ---- this lane's 32 payload slots: bases [y0, y0 + nvalid) of the read ---------------------------------------------
    uint8_t* const b = p.blob + (size_t)o8 * 8;
    const uint32_t chunks = blob_chunks((uint32_t)plen, lb);
    const bool has = (uint32_t)c < chunks;
    const int x0 = c * (int)lb;
    const int nvalid = plen - x0 < (int)lb ? plen - x0 : (int)lb;
    const int y0 = pq + x0;
    u32x4_a1 qa, qb, sa;
    uint32_t s4 = 0;
    if (has) {
      qa = *reinterpret_cast<const u32x4_a1*>(qsrc + y0);
      qb = *reinterpret_cast<const u32x4_a1*>(qsrc + y0 + 16);
      const uint8_t* sp = ssrc + (y0 >> 1);
      sa = *reinterpret_cast<const u32x4_a1*>(sp);
      s4 = *reinterpret_cast<const u32_a1*>(sp + 16);
    }
    // ---- floor(mean quality) of the whole read (clipped bases included; np.mean(aln.query_qualities),
    // midas/run/snps.py:151): from the payload's own bytes when the record is the whole read, else one more pass over
    // the read's bytes; reduced over the group's lanes -----------------------------------------------------------------
    uint32_t part = 0;
    if (!whole && 32 * c < l) {
      const u32x4_a1 a = *reinterpret_cast<const u32x4_a1*>(qsrc + 32 * c);
      const u32x4_a1 bb = *reinterpret_cast<const u32x4_a1*>(qsrc + 32 * c + 16);
      const int nb = l - 32 * c;
      const uint32_t w[8] = {a.x, a.y, a.z, a.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) part = __builtin_amdgcn_sad_u8(w[k] & low_bytes_mask(nb - 4 * k), 0u, part);
      if (c == 0) part |= ((a.x & 0xFFu) == 0xFFu) ? 0x80000000u : 0u;   // QUAL absent (BAM: first byte 0xFF)
    }
    if (has) {
      uint32_t qw[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
      uint32_t tm[8];   // byte masks of the slots that hold bases of the record (the last lane of a record is partial)
#pragma unroll
      for (int k = 0; k < 8; ++k) tm[k] = low_bytes_mask(nvalid - 4 * k);
      if (whole) {
#pragma unroll
        for (int k = 0; k < 8; ++k) part = __builtin_amdgcn_sad_u8(qw[k] & tm[k], 0u, part);
        if (c == 0) part |= ((qa.x & 0xFFu) == 0xFFu) ? 0x80000000u : 0u;
      }
      uint32_t cw[4] = {sa.x, sa.y, sa.z, sa.w};
      if (y0 & 1) {   // the first base is a low nibble: shift the nibble stream by one
        const uint32_t nx[4] = {sa.y, sa.z, sa.w, s4};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t x = __builtin_amdgcn_alignbyte(nx[k], cw[k], 1);   // bytes 1..4 of {cw[k], nx[k]}
          cw[k] = ((cw[k] & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) cw[k] = call_codes8(cw[k]);
      // one byte per slot: {valid << 2 | code} of slots 4k .. 4k + 3 in word k
      const uint32_t cb[8] = {spread_nibbles<0>(cw[0]), spread_nibbles<1>(cw[0]), spread_nibbles<0>(cw[1]), spread_nibbles<1>(cw[1]),
                              spread_nibbles<0>(cw[2]), spread_nibbles<1>(cw[2]), spread_nibbles<0>(cw[3]), spread_nibbles<1>(cw[3])};
      // qualities above 62 (bit 7 of `over` in their bytes): rare enough that the clamp is a branch the wave usually skips
      uint32_t high = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) high |= (((qw[k] & 0x7F7F7F7Fu) + 0x41414141u) | qw[k]) & 0x80808080u & tm[k];
      if (__builtin_expect(high != 0u, 0)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t over = (((qw[k] & 0x7F7F7F7Fu) + 0x41414141u) | qw[k]) & 0x80808080u;
          const uint32_t om = (over << 1) - (over >> 7);                       // 0xFF in the bytes above 62
          qw[k] = (qw[k] & ~om) | (0x3E3E3E3Eu & om);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        // layout.h base_byte, four bases at a time: (min(q, 62) + 1) << 2 | code where the base is A/C/G/T and inside the
        // record, 0 elsewhere (padding slot, tail, other letters)
        const uint32_t vb = (cb[k] >> 2) & 0x01010101u;
        // 0xFF where the slot holds an A/C/G/T base: (vb << 8) - vb, with the shift spelt as a byte alignment so that the
        // compiler does not fold the pair into a quarter-rate multiply by 255
        const uint32_t keep = (__builtin_amdgcn_alignbyte(vb, 0u, 3u) - vb) & tm[k];
        qw[k] = (((qw[k] + 0x01010101u) << 2) | (cb[k] & 0x03030303u)) & keep;
      }
      // a quality above 62 among the record's bases (QUAL present): the batch will refuse a baseq above 62
      if (high && qsrc[0] != 0xFFu) atomicOr(&p.facts->high_qual, 1u);
      u32x4_a8 v0, v1;
      v0.x = qw[0]; v0.y = qw[1]; v0.z = qw[2]; v0.w = qw[3];
      v1.x = qw[4]; v1.y = qw[5]; v1.z = qw[6]; v1.w = qw[7];
      *reinterpret_cast<u32x4_a8*>(b + c * kChunk) = v0;
      *reinterpret_cast<u32x4_a8*>(b + c * kChunk + 16) = v1;
    }
    if (valid && !simple) {   // the record keeps its CIGAR behind the payload (zero padding to 8 bytes)
      const uint32_t nc = cur.d0.z >> 16;
      const uint32_t* cg = p.cigar + p.cigar_off[cur.d0.y];
      uint32_t* cd = reinterpret_cast<uint32_t*>(b + blob_cigar_off((uint32_t)l, lb));
      for (uint32_t k = (uint32_t)c; k < nc; k += (uint32_t)lpr) cd[k] = cg[k];
      if (c == 0 && (nc & 1u)) cd[nc] = 0u;
    }
    uint32_t qsum = 0;
    for (int cc = 0; cc < lpr; ++cc) qsum += __shfl(part, g * lpr + cc);
    if (valid && c == 0) {
      const uint32_t qflag = (qsum >> 31) ? (uint32_t)kRecQualAbsent : 0u;
      qsum &= 0x7FFFFFFFu;
      const uint32_t qmean = l > 0 ? qsum / (uint32_t)l : 0u;
      uint4 rec;
      rec.x = cur.d0.x;
      rec.y = o8;
      rec.z = cur.d0.z | ((qmean & 31u) << kRecLBits);
      rec.w = cur.d0.w | ((qflag | ((qmean >> 5) << 4)) << 24);
      reinterpret_cast<uint4*>(p.rec)[d] = rec;
      p.orig[d] = cur.d0.y;
      p.key_out[d] = cur.d1.w;
    }
  }
}

}  // namespace

size_t pack_sort_temp_bytes(int64_t max_records, int key_bits) {
  (void)key_bits;
  const size_t a = sort_scratch_words((long long)max_records), b = scan_scratch_words((long long)max_records + 1);
  return 4 * (a > b ? a : b) + 256;
}

int pack_key_bits(int32_t n_tiles) {
  unsigned long long nbins = (unsigned long long)(n_tiles > 0 ? n_tiles : 1) * kPackBinsPerTile;
  int bits = 1;
  while ((1ull << bits) < nbins) ++bits;
  return bits;
}

static int plan_grid(int n_reads) {   // grid-stride: enough workgroups to fill the chip, few enough for one atomic each
  const int need = (n_reads + kPlanBlock - 1) / kPlanBlock;
  return need < 1 ? 1 : (need > 4096 ? 4096 : need);
}

hipError_t launch_pack_plan(const PackParams& p, void* tmp, size_t tmp_bytes, hipStream_t s) {
  hipError_t e = hipMemsetAsync(p.facts, 0, sizeof(PackFacts) * kPackFactSlots, s);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(&p.facts->status, 0xFF, 8, s);   // (the status word of slot 0 is the batch's: errors are rare)
  if (e != hipSuccess) return e;
  if (p.n_reads > 0) {
    hipLaunchKernelGGL(pack_plan_kernel, dim3(plan_grid(p.n_reads)), dim3(kPlanBlock), 0, s, p);
    (void)tmp_bytes;
    e = launch_scan_u32(p.cnt, p.first, (long long)p.n_reads + 1, static_cast<uint32_t*>(tmp), s);
    if (e != hipSuccess) return e;
  } else {
    e = hipMemsetAsync(p.first, 0, 4, s);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}

hipError_t launch_pack_keys(const PackParams& p, hipStream_t s) {
  hipError_t e = hipMemsetAsync(p.tile_extra, 0, (size_t)(p.n_tiles > 0 ? p.n_tiles : 1) * 4, s);
  if (e != hipSuccess) return e;
  if (p.n_reads > 0)
    hipLaunchKernelGGL(pack_keys_kernel, dim3(plan_grid(p.n_reads)), dim3(kPlanBlock), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_pack_order(const PackParams& p, void* tmp, size_t tmp_bytes, int key_bits, hipStream_t s) {
  const int m = p.n_records;
  hipError_t e = hipSuccess;
  if (m > 0) {
    // (the library's own stable radix sort, device_sort.hip: the pairs end up in one of the two buffer pairs)
    uint32_t* ks = nullptr;
    uint32_t* vs = nullptr;
    e = launch_sort_pairs_u32(p.sort_key, p.sort_val, p.key_sorted, p.val_sorted, m, key_bits, static_cast<uint32_t*>(tmp), s, &ks, &vs);
    if (e != hipSuccess) return e;
    if (ks != p.key_sorted) {
      e = hipMemcpyAsync(p.key_sorted, ks, (size_t)m * 4, hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) return e;
      e = hipMemcpyAsync(p.val_sorted, vs, (size_t)m * 4, hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) return e;
    }
  }
  const int nbins = p.n_tiles * kPackBinsPerTile;
  const int gb = m > 0 ? (m + kPlanBlock - 1) / kPlanBlock : (nbins + kPlanBlock) / kPlanBlock;
  hipLaunchKernelGGL(pack_bounds_kernel, dim3(gb > 0 ? gb : 1), dim3(kPlanBlock), 0, s, p);
  const int nd = m > p.n_tiles ? m : p.n_tiles;
  if (nd > 0) hipLaunchKernelGGL(pack_dest_kernel, dim3((nd + kPlanBlock - 1) / kPlanBlock), dim3(kPlanBlock), 0, s, p);
  if (m == 0) {
    e = hipMemsetAsync(p.off8, 0, 4, s);
    if (e != hipSuccess) return e;
  } else {
    (void)tmp_bytes;
    e = launch_scan_u32(p.bytes8_dev, p.off8, (long long)m + 1, static_cast<uint32_t*>(tmp), s);
    if (e != hipSuccess) return e;
  }
  return hipGetLastError();
}

hipError_t launch_pack_scatter(const PackParams& p, hipStream_t s) {
  const int rpw = 64 / p.lanes_per_read;
  const long long batches = ((long long)p.n_records + rpw - 1) / rpw;
  long long blocks = (batches + (kScatterBlock / 64) - 1) / (kScatterBlock / 64);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(pack_scatter_kernel, dim3((unsigned)blocks), dim3(kScatterBlock), 0, s, p);
  return hipGetLastError();
}

}  // namespace midas
