// gfx950 pileup path for the reads the two fast paths do not take: l_seq above 1024, a CIGAR of more than 65 534 ops or an NM
// above 65 534 (layout.h kMaxLSeq / kMaxField16).  pysam has no such limits (the reference streams whatever the BAM holds,
// midas/run/snps.py:187-199), so a batch that holds one such read runs HERE instead of being refused: one thread per read,
// the CIGAR walked op by op and base by base straight from the caller's arrays, tallies by global atomics, keep_read's two
// ratio tests evaluated in fp64 as the reference writes them.  Built for exactness, not for speed -- long-read batches are not
// what MIDAS aligns (bowtie2, 100-250 bp) -- and as a second, structurally different implementation the parity tests hold the
// fast paths to.  Integer counting: no MFMA.
//
// Reference semantics (citations into /root/reference):
//   keep_read                       midas/run/snps.py:141-162
//   count_coverage call site        midas/run/snps.py:194-199  ([EXT] pysam: get_aligned_pairs(matches_only), qual >= quality_threshold,
//                                   only 'A','C','G','T' counted; IndexError when a kept read's match op runs past SEQ inside the contig)
//   depth / covered / total_depth   midas/run/snps.py:204-213 ; str(rec.seq).upper() :62
#include "direct_common.h"
#include "pileup_common.h"

namespace midas {

using namespace dev;
using namespace direct;

namespace {

constexpr int kLongBlock = 256;

__device__ __forceinline__ int long_contig_of(const LongParams& p, long long i) {
  int lo = 0, hi = p.n_contigs;
  while (lo < hi) {       // the last contig whose first read is <= i
    const int mid = (lo + hi) >> 1;
    if ((long long)p.contig_read_begin[mid] > i) hi = mid; else lo = mid + 1;
  }
  int c = lo - 1;
  c = c < 0 ? 0 : c;
  return c > p.n_contigs - 1 ? p.n_contigs - 1 : c;
}

// counts zeroed, counters at zero, error word at "no error"
__global__ __launch_bounds__(kLongBlock) void long_reset_kernel(LongParams p) {
  const size_t stride = (size_t)gridDim.x * kLongBlock;
  uint4* c4 = reinterpret_cast<uint4*>(p.out_counts);
  for (size_t i = (size_t)blockIdx.x * kLongBlock + threadIdx.x; i < (size_t)p.n_sites; i += stride) c4[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < p.n_stat_words; i += kLongBlock) p.stats[i] = 0ull;
    if (threadIdx.x == 0) *p.err = kNoError;
  }
}

__global__ __launch_bounds__(kLongBlock) void long_reads_kernel(LongParams p) {
  const long long i = (long long)blockIdx.x * kLongBlock + threadIdx.x;
  if (i >= p.n_reads) return;
  const int c = long_contig_of(p, i);
  const Tile t0 = p.tiles[p.contig_tile_base[c]];          // the contig's first tile: species, first site, piece flag
  const long long clen = t0.contig_len;
  const long long pos = p.pos[i];
  const long long l = p.l_seq[i];
  const long long co = p.cigar_off[i], nc = p.cigar_off[i + 1] - co;
  const uint32_t* cig = p.cigar + co;
  const uint8_t* qual = p.qual + p.qual_off[i];
  const uint8_t* seq4 = p.seq4 + p.seq_off[i];
  const int32_t nm = p.nm[i];
  const bool halo = t0.halo && pos < 0;                     // a read of the piece in front: counted and reported there
  unsigned long long* st = p.stats + (size_t)t0.species * MIDAS_STATS;
  if (!halo) atomicAdd(&st[MIDAS_STAT_ALIGNED], 1ull);
  auto raise = [&](uint32_t kind) { atomicMin(p.err, ((unsigned long long)i << 8) | kind); };
  // ---- keep_read (midas/run/snps.py:141-162), test by test in the reference's order ----------------------------------------
  if (l == 0) { if (!halo) raise(E_NO_SEQ); return; }
  long long qs = 0, qe = l;
  {
    long long k = 0;
    for (; k < nc; ++k) {                // [EXT] getQueryStart
      const uint32_t op = cig[k] & 15u;
      if (op == OP_H) continue;
      if (op == OP_S) qs += cig[k] >> 4; else break;
    }
    for (k = nc - 1; k >= 1; --k) {      // [EXT] getQueryEnd: index 0 is never looked at
      const uint32_t op = cig[k] & 15u;
      if (op == OP_H) continue;
      if (op == OP_S) qe -= cig[k] >> 4; else break;
    }
  }
  long long al = qe - qs;
  al = al < 0 ? 0 : al;
  if (nm < 0) { if (!halo) raise(E_NO_NM); return; }
  if (al == 0) { if (!halo) raise(E_ZERO_ALIGN); return; }
  if ((double)(100ll * (al - (long long)nm)) / (double)al < p.mapid) return;
  if (qual[0] == 0xFFu) { if (!halo) raise(E_NO_QUAL); return; }
  {
    unsigned long long sum = 0;          // (np.mean: a float64 sum of small integers is exact)
    for (long long k = 0; k < l; ++k) sum += qual[k];
    if ((double)sum / (double)l < (double)p.readq) return;
  }
  if ((int)p.mapq[i] < p.mapq_min) return;
  if ((double)al / (double)l < p.aln_cov) return;
  if (!halo) atomicAdd(&st[MIDAS_STAT_MAPPED], 1ull);
  // ---- the walk ([EXT] get_aligned_pairs(matches_only=True) + count_coverage's body) ----------------------------------------
  // pysam raises IndexError at the first match position that lies inside the contig with a query position >= l_seq: the tallies
  // of such a read are discarded with everything else, so the order in which its bases are added does not matter.
  uint32_t* counts = p.out_counts + 4 * (size_t)t0.site_base;
  long long qpos = 0, rpos = pos;
  for (long long k = 0; k < nc; ++k) {
    const uint32_t v = cig[k], op = v & 15u;
    const long long len = (long long)(v >> 4);
    if (op_is_match(op)) {
      long long lo = rpos < 0 ? -rpos : 0, hi = clen - rpos < len ? clen - rpos : len;      // offsets of the op inside the contig
      for (long long j = lo; j < hi; ++j) {
        const long long q = qpos + j;
        if (q >= l) { raise(E_CIGAR_OVERRUN); return; }
        if (p.baseq == 0 || (int)qual[q] >= p.baseq) {
          const uint32_t b = seq4[q >> 1], code = (q & 1) ? (b & 15u) : (b >> 4);
          const uint32_t slot = code == 1u ? 0u : (code == 2u ? 1u : (code == 4u ? 2u : (code == 8u ? 3u : 4u)));
          if (slot < 4u) atomicAdd(&counts[4 * (size_t)(rpos + j) + slot], 1u);
        }
      }
      qpos += len;
      rpos += len;
    } else if (op == OP_I || op == OP_S || (op == OP_P && p.pad_advances)) {
      qpos += len;
    } else if (op == OP_D || op == OP_N) {
      rpos += len;
    }
  }
}

// per site: depth, covered, the reference letter upper-cased; per species the sums (midas/run/snps.py:201-213)
__global__ __launch_bounds__(kLongBlock) void long_sites_kernel(LongParams p) {
  __shared__ unsigned long long red[4];
  const int tile = (int)blockIdx.x;
  const Tile t = p.tiles[tile];
  const uint4* c4 = reinterpret_cast<const uint4*>(p.out_counts) + t.site_base;
  unsigned long long depth = 0, cov = 0;
  for (int s = threadIdx.x; s < t.len; s += kLongBlock) {
    const uint4 v = c4[s];
    const unsigned long long d = (unsigned long long)v.x + v.y + v.z + v.w;
    depth += d;
    cov += d > 0 ? 1u : 0u;
    if (p.out_allele) {
      uint32_t ch = p.ref[t.site_base + s];
      if (ch >= 'a' && ch <= 'z') ch -= 32u;
      p.out_allele[t.site_base + s] = (uint8_t)ch;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) { depth += __shfl_down(depth, d); cov += __shfl_down(cov, d); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = depth; }
  __syncthreads();
  const unsigned long long dsum = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = cov; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long csum = red[0] + red[1] + red[2] + red[3];
    unsigned long long* st = p.stats + (size_t)t.species * MIDAS_STATS;
    if (dsum) atomicAdd(&st[MIDAS_STAT_DEPTH], dsum);
    if (csum) atomicAdd(&st[MIDAS_STAT_COVERED], csum);
  }
}

}  // namespace

hipError_t launch_pileup_long(const LongParams& p, hipStream_t s) {
  const long long sites = p.n_sites > 0 ? p.n_sites : 1;
  const unsigned reset_grid = (unsigned)((sites + kLongBlock - 1) / kLongBlock < 4096 ? (sites + kLongBlock - 1) / kLongBlock : 4096);
  hipLaunchKernelGGL(long_reset_kernel, dim3(reset_grid), dim3(kLongBlock), 0, s, p);
  if (p.n_reads > 0)
    hipLaunchKernelGGL(long_reads_kernel, dim3((unsigned)((p.n_reads + kLongBlock - 1) / kLongBlock)), dim3(kLongBlock), 0, s, p);
  if (p.n_tiles > 0) hipLaunchKernelGGL(long_sites_kernel, dim3((unsigned)p.n_tiles), dim3(kLongBlock), 0, s, p);
  return hipGetLastError();
}

}  // namespace midas
