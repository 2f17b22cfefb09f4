// gfx950 index kernel of the MIDAS SNP pileup: per-tile read ranges, the device-side `samtools index`
// (reference: midas/run/snps.py:130-137 index_bam; pysam's fetch(contig, 0, length) then walks the index).
#include "device_common.h"

namespace midas {

using namespace dev;

namespace {

// ------------------------------------------------------------------------------------------------
// Index kernel: one thread per read.  Finds the reference span (from the record alone for the
// common single-match CIGAR, else by walking the CIGAR) and records, for every tile the read
// overlaps, the lowest and highest read index seen.  The pileup kernel scans exactly that range per
// tile, so a long deletion in one read widens the scan of the tiles it really crosses and of no
// other.  Sortedness of the input only affects how tight these ranges are.  Block 0 also resets the
// per-species counters and the error word for this run (the tile ranges reset themselves: each
// tile ranges are double-buffered across runs: this kernel zeroes the other parity's for the next run).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kIndexBlock) void index_reads_kernel(IndexParams p) {
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < p.n_stat_words; i += kIndexBlock) p.stats[i] = 0ull;
    if (threadIdx.x == 0) *p.err = kNoError;
  }
  // the OTHER parity's tile ranges are zeroed here for the next run (this run's were zeroed by the previous one),
  // so the pileup kernel only ever reads its ranges and no per-run memset is needed
  for (int i = blockIdx.x * kIndexBlock + threadIdx.x; i < 3 * p.n_tiles; i += gridDim.x * kIndexBlock) {
    p.rbinv_next[i] = 0u;
    p.rend_next[i] = 0u;
  }
  const int i = blockIdx.x * kIndexBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = i < p.n_reads;
  int gt0 = -1, gt1 = -1;
  // contig of the block's first read: one bisection per block (uniform, scalar loads); the other threads walk on
  // from there -- reads are grouped by contig, so that is 0 steps for almost every thread
  int c0 = 0;
  {
    const int i0 = blockIdx.x * kIndexBlock;
    int lo = 0, hi = p.n_contigs;  // read_begin[lo] <= i0 < read_begin[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (p.contig_read_begin[mid] <= i0) lo = mid; else hi = mid;
    }
    c0 = lo;
  }
  // the record load does not depend on the contig: issue it first
  const uint4 r = reinterpret_cast<const uint4*>(p.rec)[valid ? i : p.n_reads];
  if (valid) {
    int lo = c0;
    while (lo + 1 < p.n_contigs && p.contig_read_begin[lo + 1] <= i) ++lo;
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
    // three read ranges per tile (slots 3t, 3t+1, 3t+2), matching the packer's order inside a tile window:
    //   S: simple reads that stay inside the tile          G: every other read that STARTS in the tile
    //   I: reads that start in an earlier tile and reach into this one ("incoming"; they sit at the end of
    //      their own tile's G run)
    const int t0 = tb + (int)((uint32_t)p0 / (uint32_t)p.tile_len);
    const int t1 = tb + (int)((uint32_t)p1 / (uint32_t)p.tile_len);
    gt0 = 3 * t0 + (((rec_flags(r) & kRecSimple) && t1 == t0) ? 0 : 1);
    gt1 = 3 * t1 + 2;   // last incoming slot
  }
  // Reads are (normally) sorted, so a wave mostly sees runs of one tile: only the first lane of a run
  // publishes the low bound and only the last one the high bound.
  const int prev = __shfl_up(gt0, 1);
  const int next = __shfl_down(gt0, 1);
  if (valid) {
    const uint32_t inv = (uint32_t)(p.n_reads - i);
    if (lane == 0 || prev != gt0) atomicMax(&p.rbinv[gt0], inv);
    if (lane == 63 || next != gt0) atomicMax(&p.rend[gt0], (uint32_t)(i + 1));
    for (int t = 3 * (gt0 / 3) + 5; t <= gt1; t += 3) {   // incoming slot of every later tile touched
      atomicMax(&p.rbinv[t], inv);
      atomicMax(&p.rend[t], (uint32_t)(i + 1));
    }
  }
}

}  // namespace

hipError_t launch_index_reads(const IndexParams& p, hipStream_t stream) {
  // always launched (even with no reads): block 0 resets the counters, the error word and the tile queue
  const int grid = p.n_reads > 0 ? (p.n_reads + kIndexBlock - 1) / kIndexBlock : 1;
  hipLaunchKernelGGL(index_reads_kernel, dim3(grid), dim3(kIndexBlock), 0, stream, p);
  return hipGetLastError();
}

}  // namespace midas
