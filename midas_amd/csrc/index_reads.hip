// gfx950 index kernel of the MIDAS SNP pileup: per-tile read ranges, the device-side `samtools index`
// (reference: midas/run/snps.py:130-137 index_bam; pysam's fetch(contig, 0, length) then walks the index).
#include "device_common.h"

namespace midas {

using namespace dev;

namespace {

// ------------------------------------------------------------------------------------------------
// Index kernel: one thread per read, every pass.  Input is one 32-bit key per record, written by the packer
// when it lays the records out in tile order:  key = tile << 7 | reach << 2 | class
//   class 0: simple read inside one tile (range S)    1: other read inside one tile (range G)
//   class 2: read reaching `reach` tiles further (range G of its own tile, range I of every tile it reaches);
//            reach == 31 means "31 or more": those (a > 100 kb deletion / skip) take the slow path, which walks
//            the CIGAR for the exact last tile.
// Records, for every (tile, range) slot, the lowest and highest read index seen (atomicMax on n_reads - index
// and index + 1), wave-aggregated over runs of equal slot.  The pileup kernel scans exactly those ranges, so a
// long deletion in one read widens the scan of the tiles it really crosses and of no other; unsorted input only
// makes ranges wider, never wrong.  Block 0 also resets the per-species counters and the error word; the tile
// ranges are double-buffered across runs: this kernel zeroes the other parity's for the next run, so a pass
// needs no memsets.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kIndexBlock) void index_reads_kernel(IndexParams p) {
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < p.n_stat_words; i += kIndexBlock) p.stats[i] = 0ull;
    if (threadIdx.x == 0) *p.err = kNoError;
  }
  for (int i = blockIdx.x * kIndexBlock + threadIdx.x; i < 3 * p.n_tiles; i += gridDim.x * kIndexBlock) {
    p.rbinv_next[i] = 0u;
    p.rend_next[i] = 0u;
  }
  const int i = blockIdx.x * kIndexBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = i < p.n_reads;
  int slot = -1, t0 = 0, reach = 0;
  if (valid) {
    const uint32_t key = p.key[i];
    const int cls = (int)(key & 3u);
    reach = (int)((key >> 2) & 31u);
    t0 = (int)(key >> 7);
    slot = 3 * t0 + (cls == 0 ? 0 : 1);
    if (reach == 31 && !(rec_flags(reinterpret_cast<const uint4*>(p.rec)[i]) & kRecSimple)) {
      // exact last tile of a very long record: walk its CIGAR (contig from the tile table)
      const uint4 r = reinterpret_cast<const uint4*>(p.rec)[i];
      const uint32_t* cig = reinterpret_cast<const uint32_t*>(p.blob + (size_t)rec_off8(r) * 8 +
                                                              blob_cigar_off((uint32_t)rec_l(r), (uint32_t)p.lane_bases));
      long long reflen = 0;
      const int n = rec_n(r);
      for (int k = 0; k < n; ++k) {
        const uint32_t v = cig[k];
        const uint32_t op = v & 15u;
        if (consumes_both(op) || op == OP_D || op == OP_N) reflen += (long long)(v >> 4);
      }
      const Tile tl = p.tiles[t0];
      const long long clen = tl.contig_len;
      long long p0 = rec_pos(r);
      p0 = p0 < 0 ? 0 : (p0 > clen - 1 ? clen - 1 : p0);
      long long p1 = (long long)rec_pos(r) + (reflen > 0 ? reflen : 1) - 1;
      p1 = p1 < p0 ? p0 : (p1 > clen - 1 ? clen - 1 : p1);
      reach = (int)(p1 / p.tile_len - p0 / p.tile_len);
    }
  }
  // Records are in (tile, class) order, so a wave mostly sees runs of one slot: only the first lane of a run
  // publishes the low bound and only the last one the high bound.  The same holds for the reads that reach into
  // the next tile (class 2 of one tile is one run): their bounds on that tile's incoming slot are published by the
  // run's first and last lane; only tiles further away (reach >= 2) are updated by every read.
  const int prev = __shfl_up(slot, 1);
  const int next = __shfl_down(slot, 1);
  const int reaching = (valid && reach >= 1) ? t0 : -1;     // run key of the straddlers of tile t0
  const int rprev = __shfl_up(reaching, 1);
  const int rnext = __shfl_down(reaching, 1);
  if (valid) {
    const uint32_t inv = (uint32_t)(p.n_reads - i);
    if (lane == 0 || prev != slot) atomicMax(&p.rbinv[slot], inv);
    if (lane == 63 || next != slot) atomicMax(&p.rend[slot], (uint32_t)(i + 1));
    if (reach >= 1) {
      if (lane == 0 || rprev != reaching) atomicMax(&p.rbinv[3 * (t0 + 1) + 2], inv);
      if (lane == 63 || rnext != reaching) atomicMax(&p.rend[3 * (t0 + 1) + 2], (uint32_t)(i + 1));
    }
    for (int k = 2; k <= reach; ++k) {   // incoming slot of every further tile touched
      atomicMax(&p.rbinv[3 * (t0 + k) + 2], inv);
      atomicMax(&p.rend[3 * (t0 + k) + 2], (uint32_t)(i + 1));
    }
  }
}

}  // namespace

hipError_t launch_index_reads(const IndexParams& p, hipStream_t stream) {
  // always launched (even with no reads): block 0 resets the counters and the error word
  const int grid = p.n_reads > 0 ? (p.n_reads + kIndexBlock - 1) / kIndexBlock : 1;
  hipLaunchKernelGGL(index_reads_kernel, dim3(grid), dim3(kIndexBlock), 0, stream, p);
  return hipGetLastError();
}

}  // namespace midas
