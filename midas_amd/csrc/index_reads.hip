// gfx950 index kernel of the MIDAS SNP pileup: per-tile read ranges, the device-side `samtools index`
// (reference: midas/run/snps.py:130-137 index_bam; pysam's fetch(contig, 0, length) then walks the index).
#include "device_common.h"

namespace midas {

using namespace dev;

namespace {

// ------------------------------------------------------------------------------------------------
// Index kernel: four consecutive records per thread, every pass.  Input is one 32-bit key per record, written by the packer
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
// Four consecutive records per thread (one 16-byte load of keys): with one record per thread the kernel was bound by the
// number of waves it takes to touch 11.7 M records (34 us on configs[2] for 47 MB of keys).
constexpr int kIndexPer = 4;

__global__ __launch_bounds__(kIndexBlock) void index_reads_kernel(IndexParams p) {
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < p.n_stat_words; i += kIndexBlock) p.stats[i] = 0ull;
    if (threadIdx.x == 0) *p.err = kNoError;
  }
  for (int i = blockIdx.x * kIndexBlock + threadIdx.x; i < 3 * p.n_tiles; i += gridDim.x * kIndexBlock) {
    p.rbinv_next[i] = 0u;
    p.rend_next[i] = 0u;
  }
  const int first = (blockIdx.x * kIndexBlock + threadIdx.x) * kIndexPer;   // this thread's records: first .. first + 3
  const int lane = threadIdx.x & 63;
  uint32_t key[kIndexPer];
  if (first + kIndexPer <= p.n_reads) {
    const uint4 k4 = *reinterpret_cast<const uint4*>(p.key + first);          // (hipMalloc'ed: 16-byte aligned)
    key[0] = k4.x; key[1] = k4.y; key[2] = k4.z; key[3] = k4.w;
  } else {
#pragma unroll
    for (int j = 0; j < kIndexPer; ++j) key[j] = first + j < p.n_reads ? p.key[first + j] : 0u;
  }
  int slot[kIndexPer], reaching[kIndexPer], t0[kIndexPer], reach[kIndexPer];
#pragma unroll
  for (int j = 0; j < kIndexPer; ++j) {
    const int i = first + j;
    const bool valid = i < p.n_reads;
    const int cls = (int)(key[j] & 3u);
    reach[j] = (int)((key[j] >> 2) & 31u);
    t0[j] = (int)(key[j] >> 7);
    slot[j] = valid ? 3 * t0[j] + (cls == 0 ? 0 : 1) : -1;
    if (valid && reach[j] == 31 && !(rec_flags(reinterpret_cast<const uint4*>(p.rec)[i]) & kRecSimple)) {
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
      const Tile tl = p.tiles[t0[j]];
      const long long clen = tl.contig_len;
      long long p0 = rec_pos(r);
      p0 = p0 < 0 ? 0 : (p0 > clen - 1 ? clen - 1 : p0);
      long long p1 = (long long)rec_pos(r) + (reflen > 0 ? reflen : 1) - 1;
      p1 = p1 < p0 ? p0 : (p1 > clen - 1 ? clen - 1 : p1);
      reach[j] = (int)(p1 / p.tile_len - p0 / p.tile_len);
    }
    reaching[j] = (valid && reach[j] >= 1) ? t0[j] : -1;     // run key of the straddlers of tile t0
  }
  // Records are in (tile, class) order, so consecutive records mostly share a slot: only the first record of a run
  // publishes the low bound and only the last one the high bound.  The same holds for the reads that reach into the
  // next tile (class 2 of one tile is one run): their bounds on that tile's incoming slot are published by the run's
  // first and last record; only tiles further away (reach >= 2) are updated by every read.  A run is cut at wave borders
  // (the neighbours' values come by shuffle): one more atomic there, never a wrong bound.
  const int slot_before = __shfl_up(slot[kIndexPer - 1], 1), slot_after = __shfl_down(slot[0], 1);
  const int reach_before = __shfl_up(reaching[kIndexPer - 1], 1), reach_after = __shfl_down(reaching[0], 1);
#pragma unroll
  for (int j = 0; j < kIndexPer; ++j) {
    if (slot[j] < 0) continue;
    const int i = first + j;
    const uint32_t inv = (uint32_t)(p.n_reads - i);
    const bool run_starts = j == 0 ? (lane == 0 || slot_before != slot[0]) : slot[j - 1] != slot[j];
    const bool run_ends = j == kIndexPer - 1 ? (lane == 63 || slot_after != slot[j]) : slot[j + 1] != slot[j];
    if (run_starts) atomicMax(&p.rbinv[slot[j]], inv);
    if (run_ends) atomicMax(&p.rend[slot[j]], (uint32_t)(i + 1));
    if (reach[j] >= 1) {
      const bool r_starts = j == 0 ? (lane == 0 || reach_before != reaching[0]) : reaching[j - 1] != reaching[j];
      const bool r_ends = j == kIndexPer - 1 ? (lane == 63 || reach_after != reaching[j]) : reaching[j + 1] != reaching[j];
      if (r_starts) atomicMax(&p.rbinv[3 * (t0[j] + 1) + 2], inv);
      if (r_ends) atomicMax(&p.rend[3 * (t0[j] + 1) + 2], (uint32_t)(i + 1));
      for (int k = 2; k <= reach[j]; ++k) {   // incoming slot of every further tile touched
        atomicMax(&p.rbinv[3 * (t0[j] + k) + 2], inv);
        atomicMax(&p.rend[3 * (t0[j] + k) + 2], (uint32_t)(i + 1));
      }
    }
  }
}

}  // namespace

hipError_t launch_index_reads(const IndexParams& p, hipStream_t stream) {
  // always launched (even with no reads): block 0 resets the counters and the error word
  const int per_block = kIndexBlock * kIndexPer;
  const int grid = p.n_reads > 0 ? (p.n_reads + per_block - 1) / per_block : 1;
  hipLaunchKernelGGL(index_reads_kernel, dim3(grid), dim3(kIndexBlock), 0, stream, p);
  return hipGetLastError();
}

}  // namespace midas
