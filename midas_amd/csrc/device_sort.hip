// The two generic device primitives this library needs, written for it (gfx950; no hipCUB / rocPRIM on the product path):
//   * an exclusive scan of 32-bit counters (three phases: sums of 4096, their scan by one workgroup, the scan applied);
//   * a STABLE least-significant-digit radix sort of (32-bit key, value) pairs, eight bits of the key a pass, for keys whose
//     range is known (a gene index, a tile bin): as many passes as the range has bytes.  Input order survives inside a key --
//     what the callers are after (genes_count.hip: BAM order inside a gene; pack_reads.hip: read order inside a tile bin).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "kernels.h"

namespace midas {
namespace {

constexpr int kScanThreads = 1024, kScanBlock = 4 * kScanThreads;      // counters per workgroup

// inclusive scan of one value per thread over a 1024-thread workgroup; returns the exclusive prefix of the thread and, in
// *total, the workgroup's sum
__device__ __forceinline__ uint32_t block_exclusive(uint32_t mine, uint32_t* wsum /* [16] */, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const uint32_t x = wsum[w];
    before += w < wave ? x : 0u;
    all += x;
  }
  __syncthreads();       // (wsum may be written again by the caller's next round)
  *total = all;
  return before + inc - mine;
}

__global__ __launch_bounds__(kScanThreads) void scan_sums_kernel(const uint32_t* v, long long n, uint32_t* sums) {
  __shared__ uint32_t wsum[16];
  const long long i = (long long)blockIdx.x * kScanBlock + 4ll * threadIdx.x;
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) mine += i + k < n ? v[i + k] : 0u;
  uint32_t total;
  (void)block_exclusive(mine, wsum, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one workgroup: exclusive scan of m counters in place, 4096 at a time with a running carry
__global__ __launch_bounds__(kScanThreads) void scan_one_group_kernel(uint32_t* v, long long m) {
  __shared__ uint32_t wsum[16];
  uint32_t carry = 0;
  for (long long t0 = 0; t0 < m; t0 += kScanBlock) {
    const long long i = t0 + 4ll * threadIdx.x;
    uint32_t a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = i + k < m ? v[i + k] : 0u;
    uint32_t total;
    uint32_t run = carry + block_exclusive(a[0] + a[1] + a[2] + a[3], wsum, &total);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i + k < m) v[i + k] = run;
      run += a[k];
    }
    carry += total;
  }
}

__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(const uint32_t* in, uint32_t* out, long long n, const uint32_t* sums) {
  __shared__ uint32_t wsum[16];
  const long long i = (long long)blockIdx.x * kScanBlock + 4ll * threadIdx.x;
  uint32_t a[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) a[k] = i + k < n ? in[i + k] : 0u;
  uint32_t total;
  uint32_t run = sums[blockIdx.x] + block_exclusive(a[0] + a[1] + a[2] + a[3], wsum, &total);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i + k < n) out[i + k] = run;
    run += a[k];
  }
}

// ---- the sort --------------------------------------------------------------------------------------------------------------
// A workgroup owns kSortBlock consecutive pairs, each of its four waves a quarter of them, taken 64 at a time: the order of
// equal digits is (workgroup, wave, round, lane) = the input order.
//   hist     per workgroup the number of keys of every digit            -> hist[digit][workgroup]
//   scan     exclusive scan over hist in that (digit-major) order       -> where a workgroup's keys of a digit go
//   scatter  a wave finds, per round, the lanes that share a lane's digit (eight ballots), ranks the lane among them and
//            moves its pair to the digit's cursor of the wave (LDS), which the lowest of those lanes then advances
constexpr int kSortThreads = 256, kSortWaves = kSortThreads / 64, kSortRounds = 16;
constexpr int kSortBlock = kSortThreads * kSortRounds;      // 4096 pairs

__global__ __launch_bounds__(kSortThreads) void sort_hist_kernel(const uint32_t* key, long long n, int shift, uint32_t* hist, uint32_t n_blocks) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const long long base = (long long)blockIdx.x * kSortBlock;
#pragma unroll 4
  for (int r = 0; r < kSortRounds; ++r) {
    const long long i = base + (long long)r * kSortThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[(key[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

template <class V>
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(const uint32_t* key, const V* val, long long n, int shift, const uint32_t* hist,
                                                                     uint32_t n_blocks, uint32_t* key_out, V* val_out) {
  __shared__ uint32_t cursor[kSortWaves][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int d = threadIdx.x; d < kSortWaves * 256; d += kSortThreads) (&cursor[0][0])[d] = 0u;
  __syncthreads();
  // the wave's own counts per digit ...
  const long long wbase = (long long)blockIdx.x * kSortBlock + (long long)wave * (kSortBlock / kSortWaves);
  for (int r = 0; r < kSortRounds; ++r) {
    const long long i = wbase + (long long)r * 64 + lane;
    if (i < n) atomicAdd(&cursor[wave][(key[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  // ... become where its keys of a digit start: the workgroup's place for the digit + the waves in front
  {
    const int d = threadIdx.x;       // 256 threads, 256 digits
    uint32_t at = hist[(size_t)d * n_blocks + blockIdx.x];
    for (int w = 0; w < kSortWaves; ++w) {
      const uint32_t c = cursor[w][d];
      cursor[w][d] = at;
      at += c;
    }
  }
  __syncthreads();
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < kSortRounds; ++r) {
    const long long i = wbase + (long long)r * 64 + lane;
    const bool live = i < n;
    const uint32_t k = live ? key[i] : 0u;
    const uint32_t d = (k >> shift) & 255u;
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long has = __ballot(live && ((d >> b) & 1u));
      peers &= ((d >> b) & 1u) ? has : ~has;
    }
    if (live) {
      const uint32_t at = cursor[wave][d] + (uint32_t)__popcll(peers & below);
      key_out[at] = k;
      val_out[at] = val[i];
    }
    // (LDS operations of one wave are carried out in order: every lane has read the cursor before its lowest peer moves it,
    // and the next round reads what this one wrote)
    if (live && (peers & below) == 0ull) cursor[wave][d] += (uint32_t)__popcll(peers);
  }
}

template <class V>
hipError_t sort_pairs(uint32_t* key_a, V* val_a, uint32_t* key_b, V* val_b, long long n, int bits, uint32_t* scratch, hipStream_t s,
                      uint32_t** key_sorted, V** val_sorted) {
  uint32_t* kin = key_a; V* vin = val_a; uint32_t* kout = key_b; V* vout = val_b;
  const uint32_t n_blocks = (uint32_t)((n + kSortBlock - 1) / kSortBlock);
  uint32_t* hist = scratch;
  uint32_t* sums = scratch + 256ull * n_blocks;
  for (int shift = 0; shift < bits && n > 0; shift += 8) {
    hipLaunchKernelGGL(sort_hist_kernel, dim3(n_blocks), dim3(kSortThreads), 0, s, kin, n, shift, hist, n_blocks);
    const hipError_t es = launch_scan_u32(hist, hist, 256ll * n_blocks, sums, s);
    if (es != hipSuccess) return es;
    hipLaunchKernelGGL(sort_scatter_kernel<V>, dim3(n_blocks), dim3(kSortThreads), 0, s, kin, vin, n, shift, hist, n_blocks, kout, vout);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    std::swap(kin, kout);
    std::swap(vin, vout);
  }
  *key_sorted = kin;
  *val_sorted = vin;
  return hipSuccess;
}

}  // namespace

size_t scan_scratch_words(long long n) { return (size_t)((n + kScanBlock - 1) / kScanBlock) + 1; }

hipError_t launch_scan_u32(const uint32_t* in, uint32_t* out, long long n, uint32_t* sums, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const long long nb = (n + kScanBlock - 1) / kScanBlock;
  if (nb == 1 && in == out) {
    hipLaunchKernelGGL(scan_one_group_kernel, dim3(1), dim3(kScanThreads), 0, s, out, n);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(scan_sums_kernel, dim3((unsigned)nb), dim3(kScanThreads), 0, s, in, n, sums);
  hipLaunchKernelGGL(scan_one_group_kernel, dim3(1), dim3(kScanThreads), 0, s, sums, nb);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(kScanThreads), 0, s, in, out, n, sums);
  return hipGetLastError();
}

size_t sort_scratch_words(long long n) {
  const size_t nb = (size_t)((n > 0 ? n : 1) + kSortBlock - 1) / kSortBlock;
  return 256 * nb + scan_scratch_words(256ll * (long long)nb);
}

hipError_t launch_sort_pairs_u32(uint32_t* key_a, uint32_t* val_a, uint32_t* key_b, uint32_t* val_b, long long n, int bits, uint32_t* scratch,
                                 hipStream_t s, uint32_t** key_sorted, uint32_t** val_sorted) {
  return sort_pairs<uint32_t>(key_a, val_a, key_b, val_b, n, bits, scratch, s, key_sorted, val_sorted);
}

hipError_t launch_sort_pairs_f64(uint32_t* key_a, double* val_a, uint32_t* key_b, double* val_b, long long n, int bits, uint32_t* scratch,
                                 hipStream_t s, uint32_t** key_sorted, double** val_sorted) {
  return sort_pairs<double>(key_a, val_a, key_b, val_b, n, bits, scratch, s, key_sorted, val_sorted);
}

}  // namespace midas
