// Measurement aids of the C-ABI (no reference counterpart): what THIS device streams right now, and known-byte-count kernels
// that calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE counters for the access widths the step's kernels use.
//   midas_snps_stream_rates        a read stream, a write stream and a copy, each by a kernel built to saturate the memory pipe:
//                                  persistent grids of 2 / 4 / 8 / 16 workgroups per CU, 16 bytes per lane, four independent
//                                  accesses in flight per lane, plain and non-temporal, buffers far beyond the 256 MiB Infinity
//                                  Cache: the best shape of each stream is its rate
//   midas_snps_calibration_pass    reads `bytes` ONCE with 4-, 8- and 16-byte-per-lane loads and writes `bytes` once with
//                                  16-byte stores, one kernel each (midas_calib_read4_kernel ...): under rocprofv3 --pmc the
//                                  counter value of each kernel against the bytes it is known to move gives the counter's
//                                  factor for that width (MI355X_MICROARCH.md: "calibrate on a known byte count in your own
//                                  access pattern")
#include "ctx_internal.h"
#include "../../include/midas_snps.h"

#include <stdint.h>

#include <algorithm>

namespace {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;

// ---- saturating streams ---------------------------------------------------------------------------------------------------
template <bool NT> __device__ __forceinline__ v4u ld(const v4u* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(v4u v, v4u* p) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT>
__global__ __launch_bounds__(kBlock) void midas_stream_read_kernel(const v4u* __restrict__ src, size_t n16, uint32_t* sink) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  v4u acc = {0u, 0u, 0u, 0u};
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const v4u a = ld<NT>(src + i), b = ld<NT>(src + i + stride), c = ld<NT>(src + i + 2 * stride), d = ld<NT>(src + i + 3 * stride);
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < n16; i += stride) acc ^= ld<NT>(src + i);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1u;       // (never true for the fill pattern: keeps the loads)
}
template <bool NT>
__global__ __launch_bounds__(kBlock) void midas_stream_write_kernel(v4u* __restrict__ dst, size_t n16, uint32_t seed) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  const v4u v = {seed, seed + 1u, seed + 2u, seed + 3u};
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride) st<NT>(v, dst + i);
}
template <bool NT>
__global__ __launch_bounds__(kBlock) void midas_stream_copy_kernel(v4u* __restrict__ dst, const v4u* __restrict__ src, size_t n16) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const v4u a = ld<NT>(src + i), b = ld<NT>(src + i + stride), c = ld<NT>(src + i + 2 * stride), d = ld<NT>(src + i + 3 * stride);
    st<NT>(a, dst + i);
    st<NT>(b, dst + i + stride);
    st<NT>(c, dst + i + 2 * stride);
    st<NT>(d, dst + i + 3 * stride);
  }
  for (; i < n16; i += stride) st<NT>(ld<NT>(src + i), dst + i);
}

// ---- counter calibration: every byte of the buffer moved exactly once -----------------------------------------------------
template <class T>
__device__ __forceinline__ void calib_read(const T* __restrict__ src, size_t n, uint32_t* sink) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  uint32_t acc = 0u;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const T v = src[i];
    if constexpr (sizeof(T) == 4) acc ^= v;
    else if constexpr (sizeof(T) == 8) acc ^= v.x ^ v.y;
    else acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = 1u;
}
__global__ __launch_bounds__(kBlock) void midas_calib_read4_kernel(const uint32_t* src, size_t n, uint32_t* sink) { calib_read(src, n, sink); }
__global__ __launch_bounds__(kBlock) void midas_calib_read8_kernel(const v2u* src, size_t n, uint32_t* sink) { calib_read(src, n, sink); }
__global__ __launch_bounds__(kBlock) void midas_calib_read16_kernel(const v4u* src, size_t n, uint32_t* sink) { calib_read(src, n, sink); }
__global__ __launch_bounds__(kBlock) void midas_calib_write16_kernel(v4u* dst, size_t n, uint32_t seed) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  const v4u v = {seed, seed, seed, seed};
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) dst[i] = v;
}

int32_t fail(midas_snps_ctx* ctx, int32_t st, const char* msg) {
  if (ctx) ctx->set_error(msg);
  return st;
}

}  // namespace

extern "C" {

int32_t midas_snps_stream_rates(midas_snps_ctx* ctx, int64_t bytes, int32_t reps, double out_gbps[3]) {
  if (!ctx || !out_gbps || bytes < (1 << 20) || reps < 1) return MIDAS_SNPS_ERR_INVALID_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, MIDAS_SNPS_ERR_HIP, "stream_rates: hipSetDevice");
  const size_t n16 = (size_t)bytes / 16;
  v4u *a = nullptr, *b = nullptr;
  uint32_t* sink = nullptr;
  if (hipMalloc(&a, n16 * 16) != hipSuccess || hipMalloc(&b, n16 * 16) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) {
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink); (void)hipGetLastError();
    return fail(ctx, MIDAS_SNPS_ERR_OUT_OF_MEMORY, "stream_rates: out of device memory");
  }
  hipStream_t s = ctx->stream;
  hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
  int32_t st = MIDAS_SNPS_OK;
  for (auto& x : e) if (hipEventCreate(&x) != hipSuccess) st = MIDAS_SNPS_ERR_HIP;
  if (st == MIDAS_SNPS_OK && hipMemsetAsync(a, 0x5A, n16 * 16, s) != hipSuccess) st = MIDAS_SNPS_ERR_HIP;
  // every stream in a few shapes (workgroups per CU, plain or non-temporal accesses): the best one is this box's rate
  double best[3] = {0.0, 0.0, 0.0};
  const double moved = (double)(n16 * 16) * reps;
  for (int per_cu : {2, 4, 8, 16}) {
    for (int nt = 0; nt < 2 && st == MIDAS_SNPS_OK; ++nt) {
      const int grid = ctx->prop.multiProcessorCount * per_cu;
      auto rd = [&]() { if (nt) hipLaunchKernelGGL(midas_stream_read_kernel<true>, dim3(grid), dim3(kBlock), 0, s, a, n16, sink); else hipLaunchKernelGGL(midas_stream_read_kernel<false>, dim3(grid), dim3(kBlock), 0, s, a, n16, sink); };
      auto wr = [&](uint32_t seed) { if (nt) hipLaunchKernelGGL(midas_stream_write_kernel<true>, dim3(grid), dim3(kBlock), 0, s, b, n16, seed); else hipLaunchKernelGGL(midas_stream_write_kernel<false>, dim3(grid), dim3(kBlock), 0, s, b, n16, seed); };
      auto cp = [&]() { if (nt) hipLaunchKernelGGL(midas_stream_copy_kernel<true>, dim3(grid), dim3(kBlock), 0, s, b, a, n16); else hipLaunchKernelGGL(midas_stream_copy_kernel<false>, dim3(grid), dim3(kBlock), 0, s, b, a, n16); };
      rd(); wr(1u);                                   // (warm-up)
      (void)hipEventRecord(e[0], s);
      for (int k = 0; k < reps; ++k) rd();
      (void)hipEventRecord(e[1], s);
      for (int k = 0; k < reps; ++k) wr(2u + (uint32_t)k);
      (void)hipEventRecord(e[2], s);
      for (int k = 0; k < reps; ++k) cp();
      (void)hipEventRecord(e[3], s);
      if (hipEventSynchronize(e[3]) != hipSuccess) st = MIDAS_SNPS_ERR_HIP;
      float ms[3] = {0.f, 0.f, 0.f};
      for (int k = 0; k < 3 && st == MIDAS_SNPS_OK; ++k)
        if (hipEventElapsedTime(&ms[k], e[k], e[k + 1]) != hipSuccess || ms[k] <= 0.f) st = MIDAS_SNPS_ERR_HIP;
      if (st == MIDAS_SNPS_OK) {
        best[0] = std::max(best[0], moved / ((double)ms[0] * 1e-3) / 1e9);
        best[1] = std::max(best[1], moved / ((double)ms[1] * 1e-3) / 1e9);
        best[2] = std::max(best[2], 2.0 * moved / ((double)ms[2] * 1e-3) / 1e9);
      }
    }
  }
  for (auto x : e) if (x) (void)hipEventDestroy(x);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
  if (st != MIDAS_SNPS_OK) { (void)hipGetLastError(); return fail(ctx, st, "stream_rates: HIP runtime error"); }
  out_gbps[0] = best[0];
  out_gbps[1] = best[1];
  out_gbps[2] = best[2];
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_calibration_pass(midas_snps_ctx* ctx, int64_t bytes) {
  if (!ctx || bytes < (1 << 20)) return MIDAS_SNPS_ERR_INVALID_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, MIDAS_SNPS_ERR_HIP, "calibration_pass: hipSetDevice");
  const size_t n16 = (size_t)bytes / 16;
  v4u* a = nullptr;
  uint32_t* sink = nullptr;
  if (hipMalloc(&a, n16 * 16) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) {
    (void)hipFree(a); (void)hipFree(sink); (void)hipGetLastError();
    return fail(ctx, MIDAS_SNPS_ERR_OUT_OF_MEMORY, "calibration_pass: out of device memory");
  }
  hipStream_t s = ctx->stream;
  const int grid = ctx->prop.multiProcessorCount * 4;
  int32_t st = MIDAS_SNPS_OK;
  if (hipMemsetAsync(a, 0x5A, n16 * 16, s) != hipSuccess) st = MIDAS_SNPS_ERR_HIP;
  if (st == MIDAS_SNPS_OK) {
    hipLaunchKernelGGL(midas_calib_read4_kernel, dim3(grid), dim3(kBlock), 0, s, reinterpret_cast<const uint32_t*>(a), n16 * 4, sink);
    hipLaunchKernelGGL(midas_calib_read8_kernel, dim3(grid), dim3(kBlock), 0, s, reinterpret_cast<const v2u*>(a), n16 * 2, sink);
    hipLaunchKernelGGL(midas_calib_read16_kernel, dim3(grid), dim3(kBlock), 0, s, a, n16, sink);
    hipLaunchKernelGGL(midas_calib_write16_kernel, dim3(grid), dim3(kBlock), 0, s, a, n16, 7u);
    if (hipStreamSynchronize(s) != hipSuccess) st = MIDAS_SNPS_ERR_HIP;
  }
  (void)hipFree(a); (void)hipFree(sink);
  if (st != MIDAS_SNPS_OK) { (void)hipGetLastError(); return fail(ctx, st, "calibration_pass: HIP runtime error"); }
  return MIDAS_SNPS_OK;
}

}  // extern "C"
