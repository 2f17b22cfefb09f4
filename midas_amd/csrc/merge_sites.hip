// `merge_midas.py snps` on MI355X: the per-site cross-sample arithmetic of GenomicSite
// (/root/reference/midas/merge/snps.py:13-114): pooled counts, major/minor allele call, snp_type,
// per-sample depth / minor-allele count, prevalence and the site flag.  One thread per site, streaming over the
// samples' count tables ([sample][site][A,C,G,T] u32, i.e. exactly what the pileup stage emits): HBM-bound,
// 16 B read + 8 B written per (site, sample), 40 B of per-site results.  Annotation and text emission stay on the host.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../../include/midas_snps.h"
#include "ctx_internal.h"

namespace {

struct MergeKParams {
  const uint32_t* counts;     // [n_samples][n_sites][4]
  const double* mean_depth;   // [n_samples]
  uint32_t* calls;            // [n_sites] bytes {major, minor, snp_type, flag}: 0..3 = A,C,G,T, 255 = None;
                              // snp_type 0 None, 1 mono, 2 bi, 3 tri, 4 quad; flag 0 keep, 1 min_prev, 2 snp_type
  uint32_t* count_samples;
  unsigned long long* pooled; // [n_sites][4]
  uint32_t* depth;            // [n_samples][n_sites]  major + minor count
  uint32_t* minor_count;      // [n_samples][n_sites]
  unsigned long long* err;    // lowest site where the reference would raise ZeroDivisionError, else ~0
  long long n_sites;
  int n_samples;
  int site_depth;
  int snp_types;
  double allele_freq, site_ratio, site_prev;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifndef MIDAS_MERGE_SPLIT_FROM
#define MIDAS_MERGE_SPLIT_FROM 28
#endif
constexpr int kSplitFrom = MIDAS_MERGE_SPLIT_FROM;   // samples from which several waves share a site group

// ---- per-site arithmetic shared by the two kernel forms ---------------------------------------------------------
// call_alleles (:49-76) from the pooled counts
struct SiteCall { int major, minor, snp; };
__device__ __forceinline__ SiteCall call_site(const unsigned long long (&pc)[4], double allele_freq) {
  SiteCall r{255, 255, 0};
  const unsigned long long pooled_depth = pc[0] + pc[1] + pc[2] + pc[3];
  if (pooled_depth > 0) {
    // stable descending sort of the four alleles by frequency; count/depth is strictly monotone in count, so
    // sorting by count is the same order, and ties keep A,C,G,T order (Python's sorted is stable)
    int ord[4] = {0, 1, 2, 3};
#pragma unroll
    for (int a = 1; a < 4; ++a) {
#pragma unroll
      for (int b = a; b > 0; --b) {
        if (pc[ord[b]] > pc[ord[b - 1]]) { const int t = ord[b]; ord[b] = ord[b - 1]; ord[b - 1] = t; }
      }
    }
    const double d = (double)pooled_depth;
    const double f0 = (double)pc[ord[0]] / d, f1 = (double)pc[ord[1]] / d, f2 = (double)pc[ord[2]] / d,
                 f3 = (double)pc[ord[3]] / d;
    if (f0 > 0) r.major = ord[0];
    if (f1 > 0) r.minor = ord[1];
    if (f3 >= allele_freq) r.snp = 4;
    else if (f2 >= allele_freq) r.snp = 3;
    else if (f1 >= allele_freq) r.snp = 2;
    else if (f0 >= allele_freq) r.snp = 1;
  }
  return r;
}

// compute_per_sample_mafs (:78-91) + compute_prevalence (:93-104) for one sample's row of the site
struct SampleAcc { uint32_t pass = 0; bool zero_div = false; };
__device__ __forceinline__ void sample_row(const MergeKParams& p, long long i, int s, const u32x4 c, const SiteCall& sc,
                                           SampleAcc& acc) {
  // counts of the major / minor allele without indexing a register array by a runtime value
  const uint32_t cmaj = sc.major == 0 ? c.x : sc.major == 1 ? c.y : sc.major == 2 ? c.z : sc.major == 3 ? c.w : 0u;
  const uint32_t mc = sc.minor == 0 ? c.x : sc.minor == 1 ? c.y : sc.minor == 2 ? c.z : sc.minor == 3 ? c.w : 0u;
  const uint32_t sd = cmaj + mc;                             // the table reader bounds counts to 31 bits
  __builtin_nontemporal_store(sd, &p.depth[(size_t)s * p.n_sites + i]);
  __builtin_nontemporal_store(mc, &p.minor_count[(size_t)s * p.n_sites + i]);
  if ((long long)sd < (long long)p.site_depth) return;
  const double md = p.mean_depth[s];
  if (md == 0.0) { acc.zero_div = true; return; }            // Python: ZeroDivisionError
  if ((double)sd / md > p.site_ratio) return;
  ++acc.pass;
}

// flag (:106-114) and the per-site outputs
__device__ __forceinline__ void finish_site(const MergeKParams& p, long long i, const unsigned long long (&pc)[4],
                                            const SiteCall& sc, uint32_t pass, bool zero_div) {
  const double prevalence = (double)pass / (double)p.n_samples;
  int flag = 0;
  if (prevalence < p.site_prev) flag = 1;
  else if (!(p.snp_types & 1) && !(sc.snp > 0 && (p.snp_types & (1 << sc.snp)))) flag = 2;
  p.calls[i] = (uint32_t)sc.major | ((uint32_t)sc.minor << 8) | ((uint32_t)sc.snp << 16) | ((uint32_t)flag << 24);
  p.count_samples[i] = pass;
  reinterpret_cast<ulonglong2*>(p.pooled)[2 * i] = make_ulonglong2(pc[0], pc[1]);
  reinterpret_cast<ulonglong2*>(p.pooled)[2 * i + 1] = make_ulonglong2(pc[2], pc[3]);
  if (zero_div) atomicMin(p.err, (unsigned long long)i);
}

// Thread-per-site form: two passes over the site's count rows -- pooling, then per-sample depth.  The second pass hits
// L2 / Infinity Cache while the rows of all resident threads fit there (up to a couple of dozen samples).
__global__ __launch_bounds__(256) void merge_sites_kernel(MergeKParams p) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.n_sites; i += stride) {
    const u32x4* cnt = reinterpret_cast<const u32x4*>(p.counts) + i;
    unsigned long long pc[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll 8                                                 // eight independent row loads in flight per thread
    for (int s = 0; s < p.n_samples; ++s) {                      // compute_pooled_counts (:38-43)
      const u32x4 c = cnt[(size_t)s * p.n_sites];
      pc[0] += c.x; pc[1] += c.y; pc[2] += c.z; pc[3] += c.w;
    }
    const SiteCall sc = call_site(pc, p.allele_freq);
    SampleAcc acc;
#pragma unroll 8
    for (int s = 0; s < p.n_samples; ++s) sample_row(p, i, s, cnt[(size_t)s * p.n_sites], sc, acc);
    finish_site(p, i, pc, sc, acc.pass, acc.zero_div);
  }
}

// Several-threads-per-site form for many samples (BASELINE config 5 merges 50).  With dozens of samples the rows that all
// resident threads hold between their two passes (2 048 threads x 256 CUs x 16 B x samples) outgrow the 256 MB Infinity
// Cache and the second pass comes from HBM again (PMC: 4.8 GB read for 2.4 GB of tables at 50 samples).  Here a
// workgroup takes 64 consecutive sites and its G waves split the samples, so the chip holds G times fewer sites at a
// time and the second pass finds its rows in the Infinity Cache.  Wave w handles samples w, w + G, ...: 64 lanes read
// 64 consecutive rows (1 KiB, coalesced).  The waves' partial sums meet in LDS (a few hundred bytes, three barriers per
// 64 sites; the accumulators alternate between two sets so that re-zeroing them needs no barrier of its own).
template <int G>
__global__ __launch_bounds__(64 * G) void merge_sites_split_kernel(MergeKParams p) {
  constexpr int T = 64;
  __shared__ unsigned long long s_pool[2][4][T];
  __shared__ uint32_t s_pass[2][T], s_zero[2][T], s_call[T];
  const int tid = threadIdx.x, site_l = tid % T, g = tid / T;
  const long long n_tiles = (p.n_sites + T - 1) / T;
  if (tid < T) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      s_pool[b][0][tid] = s_pool[b][1][tid] = s_pool[b][2][tid] = s_pool[b][3][tid] = 0ull;
      s_pass[b][tid] = 0u;
      s_zero[b][tid] = 0u;
    }
  }
  __syncthreads();
  int par = 0;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, par ^= 1) {
    const long long i = tile * T + site_l;
    const bool valid = i < p.n_sites;
    const u32x4* cnt = reinterpret_cast<const u32x4*>(p.counts) + (valid ? i : 0);
    if (valid) {                                                 // compute_pooled_counts (:38-43), this wave's samples
      unsigned long long pc[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll 8
      for (int s = g; s < p.n_samples; s += G) {
        const u32x4 c = cnt[(size_t)s * p.n_sites];
        pc[0] += c.x; pc[1] += c.y; pc[2] += c.z; pc[3] += c.w;
      }
      atomicAdd(&s_pool[par][0][site_l], pc[0]); atomicAdd(&s_pool[par][1][site_l], pc[1]);
      atomicAdd(&s_pool[par][2][site_l], pc[2]); atomicAdd(&s_pool[par][3][site_l], pc[3]);
    }
    __syncthreads();
    if (g == 0) {                                                // one wave calls the alleles of the 64 sites
      const unsigned long long tot[4] = {s_pool[par][0][site_l], s_pool[par][1][site_l], s_pool[par][2][site_l],
                                         s_pool[par][3][site_l]};
      const SiteCall sc = call_site(tot, p.allele_freq);
      s_call[site_l] = (uint32_t)sc.major | ((uint32_t)sc.minor << 8) | ((uint32_t)sc.snp << 16);
    }
    __syncthreads();
    const uint32_t cw = s_call[site_l];
    const SiteCall sc{(int)(cw & 255u), (int)((cw >> 8) & 255u), (int)(cw >> 16)};
    if (valid) {                                                 // second pass over this wave's samples
      SampleAcc acc;
#pragma unroll 8
      for (int s = g; s < p.n_samples; s += G) sample_row(p, i, s, cnt[(size_t)s * p.n_sites], sc, acc);
      if (acc.pass) atomicAdd(&s_pass[par][site_l], acc.pass);
      if (acc.zero_div) s_zero[par][site_l] = 1u;
    }
    __syncthreads();
    if (g == 0) {
      if (valid) {
        const unsigned long long tot[4] = {s_pool[par][0][site_l], s_pool[par][1][site_l], s_pool[par][2][site_l],
                                           s_pool[par][3][site_l]};
        finish_site(p, i, tot, sc, s_pass[par][site_l], s_zero[par][site_l] != 0u);
      }
      // this set is used again two tiles on; the barriers of the next tile order the re-zeroing against that
      s_pool[par][0][site_l] = s_pool[par][1][site_l] = s_pool[par][2][site_l] = s_pool[par][3][site_l] = 0ull;
      s_pass[par][site_l] = 0u;
      s_zero[par][site_l] = 0u;
    }
  }
}

int32_t mfail(midas_snps_ctx* ctx, int32_t st, const char* what, hipError_t e) {
  char buf[384];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  (void)hipGetLastError();
  ctx->err = buf;
  return st;
}

}  // namespace

#define M_TRY(call)                                                                                             \
  do {                                                                                                          \
    hipError_t e__ = (call);                                                                                    \
    if (e__ != hipSuccess) {                                                                                    \
      for (void* q__ : dev) (void)hipFree(q__);                                                                 \
      return mfail(ctx, e__ == hipErrorOutOfMemory ? MIDAS_SNPS_ERR_OUT_OF_MEMORY : MIDAS_SNPS_ERR_HIP, #call, e__); \
    }                                                                                                           \
  } while (0)

extern "C" int32_t midas_merge_sites(midas_snps_ctx* ctx, const midas_merge_params* prm, int32_t n_samples,
                                     int64_t n_sites, const uint32_t* const* sample_counts, const double* mean_depth,
                                     uint8_t* out_calls, uint32_t* out_count_samples, uint64_t* out_pooled, uint32_t* out_depth,
                                     uint32_t* out_minor_count, float* out_kernel_ms) {
  if (!ctx || !prm || n_samples <= 0 || n_sites < 0 || !sample_counts || !mean_depth || !out_calls ||
      !out_count_samples || !out_pooled || !out_depth || !out_minor_count)
    return MIDAS_SNPS_ERR_INVALID_ARG;
  ctx->err.clear();
  ctx->err_read = -1;
  if (out_kernel_ms) *out_kernel_ms = 0.f;
  std::vector<void*> dev;
  M_TRY(hipSetDevice(ctx->device));
  // sites are processed in chunks so that any number of samples fits: <= ~4 GiB of count tables at a time
  long long chunk = (long long)((4ull << 30) / (16ull * (unsigned long long)n_samples));
  if (chunk < 1024) chunk = 1024;
  if (chunk > n_sites) chunk = n_sites > 0 ? n_sites : 1;
  uint32_t* d_counts = nullptr; double* d_md = nullptr; uint32_t* d_b = nullptr; uint32_t* d_cs = nullptr;
  unsigned long long* d_pool = nullptr; uint32_t* d_depth = nullptr; uint32_t* d_mc = nullptr; unsigned long long* d_err = nullptr;
  M_TRY(hipMalloc(&d_counts, (size_t)chunk * n_samples * 16)); dev.push_back(d_counts);
  M_TRY(hipMalloc(&d_md, (size_t)n_samples * 8)); dev.push_back(d_md);
  M_TRY(hipMalloc(&d_b, (size_t)chunk * 4)); dev.push_back(d_b);
  M_TRY(hipMalloc(&d_cs, (size_t)chunk * 4)); dev.push_back(d_cs);
  M_TRY(hipMalloc(&d_pool, (size_t)chunk * 32)); dev.push_back(d_pool);
  M_TRY(hipMalloc(&d_depth, (size_t)chunk * n_samples * 4)); dev.push_back(d_depth);
  M_TRY(hipMalloc(&d_mc, (size_t)chunk * n_samples * 4)); dev.push_back(d_mc);
  M_TRY(hipMalloc(&d_err, 8)); dev.push_back(d_err);
  M_TRY(hipMemcpy(d_md, mean_depth, (size_t)n_samples * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  M_TRY(hipEventCreate(&e0));
  M_TRY(hipEventCreate(&e1));
  float total_ms = 0.f;
  int32_t status = MIDAS_SNPS_OK;
  for (long long lo = 0; lo < n_sites && status == MIDAS_SNPS_OK; lo += chunk) {
    const long long m = (n_sites - lo) < chunk ? (n_sites - lo) : chunk;
    for (int s = 0; s < n_samples; ++s)
      M_TRY(hipMemcpyAsync(d_counts + (size_t)s * m * 4, sample_counts[s] + (size_t)lo * 4, (size_t)m * 16,
                           hipMemcpyHostToDevice, ctx->stream));
    M_TRY(hipMemsetAsync(d_err, 0xFF, 8, ctx->stream));
    MergeKParams k;
    k.counts = d_counts; k.mean_depth = d_md;
    k.calls = d_b;
    k.count_samples = d_cs; k.pooled = d_pool; k.depth = d_depth; k.minor_count = d_mc; k.err = d_err;
    k.n_sites = m; k.n_samples = n_samples; k.site_depth = prm->site_depth; k.snp_types = prm->snp_types;
    k.allele_freq = prm->allele_freq; k.site_ratio = prm->site_ratio; k.site_prev = prm->site_prev;
    const int grid = (int)((m + 255) / 256 < 4096 ? (m + 255) / 256 : 4096);
    M_TRY(hipEventRecord(e0, ctx->stream));
    const dim3 g(grid > 0 ? grid : 1), b(256);
    if (n_samples >= kSplitFrom) {
      // waves per site group: enough that the rows resident between the passes stay below ~100 MB
      const long long n_tiles = (m + 63) / 64;
      const dim3 tg((unsigned)(n_tiles < 16384 ? (n_tiles > 0 ? n_tiles : 1) : 16384));
      if (n_samples < 2 * kSplitFrom) hipLaunchKernelGGL(merge_sites_split_kernel<4>, tg, dim3(256), 0, ctx->stream, k);
      else if (n_samples < 4 * kSplitFrom) hipLaunchKernelGGL(merge_sites_split_kernel<8>, tg, dim3(512), 0, ctx->stream, k);
      else hipLaunchKernelGGL(merge_sites_split_kernel<16>, tg, dim3(1024), 0, ctx->stream, k);
    } else {
      hipLaunchKernelGGL(merge_sites_kernel, g, b, 0, ctx->stream, k);
    }
    M_TRY(hipGetLastError());
    M_TRY(hipEventRecord(e1, ctx->stream));
    M_TRY(hipMemcpyAsync(out_calls + lo * 4, d_b, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
    M_TRY(hipMemcpyAsync(out_count_samples + lo, d_cs, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
    M_TRY(hipMemcpyAsync(out_pooled + lo * 4, d_pool, (size_t)m * 32, hipMemcpyDeviceToHost, ctx->stream));
    for (int s = 0; s < n_samples; ++s) {
      M_TRY(hipMemcpyAsync(out_depth + (size_t)s * n_sites + lo, d_depth + (size_t)s * m, (size_t)m * 4,
                           hipMemcpyDeviceToHost, ctx->stream));
      M_TRY(hipMemcpyAsync(out_minor_count + (size_t)s * n_sites + lo, d_mc + (size_t)s * m, (size_t)m * 4,
                           hipMemcpyDeviceToHost, ctx->stream));
    }
    unsigned long long err = ~0ull;
    M_TRY(hipMemcpyAsync(&err, d_err, 8, hipMemcpyDeviceToHost, ctx->stream));
    M_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    M_TRY(hipEventElapsedTime(&ms, e0, e1));
    total_ms += ms;
    if (err != ~0ull) {
      char buf[200];
      ctx->err_read = (int64_t)(lo + (long long)err);
      snprintf(buf, sizeof buf, "site %lld: a sample with mean_coverage 0 reached site_depth/mean_depth "
               "(reference: ZeroDivisionError in compute_prevalence)", (long long)ctx->err_read + 1);
      ctx->err = buf;
      status = MIDAS_MERGE_ERR_ZERO_MEAN_DEPTH;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (void* q : dev) (void)hipFree(q);
  if (out_kernel_ms) *out_kernel_ms = total_ms;
  return status;
}
