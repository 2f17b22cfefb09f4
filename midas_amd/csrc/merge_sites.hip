// `merge_midas.py snps` on MI355X: the per-site cross-sample arithmetic of GenomicSite
// (/root/reference/midas/merge/snps.py:13-114): pooled counts, major/minor allele call, snp_type,
// per-sample depth / minor-allele count, prevalence and the site flag.  One thread per site, streaming over the
// samples' count tables ([sample][site][A,C,G,T] u32, i.e. exactly what the pileup stage emits): HBM-bound,
// 16 B read + 8 B written per (site, sample), 40 B of per-site results.  Annotation and text emission stay on the host.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <utility>
#include <vector>

#include "../../include/midas_snps.h"
#include "ctx_internal.h"

namespace {

// A sample's part in compute_prevalence (:93-104), folded into integer limits on the host (sample_limits below): the
// site passes for the sample iff lo <= depth <= hi; it raises ZeroDivisionError iff zero_on and depth >= zero_from.
struct SampleLimit { uint32_t lo, hi, zero_from, zero_on; };

struct MergeKParams {
  const uint32_t* counts;     // [n_samples][n_sites][4]
  const SampleLimit* limit;   // [n_samples]
  uint32_t* calls;            // [n_sites] bytes {major, minor, snp_type, flag}: 0..3 = A,C,G,T, 255 = None;
                              // snp_type 0 None, 1 mono, 2 bi, 3 tri, 4 quad; flag 0 keep, 1 min_prev, 2 snp_type
  uint32_t* count_samples;
  unsigned long long* pooled; // [n_sites][4]
  uint32_t* depth;            // [n_samples][n_sites]  major + minor count
  uint32_t* minor_count;      // [n_samples][n_sites]
  unsigned long long* err;    // lowest site where the reference would raise ZeroDivisionError, else ~0
  uint32_t n_sites;           // sites of this chunk: < 2^26, so that every per-site byte offset fits 32 bits
  int n_samples;
  int snp_types;
  double allele_freq, site_prev;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifndef MIDAS_MERGE_SPLIT_FROM
#define MIDAS_MERGE_SPLIT_FROM 28
#endif
constexpr int kSplitFrom = MIDAS_MERGE_SPLIT_FROM;   // samples from which several waves share a site group
#ifndef MIDAS_MERGE_REGS_UP_TO
#define MIDAS_MERGE_REGS_UP_TO 64
#endif
constexpr int kRegsUpTo = MIDAS_MERGE_REGS_UP_TO;    // samples up to which a site's rows stay in registers (one pass)
constexpr long long kMaxChunkSites = 1ll << 26;

// Addressing: (uniform 64-bit base of the sample's row) + (32-bit byte offset of the site).  That is the form the
// global_load / global_store "saddr + voffset" encoding takes: no 64-bit address arithmetic in vector registers.
__device__ __forceinline__ u32x4 load_row(const MergeKParams& p, int s, uint32_t site) {
  const char* base = reinterpret_cast<const char*>(p.counts + (size_t)s * p.n_sites * 4);
  return *reinterpret_cast<const u32x4*>(base + (site * 16u));
}
__device__ __forceinline__ void store_u32(uint32_t* table, const MergeKParams& p, int s, uint32_t site, uint32_t v) {
  char* base = reinterpret_cast<char*>(table + (size_t)s * p.n_sites);
  __builtin_nontemporal_store(v, reinterpret_cast<uint32_t*>(base + (site * 4u)));
}

// ---- per-site arithmetic shared by the kernel forms --------------------------------------------------------------
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
struct SampleAcc { uint32_t pass = 0; uint32_t zero_div = 0; };
__device__ __forceinline__ void sample_row(const MergeKParams& p, uint32_t i, int s, const u32x4 c, const SiteCall& sc,
                                           SampleAcc& acc) {
  // counts of the major / minor allele without indexing a register array by a runtime value
  const uint32_t cmaj = sc.major == 0 ? c.x : sc.major == 1 ? c.y : sc.major == 2 ? c.z : sc.major == 3 ? c.w : 0u;
  const uint32_t mc = sc.minor == 0 ? c.x : sc.minor == 1 ? c.y : sc.minor == 2 ? c.z : sc.minor == 3 ? c.w : 0u;
  const uint32_t sd = cmaj + mc;                             // the table reader bounds counts to 31 bits
  store_u32(p.depth, p, s, i, sd);
  store_u32(p.minor_count, p, s, i, mc);
  // uniform, and read through the constant address space: scalar loads, not one vector load per thread
  typedef const __attribute__((address_space(4))) uint32_t* ConstWords;
  const ConstWords lw = (ConstWords)(uintptr_t)p.limit + 4 * s;
  const SampleLimit l{lw[0], lw[1], lw[2], lw[3]};
  acc.pass += (sd >= l.lo && sd <= l.hi) ? 1u : 0u;
  acc.zero_div |= (l.zero_on && sd >= l.zero_from) ? 1u : 0u;
}

// flag (:106-114) and the per-site outputs
__device__ __forceinline__ void finish_site(const MergeKParams& p, uint32_t i, const unsigned long long (&pc)[4],
                                            const SiteCall& sc, uint32_t pass, bool zero_div) {
  const double prevalence = (double)pass / (double)p.n_samples;
  int flag = 0;
  if (prevalence < p.site_prev) flag = 1;
  else if (!(p.snp_types & 1) && !(sc.snp > 0 && (p.snp_types & (1 << sc.snp)))) flag = 2;
  p.calls[i] = (uint32_t)sc.major | ((uint32_t)sc.minor << 8) | ((uint32_t)sc.snp << 16) | ((uint32_t)flag << 24);
  p.count_samples[i] = pass;
  char* pool = reinterpret_cast<char*>(p.pooled) + (i * 32u);
  reinterpret_cast<ulonglong2*>(pool)[0] = make_ulonglong2(pc[0], pc[1]);
  reinterpret_cast<ulonglong2*>(pool)[1] = make_ulonglong2(pc[2], pc[3]);
  if (zero_div) atomicMin(p.err, (unsigned long long)i);
}

// Thread-per-site form: two passes over the site's count rows -- pooling, then per-sample depth.  The second pass hits
// L2 / Infinity Cache while the rows of all resident threads fit there (up to a couple of dozen samples).
__global__ __launch_bounds__(256) void merge_sites_kernel(MergeKParams p) {
  const uint32_t stride = gridDim.x * 256u;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < p.n_sites; i += stride) {
    unsigned long long pc[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll 8                                                 // eight independent row loads in flight per thread
    for (int s = 0; s < p.n_samples; ++s) {                      // compute_pooled_counts (:38-43)
      const u32x4 c = load_row(p, s, i);
      pc[0] += c.x; pc[1] += c.y; pc[2] += c.z; pc[3] += c.w;
    }
    const SiteCall sc = call_site(pc, p.allele_freq);
    SampleAcc acc;
#pragma unroll 8
    for (int s = 0; s < p.n_samples; ++s) sample_row(p, i, s, load_row(p, s, i), sc, acc);
    finish_site(p, i, pc, sc, acc.pass, acc.zero_div != 0u);
  }
}

// One-pass form: the site's rows stay in registers between the pooling and the per-sample pass, so every count row is
// read from HBM exactly once (the two-pass form above re-reads them, and with a quarter of a million sites in flight
// the second pass mostly misses L2: PMC 1.77 x the algorithmic reads at 8 samples).  What makes it fit: rows are packed
// as they arrive -- up to 8 samples as they are (4 registers a row), above that two counts per register -- and a site
// whose counts do not fit the packing (any count >= 2^16) takes the two-pass route for itself.  Loads go out eight rows
// at a time.  The number of samples S is a template parameter: the rows are named registers, not an indexed array, and
// the loads of a batch are straight-line code (with a run-time bound every load sat in its own branch behind an
// s_waitcnt vmcnt(0): one row in flight per thread).
template <int S, bool PADDED>
__global__ __launch_bounds__(256, S <= 40 ? 4 : (S <= 60 ? 3 : 2))   // <= 128 / 170 / 256 registers
void merge_sites_regs_kernel(MergeKParams p) {
  // registers per row and what a count must stay below for its site to take the fast route: the counts fit the packing,
  // and S of them sum below 2^32, so the pooled counts are plain 32-bit adds (a 64-bit add is two instructions and a
  // register pair per allele)
  constexpr int BITS = S <= 8 ? 32 : (S <= 64 ? 16 : 8);
  constexpr int W = BITS == 32 ? 4 : (BITS == 16 ? 2 : 1);
  constexpr int kSmallBits = BITS == 32 ? 28 : BITS;
  constexpr int LOADS = 8;
  static_assert(S <= 128 && ((unsigned long long)S << kSmallBits) <= (1ull << 32), "32-bit pooled sums");
  // PADDED: S is the sample count rounded up to a multiple of eight (one instantiation serves eight counts).  The rows
  // past the last sample are read from sample 0's first site and multiplied away -- a load under `if (s < n_samples)`
  // would sit in its own branch behind an s_waitcnt vmcnt(0), one row in flight per thread.
  const int n_real = PADDED ? p.n_samples : S;
  const uint32_t stride = gridDim.x * 256u;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < p.n_sites; i += stride) {
    uint32_t sum[4] = {0u, 0u, 0u, 0u};
    uint32_t row[S][W];
    uint32_t big = 0u;
#pragma unroll
    for (int s0 = 0; s0 < S; s0 += LOADS) {
      constexpr int kMaxBatch = LOADS < S ? LOADS : S;
      u32x4 c[kMaxBatch];
#pragma unroll
      for (int k = 0; k < kMaxBatch; ++k) {
        if (s0 + k < S) {
          const bool real = !PADDED || s0 + k < n_real;
          c[k] = load_row(p, real ? s0 + k : 0, real ? i : 0u);
        }
      }
#pragma unroll
      for (int k = 0; k < kMaxBatch; ++k) {
        if (s0 + k < S) {
          const int s = s0 + k;
          if (PADDED) {
            const uint32_t keep = s < n_real ? 0xFFFFFFFFu : 0u;
            c[k].x &= keep; c[k].y &= keep; c[k].z &= keep; c[k].w &= keep;
          }
          sum[0] += c[k].x; sum[1] += c[k].y; sum[2] += c[k].z; sum[3] += c[k].w;  // compute_pooled_counts (:38-43)
          big |= c[k].x | c[k].y | c[k].z | c[k].w;
          if (BITS == 32) {
            row[s][0] = c[k].x; row[s][W > 1 ? 1 : 0] = c[k].y; row[s][W > 2 ? 2 : 0] = c[k].z; row[s][W > 3 ? 3 : 0] = c[k].w;
          } else if (BITS == 16) {
            row[s][0] = c[k].x | (c[k].y << 16);
            row[s][W > 1 ? 1 : 0] = c[k].z | (c[k].w << 16);
            // packed HERE: left alone, the compiler sinks the packing into the branch that uses it and keeps the raw
            // rows (twice the registers) alive until then
            asm volatile("" : "+v"(row[s][0]), "+v"(row[s][W > 1 ? 1 : 0]));
          } else {
            row[s][0] = c[k].x | (c[k].y << 8) | (c[k].z << 16) | (c[k].w << 24);
            asm volatile("" : "+v"(row[s][0]));
          }
        }
      }
      asm volatile("" : "+v"(big), "+v"(sum[0]), "+v"(sum[1]), "+v"(sum[2]), "+v"(sum[3]));
      // the next loads go out behind this batch's packing, not in front of it: hoisting them all (what the scheduler
      // does on its own) needs every raw row live at once
      __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long pc[4] = {sum[0], sum[1], sum[2], sum[3]};
    SampleAcc acc;
    if (big >> kSmallBits) {     // a large count: this site takes the two-pass route, with 64-bit sums
      pc[0] = pc[1] = pc[2] = pc[3] = 0ull;
#pragma unroll 1
      for (int s = 0; s < n_real; ++s) {
        const u32x4 c = load_row(p, s, i);
        pc[0] += c.x; pc[1] += c.y; pc[2] += c.z; pc[3] += c.w;
      }
      const SiteCall sc = call_site(pc, p.allele_freq);
#pragma unroll 1
      for (int s = 0; s < n_real; ++s) sample_row(p, i, s, load_row(p, s, i), sc, acc);
      finish_site(p, i, pc, sc, acc.pass, acc.zero_div != 0u);
      continue;
    }
    const SiteCall sc = call_site(pc, p.allele_freq);
#pragma unroll
    for (int s = 0; s < S; ++s) {
      if (!PADDED || s < n_real) {
        u32x4 c;
        if (BITS == 32) c = u32x4{row[s][0], row[s][W > 1 ? 1 : 0], row[s][W > 2 ? 2 : 0], row[s][W > 3 ? 3 : 0]};
        else if (BITS == 16) c = u32x4{row[s][0] & 0xFFFFu, row[s][0] >> 16, row[s][W > 1 ? 1 : 0] & 0xFFFFu, row[s][W > 1 ? 1 : 0] >> 16};
        else c = u32x4{row[s][0] & 0xFFu, (row[s][0] >> 8) & 0xFFu, (row[s][0] >> 16) & 0xFFu, row[s][0] >> 24};
        sample_row(p, i, s, c, sc, acc);
      }
    }
    finish_site(p, i, pc, sc, acc.pass, acc.zero_div != 0u);
  }
}

typedef void (*MergeKernel)(MergeKParams);
template <int... S>
constexpr MergeKernel regs_kernel_for(int n_samples, std::integer_sequence<int, S...>) {
  MergeKernel k = nullptr;
  (void)((n_samples == S + 1 ? (k = merge_sites_regs_kernel<S + 1, false>, true) : false) || ...);
  return k;
}
// 65..128 samples: one instantiation per eight counts, a byte per count
template <int... E>
constexpr MergeKernel bytes_kernel_for(int n_samples, std::integer_sequence<int, E...>) {
  MergeKernel k = nullptr;
  (void)(((n_samples + 7) / 8 == 9 + E ? (k = merge_sites_regs_kernel<8 * (9 + E), true>, true) : false) || ...);
  return k;
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
  const uint32_t n_tiles = (p.n_sites + T - 1) / T;
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
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, par ^= 1) {
    const uint32_t i = tile * T + site_l;
    const bool valid = i < p.n_sites;
    const uint32_t ia = valid ? i : 0u;
    if (valid) {                                                 // compute_pooled_counts (:38-43), this wave's samples
      unsigned long long pc[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll 8
      for (int s = g; s < p.n_samples; s += G) {
        const u32x4 c = load_row(p, s, ia);
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
      for (int s = g; s < p.n_samples; s += G) sample_row(p, i, s, load_row(p, s, ia), sc, acc);
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

// compute_prevalence (:93-104) keeps a sample's site when depth >= site_depth and depth / mean_depth <= site_ratio, and
// divides by a mean depth of zero when one gets that far.  IEEE division is monotone in its numerator, so for a given
// sample the depths that pass form one interval of integers: found here by bisection with the very expression the
// reference evaluates (double division, then the comparison), which keeps the kernel free of 64-bit divisions and
// the result identical for every depth, NaN and negative means included.
SampleLimit sample_limits(double mean_depth, double site_ratio, int site_depth) {
  const uint32_t kMax = 0xFFFFFFFFu;
  const uint32_t from = site_depth > 0 ? (uint32_t)site_depth : 0u;
  SampleLimit l{1u, 0u, 0u, 0u};                    // lo > hi: nothing passes
  if (mean_depth == 0.0) { l.zero_on = 1u; l.zero_from = from; return l; }
  auto keeps = [&](uint32_t depth) { return !((double)depth / mean_depth > site_ratio); };
  const bool k0 = keeps(0u), k1 = keeps(kMax);
  uint32_t lo = 0u, hi = kMax;
  if (!k0 && !k1) return l;
  if (k0 && !k1) {                                  // true on a prefix: the last depth that is kept
    uint32_t a = 0u, b = kMax;                      // keeps(a), !keeps(b)
    while (b - a > 1u) { const uint32_t m = a + (b - a) / 2u; (keeps(m) ? a : b) = m; }
    hi = a;
  } else if (!k0 && k1) {                           // true on a suffix (a negative mean): the first depth that is kept
    uint32_t a = 0u, b = kMax;                      // !keeps(a), keeps(b)
    while (b - a > 1u) { const uint32_t m = a + (b - a) / 2u; (keeps(m) ? b : a) = m; }
    lo = b;
  }
  if (lo < from) lo = from;
  l.lo = lo; l.hi = hi;
  return l;
}

int32_t mfail(midas_snps_ctx* ctx, int32_t st, const char* what, hipError_t e) {
  char buf[384];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  (void)hipGetLastError();
  ctx->set_error(buf);
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
  ctx->clear_error();
  ctx->err_read = -1;
  if (out_kernel_ms) *out_kernel_ms = 0.f;
  std::vector<void*> dev;
  M_TRY(hipSetDevice(ctx->device));
  // sites are processed in chunks so that any number of samples fits: <= ~4 GiB of count tables at a time
  long long chunk = (long long)((4ull << 30) / (16ull * (unsigned long long)n_samples));
  if (chunk < 1024) chunk = 1024;
  if (chunk > kMaxChunkSites) chunk = kMaxChunkSites;
  if (chunk > n_sites) chunk = n_sites > 0 ? n_sites : 1;
  // 65..128 samples: a byte per count holds a site's rows when the samples are shallow (a site with a count >= 256 takes
  // the long way round by itself, which only pays while such sites are rare: mean depths well below 256)
  bool shallow = true;
  for (int s = 0; s < n_samples; ++s) shallow = shallow && mean_depth[s] <= 64.0;
  std::vector<SampleLimit> limits((size_t)n_samples);
  for (int s = 0; s < n_samples; ++s) limits[(size_t)s] = sample_limits(mean_depth[s], prm->site_ratio, prm->site_depth);
  uint32_t* d_counts = nullptr; SampleLimit* d_md = nullptr; uint32_t* d_b = nullptr; uint32_t* d_cs = nullptr;
  unsigned long long* d_pool = nullptr; uint32_t* d_depth = nullptr; uint32_t* d_mc = nullptr; unsigned long long* d_err = nullptr;
  M_TRY(hipMalloc(&d_counts, (size_t)chunk * n_samples * 16)); dev.push_back(d_counts);
  M_TRY(hipMalloc(&d_md, (size_t)n_samples * sizeof(SampleLimit))); dev.push_back(d_md);
  M_TRY(hipMalloc(&d_b, (size_t)chunk * 4)); dev.push_back(d_b);
  M_TRY(hipMalloc(&d_cs, (size_t)chunk * 4)); dev.push_back(d_cs);
  M_TRY(hipMalloc(&d_pool, (size_t)chunk * 32)); dev.push_back(d_pool);
  M_TRY(hipMalloc(&d_depth, (size_t)chunk * n_samples * 4)); dev.push_back(d_depth);
  M_TRY(hipMalloc(&d_mc, (size_t)chunk * n_samples * 4)); dev.push_back(d_mc);
  M_TRY(hipMalloc(&d_err, 8)); dev.push_back(d_err);
  M_TRY(hipMemcpy(d_md, limits.data(), (size_t)n_samples * sizeof(SampleLimit), hipMemcpyHostToDevice));
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
    k.counts = d_counts; k.limit = d_md;
    k.calls = d_b;
    k.count_samples = d_cs; k.pooled = d_pool; k.depth = d_depth; k.minor_count = d_mc; k.err = d_err;
    k.n_sites = (uint32_t)m; k.n_samples = n_samples; k.snp_types = prm->snp_types;
    k.allele_freq = prm->allele_freq; k.site_prev = prm->site_prev;
    const int grid = (int)((m + 255) / 256 < 4096 ? (m + 255) / 256 : 4096);
    M_TRY(hipEventRecord(e0, ctx->stream));
    const dim3 g(grid > 0 ? grid : 1), b(256);
#ifndef MIDAS_MERGE_TWO_PASS
    if (n_samples <= kRegsUpTo) {     // rows in registers: every count row is read once
      hipLaunchKernelGGL(regs_kernel_for(n_samples, std::make_integer_sequence<int, kRegsUpTo>{}), g, b, 0, ctx->stream, k);
    } else if (n_samples <= 128 && kRegsUpTo >= 64 && shallow) {   // the same with a byte per count
      hipLaunchKernelGGL(bytes_kernel_for(n_samples, std::make_integer_sequence<int, 8>{}), g, b, 0, ctx->stream, k);
    } else {
#endif
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
#ifndef MIDAS_MERGE_TWO_PASS   // (developer variants: the round-1 kernels, for A/B runs)
    }
#endif
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
      ctx->set_error(buf);
      status = MIDAS_MERGE_ERR_ZERO_MEAN_DEPTH;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (void* q : dev) (void)hipFree(q);
  if (out_kernel_ms) *out_kernel_ms = total_ms;
  return status;
}
