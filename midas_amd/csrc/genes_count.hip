// `run_midas.py genes` on MI355X: count_mapped_bp (/root/reference/midas/run/genes.py:165-189) -- for every gene of the
// pangenome, how many reads align to it, how many pass keep_read (:148-163, the same predicate as the snps path), and
// the gene's depth = sum over kept reads of len(query_alignment_sequence) / float(gene.length).
//
// The reference walks the (unsorted) BAM once and accumulates `gene.depth += align_len / float(gene.length)` read by
// read.  fp64 addition is not associative, so the sum has to be reproduced in exactly that order:
//   host    per read, 8 bytes: aligned length (pysam's clip rules), l_seq, NM, floor(mean quality), mapq, three
//           "absent" flags -- a pass over the quality bytes and CIGARs on all host cores;
//   filter  one thread per read, BAM order: keep_read with the exceptions the reference would raise (lowest read
//           index wins), and the read's term  align_len / float(gene.length)  (+0.0 for a read that is dropped:
//           adding it leaves the running sum unchanged bit for bit);
//   sort    stable LSD radix sort of (gene, term) pairs by gene (hipCUB): BAM order survives inside a gene;
//   bounds  first sorted position of every gene;
//   sum     one thread per gene adds its terms one after the other (a gene with many reads: one wave stages 512
//           terms at a time in LDS and adds them in the same order).
// Genes are independent, so the device parallelism of the last step is over genes (10^5 - 10^6 per sample).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/midas_snps.h"
#include "ctx_internal.h"
#include "workers.h"
#include "device_common.h"
#include "kernels.h"

namespace midas {
namespace {

// word 0: align_len (11) | l_seq (11) << 11 | flags (3) << 22      word 1: NM (16) | qmean (8) << 16 | mapq (8) << 24
// qmean = floor(mean(query_qualities)): np.mean(q) < readq  <=>  qmean < readq for an integer readq
constexpr uint32_t kNoSeq = 1, kNoNm = 2, kNoQual = 4;
constexpr int kHeavyGene = 2048;       // reads; genes above it are summed by a whole wave

struct FilterKParams {
  const uint2* rec;              // [n] BAM order
  const uint32_t* gene;          // [n] reference id = gene index
  const int64_t* gene_len;       // [n_genes]
  const FilterTables* filt;
  double* term;                  // [n] out
  unsigned long long* err;       // atomicMin((read << 8) | kind)
  long long n;
  int mapq, readq;
};

__global__ __launch_bounds__(256) void genes_filter_kernel(FilterKParams p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const uint2 r = p.rec[i];
  const int align_len = (int)(r.x & 2047u), l_seq = (int)((r.x >> 11) & 2047u);
  const uint32_t flags = r.x >> 22;
  const int nm = (int)(r.y & 0xFFFFu), qmean = (int)((r.y >> 16) & 0xFFu), mapq = (int)(r.y >> 24);
  // keep_read (genes.py:148-163): identity, mean quality, mapping quality, aligned fraction -- in that order, with
  // the exceptions the reference would raise in the order it would raise them
  uint32_t err = 0;
  bool keep = false;
  if (flags & kNoSeq) err = dev::E_NO_SEQ;                           // len(None)
  else if (flags & kNoNm) err = dev::E_NO_NM;                        // dict(aln.tags)['NM']
  else if (align_len == 0) err = dev::E_ZERO_ALIGN;                  // / float(0)
  else if (align_len - nm < p.filt->min_match[align_len]) keep = false;
  else if (flags & kNoQual) err = dev::E_NO_QUAL;                    // np.mean(None)
  else if (qmean < p.readq) keep = false;
  else if (mapq < p.mapq) keep = false;
  else if (align_len < p.filt->min_align[l_seq]) keep = false;
  else keep = true;
  if (err) atomicMin(p.err, ((unsigned long long)i << 8) | err);
  // the reference's own expression; a dropped read contributes +0.0
  p.term[i] = keep ? (double)align_len / (double)p.gene_len[p.gene[i]] : 0.0;
}

// first sorted position of every gene: begin[g] = lowest i with key[i] >= g; begin[n_genes] = n
__global__ __launch_bounds__(256) void genes_bounds_kernel(const uint32_t* key, long long n, long long n_genes, long long* begin) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i > n) return;
  const long long prev = i == 0 ? -1 : (long long)key[i - 1];
  const long long cur = i == n ? n_genes : (long long)key[i];
  for (long long g = prev + 1; g <= cur; ++g) begin[g] = i;
}

struct SumKParams {
  const double* term;            // [n] gene order, BAM order inside a gene
  const long long* begin;        // [n_genes + 1]
  long long* aligned;            // [n_genes]
  long long* mapped;
  double* depth;
  unsigned int* heavy_count;
  unsigned int* heavy;           // [n_genes] genes left to the wave kernel
  long long n_genes;
};

__global__ __launch_bounds__(256) void genes_sum_kernel(SumKParams p) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= p.n_genes) return;
  const long long lo = p.begin[g], hi = p.begin[g + 1];
  p.aligned[g] = hi - lo;
  if (hi - lo > kHeavyGene) {
    p.heavy[atomicAdd(p.heavy_count, 1u)] = (unsigned int)g;
    return;
  }
  long long mapped = 0;
  double depth = 0.0;
  for (long long i = lo; i < hi; ++i) {
    const double t = p.term[i];
    mapped += t > 0.0;
    depth += t;
  }
  p.mapped[g] = mapped;
  p.depth[g] = depth;
}

// One wave per heavy gene: 512 terms are fetched at once (coalesced, eight loads in flight per lane) and parked in LDS;
// then every lane alike reads them back in order (same address in all lanes: a broadcast, two terms per ds_read_b128)
// and adds them one after the other, so the sequence of additions is the thread-per-gene one while the memory latency
// is paid once per 512 terms instead of once per term.
__global__ __launch_bounds__(64) void genes_sum_heavy_kernel(SumKParams p) {
  constexpr int kBatch = 8;
  __shared__ __attribute__((aligned(16))) double buf[64 * kBatch];
  if (blockIdx.x >= *p.heavy_count) return;
  const long long g = p.heavy[blockIdx.x];
  const long long lo = p.begin[g], hi = p.begin[g + 1];
  const int lane = threadIdx.x;
  long long mapped = 0;
  double depth = 0.0;
  for (long long base = lo; base < hi; base += 64 * kBatch) {
    double t[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const long long i = base + j * 64 + lane;
      t[j] = i < hi ? p.term[i] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      mapped += __popcll(__ballot(t[j] > 0.0));
      buf[j * 64 + lane] = t[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const long long left = hi - base;
    const int n = left < 64 * kBatch ? (int)left : 64 * kBatch;
    const double2* pairs = reinterpret_cast<const double2*>(buf);
    int k = 0;
    for (; k + 8 <= n; k += 8) {       // (terms past the end of the gene are +0.0, but stay out of the sum anyway)
      const double2 a = pairs[k / 2], b = pairs[k / 2 + 1], c = pairs[k / 2 + 2], d = pairs[k / 2 + 3];
      depth += a.x; depth += a.y; depth += b.x; depth += b.y;
      depth += c.x; depth += c.y; depth += d.x; depth += d.y;
    }
    for (; k < n; ++k) depth += buf[k];
    __builtin_amdgcn_wave_barrier();    // the next batch overwrites buf
  }
  if (lane == 0) {
    p.mapped[g] = mapped;
    p.depth[g] = depth;
  }
}

int32_t gfail(midas_snps_ctx* ctx, int32_t st, const char* msg) {
  ctx->set_error(msg);
  return st;
}

template <class F>
void host_ranges(int64_t n, F&& fn) {
  int nt = std::min(midas::cpu_budget(), 64);
  if (n < (int64_t)1 << 15) nt = 1;
  if (nt == 1) { fn((int64_t)0, n); return; }
  std::vector<std::thread> th;
  const int64_t per = (n + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) {
    const int64_t lo = t * per, hi = std::min(n, lo + per);
    if (lo >= hi) break;
    th.emplace_back([&fn, lo, hi] { fn(lo, hi); });
  }
  for (auto& x : th) x.join();
}

}  // namespace
}  // namespace midas

using namespace midas;

#define G_TRY(call)                                                                                              \
  do {                                                                                                           \
    hipError_t e__ = (call);                                                                                     \
    if (e__ != hipSuccess) {                                                                                     \
      for (void* q__ : dev_ptrs) (void)hipFree(q__);                                                             \
      char buf__[384];                                                                                           \
      snprintf(buf__, sizeof buf__, "%s: %s", #call, hipGetErrorString(e__));                                    \
      (void)hipGetLastError();                                                                                   \
      return gfail(ctx, e__ == hipErrorOutOfMemory ? MIDAS_SNPS_ERR_OUT_OF_MEMORY : MIDAS_SNPS_ERR_HIP, buf__);  \
    }                                                                                                            \
  } while (0)

extern "C" int32_t midas_genes_count(midas_snps_ctx* ctx, const midas_snps_thresholds* thr, const midas_snps_reads* reads,
                                     const int32_t* ref_id, int64_t n_genes, const int64_t* gene_length,
                                     int64_t* out_aligned, int64_t* out_mapped, double* out_depth, float* out_kernel_ms) {
  if (!ctx || !thr || !reads || n_genes < 0 || reads->n_reads < 0 || (n_genes > 0 && (!gene_length || !out_aligned || !out_mapped || !out_depth)) ||
      (reads->n_reads > 0 && (!ref_id || !reads->mapq || !reads->nm || !reads->l_seq || !reads->qual_off || !reads->cigar_off ||
                              !reads->qual || !reads->cigar)))
    return MIDAS_SNPS_ERR_INVALID_ARG;
  ctx->clear_error();
  ctx->err_read = -1;
  if (out_kernel_ms) *out_kernel_ms = 0.f;
  const int64_t n = reads->n_reads;
  if (n > 0x7FFFFFFFll || n_genes > 0x7FFFFFFFll) return gfail(ctx, MIDAS_SNPS_ERR_UNSUPPORTED, "more than 2^31-1 reads or genes");
  // ---- host: per read, the numbers keep_read looks at (all cores) ------------------------------------------------
  std::vector<uint2> recs((size_t)n);
  std::atomic<int64_t> bad_ref{INT64_MAX}, bad_layout{INT64_MAX}, bad_size{INT64_MAX};
  std::atomic<int32_t> max_l_all{0};
  auto lower = [](std::atomic<int64_t>& a, int64_t v) {
    int64_t cur = a.load();
    while (v < cur && !a.compare_exchange_weak(cur, v)) {}
  };
  host_ranges(n, [&](int64_t lo, int64_t hi) {
    int32_t max_l = 0;
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t g = ref_id[i];
      if (g < 0 || g >= n_genes) { lower(bad_ref, i); continue; }
      const int64_t l = reads->l_seq[i];
      const int64_t nc = reads->cigar_off[i + 1] - reads->cigar_off[i];
      if (l < 0 || nc < 0 || reads->qual_off[i + 1] - reads->qual_off[i] < l) { lower(bad_layout, i); continue; }
      if (l > kMaxLSeq || reads->nm[i] > kMaxField16) { lower(bad_size, i); continue; }
      const uint32_t* cg = reads->cigar + reads->cigar_off[i];
      // [EXT] pysam query_alignment_start / _end: leading S run (hopping over H); trailing S run found by a backward
      // walk over ops n-1 .. 1 (op 0 is never inspected)
      int64_t qs = 0;
      for (int64_t k = 0; k < nc; ++k) {
        const uint32_t op = cg[k] & 15u;
        if (op == 5u) continue;
        if (op == 4u) qs += cg[k] >> 4; else break;
      }
      int64_t qe = l;
      for (int64_t k = nc - 1; k >= 1; --k) {
        const uint32_t op = cg[k] & 15u;
        if (op == 5u) continue;
        if (op == 4u) qe -= cg[k] >> 4; else break;
      }
      int64_t al = qe - qs > 0 ? qe - qs : 0;
      if (al > l) al = l;                       // (clips shorter than the read: cannot exceed it)
      const uint8_t* q = reads->qual + reads->qual_off[i];
      uint32_t qsum = 0;
      for (int64_t x = 0; x < l; ++x) qsum += q[x];
      const uint32_t flags = (l == 0 ? kNoSeq : 0u) | (reads->nm[i] < 0 ? kNoNm : 0u) | ((l > 0 && q[0] == 0xFF) ? kNoQual : 0u);
      const uint32_t nm = (uint32_t)(reads->nm[i] < 0 ? 0 : reads->nm[i]);
      const uint32_t qmean = l > 0 ? qsum / (uint32_t)l : 0u;
      recs[(size_t)i] = make_uint2((uint32_t)al | ((uint32_t)l << 11) | (flags << 22), nm | (qmean << 16) | ((uint32_t)reads->mapq[i] << 24));
      max_l = std::max<int32_t>(max_l, (int32_t)l);
    }
    int32_t cur = max_l_all.load();
    while (max_l > cur && !max_l_all.compare_exchange_weak(cur, max_l)) {}
  });
  {
    // the first malformed read in BAM order decides, as a single forward pass would
    const int64_t first = std::min(bad_ref.load(), std::min(bad_layout.load(), bad_size.load()));
    if (first != INT64_MAX) {
      ctx->err_read = first;
      char buf[200];
      if (first == bad_ref.load()) {
        snprintf(buf, sizeof buf, "read %lld: reference id %lld is not a gene of the pangenome (the reference fails in getrname / genes[...])",
                 (long long)first, (long long)ref_id[first]);
        return gfail(ctx, MIDAS_SNPS_ERR_BAD_LAYOUT, buf);
      }
      if (first == bad_layout.load()) return gfail(ctx, MIDAS_SNPS_ERR_BAD_LAYOUT, "negative size or CSR offsets shorter than l_seq");
      return gfail(ctx, MIDAS_SNPS_ERR_UNSUPPORTED, "l_seq > 1024 or NM > 65534 is not supported");
    }
  }
  FilterTables ft;
  memset(&ft, 0, sizeof ft);
  build_filter_tables(thr->mapid, thr->aln_cov, max_l_all.load(), &ft);
  // ---- device ------------------------------------------------------------------------------------------------------
  std::vector<void*> dev_ptrs;
  G_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t ng = (size_t)(n_genes > 0 ? n_genes : 1), nr = (size_t)(n > 0 ? n : 1);
  uint2* d_recs = nullptr; uint32_t* d_key = nullptr; uint32_t* d_key_sorted = nullptr; double* d_term = nullptr; double* d_term_sorted = nullptr;
  int64_t* d_len = nullptr; FilterTables* d_ft = nullptr; long long* d_begin = nullptr; long long* d_al = nullptr; long long* d_mp = nullptr;
  double* d_dp = nullptr; unsigned long long* d_err = nullptr; unsigned int* d_heavy = nullptr; void* d_tmp = nullptr;
  G_TRY(hipMalloc(&d_recs, nr * sizeof(uint2))); dev_ptrs.push_back(d_recs);
  G_TRY(hipMalloc(&d_key, nr * 4)); dev_ptrs.push_back(d_key);
  G_TRY(hipMalloc(&d_key_sorted, nr * 4)); dev_ptrs.push_back(d_key_sorted);
  G_TRY(hipMalloc(&d_term, nr * 8)); dev_ptrs.push_back(d_term);
  G_TRY(hipMalloc(&d_term_sorted, nr * 8)); dev_ptrs.push_back(d_term_sorted);
  G_TRY(hipMalloc(&d_len, ng * 8)); dev_ptrs.push_back(d_len);
  G_TRY(hipMalloc(&d_ft, sizeof(FilterTables))); dev_ptrs.push_back(d_ft);
  G_TRY(hipMalloc(&d_begin, (ng + 1) * 8)); dev_ptrs.push_back(d_begin);
  G_TRY(hipMalloc(&d_al, ng * 8)); dev_ptrs.push_back(d_al);
  G_TRY(hipMalloc(&d_mp, ng * 8)); dev_ptrs.push_back(d_mp);
  G_TRY(hipMalloc(&d_dp, ng * 8)); dev_ptrs.push_back(d_dp);
  G_TRY(hipMalloc(&d_err, 16)); dev_ptrs.push_back(d_err);
  G_TRY(hipMalloc(&d_heavy, (ng + 1) * 4)); dev_ptrs.push_back(d_heavy);
  int key_bits = 1;
  while (key_bits < 32 && ((int64_t)1 << key_bits) < n_genes) ++key_bits;
  size_t tmp_bytes = 0;
  G_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_key, d_key_sorted, d_term, d_term_sorted, (int)n, 0, key_bits, s));
  G_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16)); dev_ptrs.push_back(d_tmp);
  if (n > 0) {
    G_TRY(hipMemcpyAsync(d_recs, recs.data(), (size_t)n * sizeof(uint2), hipMemcpyHostToDevice, s));
    G_TRY(hipMemcpyAsync(d_key, ref_id, (size_t)n * 4, hipMemcpyHostToDevice, s));
  }
  if (n_genes > 0) G_TRY(hipMemcpyAsync(d_len, gene_length, (size_t)n_genes * 8, hipMemcpyHostToDevice, s));
  G_TRY(hipMemcpyAsync(d_ft, &ft, sizeof ft, hipMemcpyHostToDevice, s));
  G_TRY(hipMemsetAsync(d_err, 0xFF, 8, s));
  G_TRY(hipMemsetAsync(d_heavy, 0, 4, s));
  hipEvent_t e0, e1;
  G_TRY(hipEventCreate(&e0));
  G_TRY(hipEventCreate(&e1));
  unsigned long long err = ~0ull;
  if (n_genes > 0) {
    G_TRY(hipEventRecord(e0, s));
    if (n > 0) {
      FilterKParams f;
      f.rec = d_recs; f.gene = d_key; f.gene_len = d_len; f.filt = d_ft; f.term = d_term; f.err = d_err; f.n = n;
      f.mapq = thr->mapq; f.readq = thr->readq;
      hipLaunchKernelGGL(genes_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, f);
      G_TRY(hipGetLastError());
      G_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_key, d_key_sorted, d_term, d_term_sorted, (int)n, 0, key_bits, s));
    }
    hipLaunchKernelGGL(genes_bounds_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, s, d_key_sorted, (long long)n,
                       (long long)n_genes, d_begin);
    G_TRY(hipGetLastError());
    SumKParams k;
    k.term = d_term_sorted; k.begin = d_begin; k.aligned = d_al; k.mapped = d_mp; k.depth = d_dp;
    k.heavy_count = d_heavy; k.heavy = d_heavy + 1; k.n_genes = n_genes;
    hipLaunchKernelGGL(genes_sum_kernel, dim3((unsigned)((n_genes + 255) / 256)), dim3(256), 0, s, k);
    G_TRY(hipGetLastError());
    // at most n / kHeavyGene genes are heavy; idle waves leave at once
    const long long max_heavy = std::min<long long>(n_genes, n / kHeavyGene);
    if (max_heavy > 0) {
      hipLaunchKernelGGL(genes_sum_heavy_kernel, dim3((unsigned)max_heavy), dim3(64), 0, s, k);
      G_TRY(hipGetLastError());
    }
    G_TRY(hipEventRecord(e1, s));
    G_TRY(hipMemcpyAsync(out_aligned, d_al, (size_t)n_genes * 8, hipMemcpyDeviceToHost, s));
    G_TRY(hipMemcpyAsync(out_mapped, d_mp, (size_t)n_genes * 8, hipMemcpyDeviceToHost, s));
    G_TRY(hipMemcpyAsync(out_depth, d_dp, (size_t)n_genes * 8, hipMemcpyDeviceToHost, s));
    G_TRY(hipMemcpyAsync(&err, d_err, 8, hipMemcpyDeviceToHost, s));
    G_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    G_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (out_kernel_ms) *out_kernel_ms = ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (void* q : dev_ptrs) (void)hipFree(q);
  if (err != ~0ull) {
    const int32_t kind = (int32_t)(err & 0xFF);
    ctx->err_read = (int64_t)(err >> 8);
    char buf[200];
    snprintf(buf, sizeof buf, "read %lld: keep_read would raise (%s)", (long long)ctx->err_read,
             kind == 1 ? "no SEQ: TypeError" : kind == 2 ? "no NM tag: KeyError" : kind == 3 ? "aligned length 0: ZeroDivisionError"
                                                                                                : "no QUAL: TypeError");
    ctx->set_error(buf);
    return kind;
  }
  return MIDAS_SNPS_OK;
}
