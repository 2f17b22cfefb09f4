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
//   sort    stable LSD radix sort of (gene, term) pairs by gene, eight bits of the gene index a pass (device_sort.hip, the
//           library's own: a gene index has as many bits as the pangenome has genes): BAM order survives inside a gene;
//   bounds  first sorted position of every gene;
//   sum     one thread per gene adds its terms one after the other (a gene with many reads: one wave stages 512
//           terms at a time in LDS and adds them in the same order).
// Genes are independent, so the device parallelism of the last step is over genes (10^5 - 10^6 per sample).
//
// N ranks (run/genes.py): the two halves are entry points of their own -- midas_genes_terms (host + filter: a rank's slice of
// the BAM -> one term per read) and midas_genes_sum (sort + bounds + sum over the pairs a gene's owner received, which arrive
// in BAM order) -- so that a gene's running sum is formed on ONE rank in the order the reference forms it.
#include <hip/hip_runtime.h>

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

namespace {

// The device buffers of one call; freed when it goes out of scope.
struct DevBufs {
  std::vector<void*> ptrs;
  ~DevBufs() { for (void* q : ptrs) (void)hipFree(q); }
  template <class T> hipError_t get(T** out, size_t bytes) {
    void* q = nullptr;
    const hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
    if (e == hipSuccess) ptrs.push_back(q);
    *out = static_cast<T*>(q);
    return e;
  }
};

#define G_TRY(call)                                                                                              \
  do {                                                                                                           \
    hipError_t e__ = (call);                                                                                     \
    if (e__ != hipSuccess) {                                                                                     \
      char buf__[384];                                                                                           \
      snprintf(buf__, sizeof buf__, "%s: %s", #call, hipGetErrorString(e__));                                    \
      (void)hipGetLastError();                                                                                   \
      return gfail(ctx, e__ == hipErrorOutOfMemory ? MIDAS_SNPS_ERR_OUT_OF_MEMORY : MIDAS_SNPS_ERR_HIP, buf__);  \
    }                                                                                                            \
  } while (0)

// host: per read, the numbers keep_read looks at (all cores).  0, or the status of the first malformed read in BAM order.
int32_t pack_records(midas_snps_ctx* ctx, const midas_snps_reads* reads, const int32_t* ref_id, int64_t n_genes, std::vector<uint2>* recs,
                     int32_t* max_l_out) {
  const int64_t n = reads->n_reads;
  recs->resize((size_t)n);
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
      (*recs)[(size_t)i] = make_uint2((uint32_t)al | ((uint32_t)l << 11) | (flags << 22), nm | (qmean << 16) | ((uint32_t)reads->mapq[i] << 24));
      max_l = std::max<int32_t>(max_l, (int32_t)l);
    }
    int32_t cur = max_l_all.load();
    while (max_l > cur && !max_l_all.compare_exchange_weak(cur, max_l)) {}
  });
  *max_l_out = max_l_all.load();
  // the first malformed read in BAM order decides, as a single forward pass would
  const int64_t first = std::min(bad_ref.load(), std::min(bad_layout.load(), bad_size.load()));
  if (first == INT64_MAX) return MIDAS_SNPS_OK;
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

int32_t raise_status(midas_snps_ctx* ctx, unsigned long long err) {
  const int32_t kind = (int32_t)(err & 0xFF);
  ctx->err_read = (int64_t)(err >> 8);
  char buf[200];
  snprintf(buf, sizeof buf, "read %lld: keep_read would raise (%s)", (long long)ctx->err_read,
           kind == 1 ? "no SEQ: TypeError" : kind == 2 ? "no NM tag: KeyError" : kind == 3 ? "aligned length 0: ZeroDivisionError"
                                                                                              : "no QUAL: TypeError");
  ctx->set_error(buf);
  return kind;
}

// One call, either half or both.  reads != nullptr: the terms are made here (host records + filter kernel) from the reads and
// their genes `gene` (= ref_id); else `term_in` holds them.  out_term != nullptr: they are handed back (and, with no sums asked
// for, that is all).  out_aligned != nullptr: sort + bounds + sums.
int32_t genes_run(midas_snps_ctx* ctx, const midas_snps_thresholds* thr, const midas_snps_reads* reads, int64_t n, const int32_t* gene,
                  const double* term_in, int64_t n_genes, const int64_t* gene_length, double* out_term, int64_t* out_aligned,
                  int64_t* out_mapped, double* out_depth, float* out_kernel_ms) {
  ctx->clear_error();
  ctx->err_read = -1;
  if (out_kernel_ms) *out_kernel_ms = 0.f;
  if (n > 0x7FFFFFFFll || n_genes > 0x7FFFFFFFll) return gfail(ctx, MIDAS_SNPS_ERR_UNSUPPORTED, "more than 2^31-1 reads or genes");
  const bool filter = reads != nullptr, sums = out_aligned != nullptr;
  std::vector<uint2> recs;
  FilterTables ft;
  memset(&ft, 0, sizeof ft);
  if (filter) {
    int32_t max_l = 0;
    const int32_t st = pack_records(ctx, reads, gene, n_genes, &recs, &max_l);
    if (st != MIDAS_SNPS_OK) return st;
    build_filter_tables(thr->mapid, thr->aln_cov, max_l, &ft);
  } else {
    for (int64_t i = 0; i < n; ++i)
      if (gene[i] < 0 || gene[i] >= n_genes) {
        ctx->err_read = i;
        return gfail(ctx, MIDAS_SNPS_ERR_BAD_LAYOUT, "a pair's gene index is outside the gene table");
      }
  }
  // ---- device ------------------------------------------------------------------------------------------------------
  DevBufs dev;
  G_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t ng = (size_t)(n_genes > 0 ? n_genes : 1), nr = (size_t)(n > 0 ? n : 1);
  uint2* d_recs = nullptr; uint32_t* d_key = nullptr; uint32_t* d_key_b = nullptr; double* d_term = nullptr; double* d_term_b = nullptr;
  int64_t* d_len = nullptr; FilterTables* d_ft = nullptr; long long* d_begin = nullptr; long long* d_al = nullptr; long long* d_mp = nullptr;
  double* d_dp = nullptr; unsigned long long* d_err = nullptr; unsigned int* d_heavy = nullptr; uint32_t* d_hist = nullptr;
  G_TRY(dev.get(&d_key, nr * 4));
  G_TRY(dev.get(&d_term, nr * 8));
  G_TRY(dev.get(&d_err, 16));
  if (filter) {
    G_TRY(dev.get(&d_recs, nr * sizeof(uint2)));
    G_TRY(dev.get(&d_len, ng * 8));
    G_TRY(dev.get(&d_ft, sizeof(FilterTables)));
  }
  if (sums) {
    G_TRY(dev.get(&d_key_b, nr * 4));
    G_TRY(dev.get(&d_term_b, nr * 8));
    G_TRY(dev.get(&d_hist, sort_scratch_words((long long)nr) * 4));
    G_TRY(dev.get(&d_begin, (ng + 1) * 8));
    G_TRY(dev.get(&d_al, ng * 8));
    G_TRY(dev.get(&d_mp, ng * 8));
    G_TRY(dev.get(&d_dp, ng * 8));
    G_TRY(dev.get(&d_heavy, (ng + 1) * 4));
  }
  int key_bits = 1;
  while (key_bits < 32 && ((int64_t)1 << key_bits) < n_genes) ++key_bits;
  if (n > 0) {
    G_TRY(hipMemcpyAsync(d_key, gene, (size_t)n * 4, hipMemcpyHostToDevice, s));
    if (filter) G_TRY(hipMemcpyAsync(d_recs, recs.data(), (size_t)n * sizeof(uint2), hipMemcpyHostToDevice, s));
    else G_TRY(hipMemcpyAsync(d_term, term_in, (size_t)n * 8, hipMemcpyHostToDevice, s));
  }
  if (filter) {
    if (n_genes > 0) G_TRY(hipMemcpyAsync(d_len, gene_length, (size_t)n_genes * 8, hipMemcpyHostToDevice, s));
    G_TRY(hipMemcpyAsync(d_ft, &ft, sizeof ft, hipMemcpyHostToDevice, s));
  }
  G_TRY(hipMemsetAsync(d_err, 0xFF, 8, s));
  if (sums) G_TRY(hipMemsetAsync(d_heavy, 0, 4, s));
  struct EvGuard {      // (constructed before either event exists: a failing second create must not leak the first)
    hipEvent_t a = nullptr, b = nullptr;
    ~EvGuard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } evg;
  G_TRY(hipEventCreate(&evg.a));
  G_TRY(hipEventCreate(&evg.b));
  const hipEvent_t e0 = evg.a, e1 = evg.b;
  unsigned long long err = ~0ull;
  G_TRY(hipEventRecord(e0, s));
  if (filter && n > 0) {
    FilterKParams f;
    f.rec = d_recs; f.gene = d_key; f.gene_len = d_len; f.filt = d_ft; f.term = d_term; f.err = d_err; f.n = n;
    f.mapq = thr->mapq; f.readq = thr->readq;
    hipLaunchKernelGGL(genes_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, f);
    G_TRY(hipGetLastError());
  }
  if (out_term && n > 0) G_TRY(hipMemcpyAsync(out_term, d_term, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  if (sums && n_genes > 0) {
    uint32_t* d_key_sorted = d_key;
    double* d_term_sorted = d_term;
    G_TRY(launch_sort_pairs_f64(d_key, d_term, d_key_b, d_term_b, n, key_bits, d_hist, s, &d_key_sorted, &d_term_sorted));
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
  } else {
    G_TRY(hipEventRecord(e1, s));
  }
  G_TRY(hipMemcpyAsync(&err, d_err, 8, hipMemcpyDeviceToHost, s));
  G_TRY(hipStreamSynchronize(s));
  float ms = 0.f;
  G_TRY(hipEventElapsedTime(&ms, e0, e1));
  if (out_kernel_ms) *out_kernel_ms = ms;
  if (err != ~0ull) return raise_status(ctx, err);
  return MIDAS_SNPS_OK;
}

bool reads_ok(const midas_snps_reads* reads, const int32_t* ref_id) {
  return reads && reads->n_reads >= 0 &&
         (reads->n_reads == 0 || (ref_id && reads->mapq && reads->nm && reads->l_seq && reads->qual_off && reads->cigar_off && reads->qual && reads->cigar));
}

}  // namespace

extern "C" int32_t midas_genes_count(midas_snps_ctx* ctx, const midas_snps_thresholds* thr, const midas_snps_reads* reads,
                                     const int32_t* ref_id, int64_t n_genes, const int64_t* gene_length,
                                     int64_t* out_aligned, int64_t* out_mapped, double* out_depth, float* out_kernel_ms) {
  if (!ctx || !thr || n_genes < 0 || !reads_ok(reads, ref_id) || (n_genes > 0 && (!gene_length || !out_aligned || !out_mapped || !out_depth)))
    return MIDAS_SNPS_ERR_INVALID_ARG;
  int64_t dummy_a = 0, dummy_m = 0;
  double dummy_d = 0.0;
  return genes_run(ctx, thr, reads, reads->n_reads, ref_id, nullptr, n_genes, gene_length, nullptr, n_genes > 0 ? out_aligned : &dummy_a,
                   n_genes > 0 ? out_mapped : &dummy_m, n_genes > 0 ? out_depth : &dummy_d, out_kernel_ms);
}

extern "C" int32_t midas_genes_terms(midas_snps_ctx* ctx, const midas_snps_thresholds* thr, const midas_snps_reads* reads,
                                     const int32_t* ref_id, int64_t n_genes, const int64_t* gene_length, double* out_term,
                                     float* out_kernel_ms) {
  if (!ctx || !thr || n_genes < 0 || !reads_ok(reads, ref_id) || (n_genes > 0 && !gene_length) || (reads->n_reads > 0 && !out_term))
    return MIDAS_SNPS_ERR_INVALID_ARG;
  return genes_run(ctx, thr, reads, reads->n_reads, ref_id, nullptr, n_genes, gene_length, out_term, nullptr, nullptr, nullptr, out_kernel_ms);
}

extern "C" int32_t midas_genes_sum(midas_snps_ctx* ctx, int64_t n_pairs, const int32_t* gene, const double* term, int64_t n_genes,
                                   int64_t* out_aligned, int64_t* out_mapped, double* out_depth, float* out_kernel_ms) {
  if (!ctx || n_pairs < 0 || n_genes < 0 || (n_pairs > 0 && (!gene || !term)) || (n_genes > 0 && (!out_aligned || !out_mapped || !out_depth)))
    return MIDAS_SNPS_ERR_INVALID_ARG;
  int64_t dummy_a = 0, dummy_m = 0;
  double dummy_d = 0.0;
  return genes_run(ctx, nullptr, nullptr, n_pairs, gene, term, n_genes, nullptr, nullptr, n_genes > 0 ? out_aligned : &dummy_a,
                   n_genes > 0 ? out_mapped : &dummy_m, n_genes > 0 ? out_depth : &dummy_d, out_kernel_ms);
}
