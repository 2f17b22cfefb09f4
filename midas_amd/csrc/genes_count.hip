// `run_midas.py genes` on MI355X: count_mapped_bp (/root/reference/midas/run/genes.py:165-189) -- for every gene of the
// pangenome, how many reads align to it, how many pass keep_read (:148-163, the same predicate as the snps path), and
// the gene's depth = sum over kept reads of len(query_alignment_sequence) / float(gene.length).
//
// The reference walks the (unsorted) BAM once and accumulates `gene.depth += align_len / float(gene.length)` read by
// read.  fp64 addition is not associative, so the sum is reproduced in exactly that order: the host groups the reads
// by gene with a stable sort (BAM order inside a gene is kept) and one device thread per gene adds its reads' terms
// one after the other.  Genes are independent, so the device parallelism is over genes (10^5 - 10^6 per sample).
// Per read the kernel needs 12 bytes: aligned length, l_seq, NM, floor(mean quality), mapq and three "absent" flags,
// all derived on the host from the BAM record with pysam's rules (query_alignment_start/end from the CIGAR clips).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/midas_snps.h"
#include "ctx_internal.h"
#include "device_common.h"
#include "kernels.h"

namespace midas {
namespace {

struct GeneRead {            // 12 bytes per read, in gene order
  uint32_t orig;             // index of the read in the caller's arrays (BAM order), for error reports
  uint16_t align_len;        // len(aln.query_alignment_sequence)
  uint16_t l_seq;            // aln.query_length
  uint16_t nm;               // NM tag
  uint8_t qmean;             // floor(mean(query_qualities)): np.mean(q) < readq  <=>  qmean < readq for an integer readq
  uint8_t mapq_flags;        // unused
};
static_assert(sizeof(GeneRead) == 12, "GeneRead must be 12 bytes");
struct GeneReadAux { uint8_t mapq; uint8_t flags; };   // flags: 1 no SEQ, 2 no NM, 4 no QUAL
constexpr uint8_t kNoSeq = 1, kNoNm = 2, kNoQual = 4;

struct GenesKParams {
  const GeneRead* reads;
  const GeneReadAux* aux;
  const int64_t* gene_begin;     // [n_genes + 1] into reads
  const int64_t* gene_len;       // [n_genes]
  const FilterTables* filt;
  long long* aligned;            // [n_genes]
  long long* mapped;
  double* depth;
  unsigned long long* err;       // atomicMin((orig << 8) | kind)
  long long n_genes;
  int mapq, readq;
};

__global__ __launch_bounds__(256) void genes_count_kernel(GenesKParams p) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= p.n_genes) return;
  const long long lo = p.gene_begin[g], hi = p.gene_begin[g + 1];
  const double glen = (double)p.gene_len[g];
  long long mapped = 0;
  double depth = 0.0;
  for (long long i = lo; i < hi; ++i) {
    const GeneRead r = p.reads[i];
    const GeneReadAux x = p.aux[i];
    // keep_read (genes.py:148-163): identity, mean quality, mapping quality, aligned fraction -- in that order,
    // with the exceptions the reference would raise in the order it would raise them
    uint32_t err = 0;
    bool keep = false;
    if (x.flags & kNoSeq) err = dev::E_NO_SEQ;                         // len(None)
    else if (x.flags & kNoNm) err = dev::E_NO_NM;                      // dict(aln.tags)['NM']
    else if (r.align_len == 0) err = dev::E_ZERO_ALIGN;                // / float(0)
    else if ((int)r.align_len - (int)r.nm < p.filt->min_match[r.align_len]) keep = false;
    else if (x.flags & kNoQual) err = dev::E_NO_QUAL;                  // np.mean(None)
    else if ((int)r.qmean < p.readq) keep = false;
    else if ((int)x.mapq < p.mapq) keep = false;
    else if ((int)r.align_len < p.filt->min_align[r.l_seq]) keep = false;
    else keep = true;
    if (err) { atomicMin(p.err, ((unsigned long long)r.orig << 8) | err); continue; }
    if (keep) {
      ++mapped;
      depth += (double)r.align_len / glen;       // the reference's own expression, accumulated in BAM order
    }
  }
  p.aligned[g] = hi - lo;
  p.mapped[g] = mapped;
  p.depth[g] = depth;
}

int32_t gfail(midas_snps_ctx* ctx, int32_t st, const char* msg) {
  ctx->err = msg;
  return st;
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
  ctx->err.clear();
  ctx->err_read = -1;
  if (out_kernel_ms) *out_kernel_ms = 0.f;
  const int64_t n = reads->n_reads;
  // ---- host: per read, the numbers keep_read looks at; reads grouped by gene, BAM order kept inside a gene ------------
  std::vector<int64_t> begin((size_t)n_genes + 1, 0);
  for (int64_t i = 0; i < n; ++i) {
    const int64_t g = ref_id[i];
    if (g < 0 || g >= n_genes) {
      char buf[160];
      snprintf(buf, sizeof buf, "read %lld: reference id %lld is not a gene of the pangenome (the reference fails in getrname / genes[...])",
               (long long)i, (long long)g);
      ctx->err_read = i;
      return gfail(ctx, MIDAS_SNPS_ERR_BAD_LAYOUT, buf);
    }
    begin[(size_t)g + 1]++;
  }
  for (int64_t g = 0; g < n_genes; ++g) begin[(size_t)g + 1] += begin[(size_t)g];
  std::vector<int64_t> cursor(begin.begin(), begin.end() - 1);
  std::vector<GeneRead> recs((size_t)n);
  std::vector<GeneReadAux> aux((size_t)n);
  int32_t max_l = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t l = reads->l_seq[i];
    const int64_t nc = reads->cigar_off[i + 1] - reads->cigar_off[i];
    if (l < 0 || nc < 0 || reads->qual_off[i + 1] - reads->qual_off[i] < l) {
      ctx->err_read = i;
      return gfail(ctx, MIDAS_SNPS_ERR_BAD_LAYOUT, "negative size or CSR offsets shorter than l_seq");
    }
    if (l > kMaxLSeq || reads->nm[i] > kMaxField16) {
      ctx->err_read = i;
      return gfail(ctx, MIDAS_SNPS_ERR_UNSUPPORTED, "l_seq > 1024 or NM > 65534 is not supported");
    }
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
    const int64_t al = qe - qs > 0 ? qe - qs : 0;
    const uint8_t* q = reads->qual + reads->qual_off[i];
    uint64_t qsum = 0;
    for (int64_t x = 0; x < l; ++x) qsum += q[x];
    GeneRead r;
    r.orig = (uint32_t)i;
    r.align_len = (uint16_t)(al > 65535 ? 65535 : al);
    r.l_seq = (uint16_t)l;
    r.nm = (uint16_t)(reads->nm[i] < 0 ? 0 : reads->nm[i]);
    r.qmean = (uint8_t)(l > 0 ? qsum / (uint64_t)l : 0);
    r.mapq_flags = 0;
    GeneReadAux a;
    a.mapq = reads->mapq[i];
    a.flags = (uint8_t)((l == 0 ? kNoSeq : 0) | (reads->nm[i] < 0 ? kNoNm : 0) | ((l > 0 && q[0] == 0xFF) ? kNoQual : 0));
    const int64_t d = cursor[(size_t)ref_id[i]]++;
    recs[(size_t)d] = r;
    aux[(size_t)d] = a;
    max_l = std::max<int32_t>(max_l, (int32_t)l);
  }
  FilterTables ft;
  memset(&ft, 0, sizeof ft);
  build_filter_tables(thr->mapid, thr->aln_cov, max_l, &ft);
  // ---- device ------------------------------------------------------------------------------------------------------
  std::vector<void*> dev_ptrs;
  G_TRY(hipSetDevice(ctx->device));
  GeneRead* d_recs = nullptr; GeneReadAux* d_aux = nullptr; int64_t* d_begin = nullptr; int64_t* d_len = nullptr;
  FilterTables* d_ft = nullptr; long long* d_al = nullptr; long long* d_mp = nullptr; double* d_dp = nullptr; unsigned long long* d_err = nullptr;
  const size_t ng = (size_t)(n_genes > 0 ? n_genes : 1), nr = (size_t)(n > 0 ? n : 1);
  G_TRY(hipMalloc(&d_recs, nr * sizeof(GeneRead))); dev_ptrs.push_back(d_recs);
  G_TRY(hipMalloc(&d_aux, nr * sizeof(GeneReadAux))); dev_ptrs.push_back(d_aux);
  G_TRY(hipMalloc(&d_begin, (ng + 1) * 8)); dev_ptrs.push_back(d_begin);
  G_TRY(hipMalloc(&d_len, ng * 8)); dev_ptrs.push_back(d_len);
  G_TRY(hipMalloc(&d_ft, sizeof(FilterTables))); dev_ptrs.push_back(d_ft);
  G_TRY(hipMalloc(&d_al, ng * 8)); dev_ptrs.push_back(d_al);
  G_TRY(hipMalloc(&d_mp, ng * 8)); dev_ptrs.push_back(d_mp);
  G_TRY(hipMalloc(&d_dp, ng * 8)); dev_ptrs.push_back(d_dp);
  G_TRY(hipMalloc(&d_err, 8)); dev_ptrs.push_back(d_err);
  hipStream_t s = ctx->stream;
  if (n > 0) {
    G_TRY(hipMemcpyAsync(d_recs, recs.data(), (size_t)n * sizeof(GeneRead), hipMemcpyHostToDevice, s));
    G_TRY(hipMemcpyAsync(d_aux, aux.data(), (size_t)n * sizeof(GeneReadAux), hipMemcpyHostToDevice, s));
  }
  G_TRY(hipMemcpyAsync(d_begin, begin.data(), ((size_t)n_genes + 1) * 8, hipMemcpyHostToDevice, s));
  if (n_genes > 0) G_TRY(hipMemcpyAsync(d_len, gene_length, (size_t)n_genes * 8, hipMemcpyHostToDevice, s));
  G_TRY(hipMemcpyAsync(d_ft, &ft, sizeof ft, hipMemcpyHostToDevice, s));
  G_TRY(hipMemsetAsync(d_err, 0xFF, 8, s));
  hipEvent_t e0, e1;
  G_TRY(hipEventCreate(&e0));
  G_TRY(hipEventCreate(&e1));
  unsigned long long err = ~0ull;
  if (n_genes > 0) {
    GenesKParams k;
    k.reads = d_recs; k.aux = d_aux; k.gene_begin = d_begin; k.gene_len = d_len; k.filt = d_ft;
    k.aligned = d_al; k.mapped = d_mp; k.depth = d_dp; k.err = d_err; k.n_genes = n_genes; k.mapq = thr->mapq; k.readq = thr->readq;
    G_TRY(hipEventRecord(e0, s));
    hipLaunchKernelGGL(genes_count_kernel, dim3((unsigned)((n_genes + 255) / 256)), dim3(256), 0, s, k);
    G_TRY(hipGetLastError());
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
    ctx->err = buf;
    return kind;
  }
  return MIDAS_SNPS_OK;
}
