// C-ABI runtime of libmidas_snps_hip.so: contexts, resident batches, launches.  See include/midas_snps.h
// for the contract and the reference lines each entry point replaces.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <thread>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/midas_snps.h"
#include "ctx_internal.h"
#include "kernels.h"
#include "layout.h"
#include "hostio.h"
#include "contigs.h"
#include "workers.h"

#include <algorithm>

using namespace midas;

// Page-locked host memory the device writes and host threads read: non-coherent (coarse-grained) allocations are ordinary
// cached memory to the CPU -- the device's writes are visible once the stream has been synchronised, which every user
// here does -- where the default, coherent kind is mapped uncached on these hosts: the row formatter read its input
// 20 x slower from it (68 ms for 16 gzip members).
#ifndef MIDAS_SNPS_HOST_ALLOC_FLAGS
#define MIDAS_SNPS_HOST_ALLOC_FLAGS hipHostMallocNonCoherent
#endif
constexpr unsigned int kHostAllocFlags = MIDAS_SNPS_HOST_ALLOC_FLAGS;

struct midas_snps_batch {
  midas_snps_ctx* ctx = nullptr;
  // device: the caller's BAM-native SoA, uploaded as it is (the device packer's input)
  int32_t* d_pos = nullptr;
  uint8_t* d_mapq = nullptr;
  int32_t* d_nm = nullptr;
  int32_t* d_lseq = nullptr;
  int64_t* d_seq_off = nullptr;
  int64_t* d_qual_off = nullptr;
  int64_t* d_cigar_off = nullptr;
  uint8_t* d_seq4 = nullptr;
  uint8_t* d_qual = nullptr;
  uint32_t* d_cigar = nullptr;
  int64_t seq_bytes = 0, qual_bytes = 0, n_cigar = 0;
  // midas_snps_batch_create_resident: the reads are a run of a device-decoded BAM's records.  The columns above and the direct
  // layout below point INTO the BAM handle's device memory (not owned: the caller keeps the handle open); SEQ / QUAL / CIGAR
  // as columns do not exist until a path that reads them asks (ensure_raw_payload: cut out of the handle's inflated stream
  // into raw_owned, the three pointers biased so that the BAM's absolute CSR offsets index them)
  bool resident = false;
  midas::ResidentReads rr;
  int64_t rr_first = 0;
  void* raw_owned = nullptr;
  // device: scratch of the packer (one slab each: per read, per record)
  uint8_t* d_pack_reads = nullptr;    // nseg, cnt, first, facts, bin_start, tile_extra, tile_reads
  uint8_t* d_pack_recs = nullptr;     // sort keys / values (in + out), bytes8, dest, bytes8_dev, off8
  void* d_sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  int key_bits = 1;
  PackParams pk;                      // every pointer of a pack, filled once the buffers exist
  uint32_t* h_tile_reads = nullptr;   // pinned: reads every tile will see, fetched once per pack for the hot-spot plan
  std::vector<int64_t> h_contig_site; // [n_contigs + 1] first site of every contig in d_counts / d_allele
  std::vector<int64_t> h_origin;      // [n_contigs] midas_snps_contigs.origin, empty when it was NULL
  std::vector<uint32_t> h_items;      // the work items last uploaded
  int64_t pack_count = 0;
  // device: the packed layout (layout.h)
  ReadRec* d_rec = nullptr;
  uint8_t* d_blob = nullptr;
  uint8_t* d_ref = nullptr;
  Tile* d_tiles = nullptr;
  int32_t* d_contig_read_begin = nullptr;
  int32_t* d_contig_tile_base = nullptr;
  int32_t* d_contig_len = nullptr;
  uint8_t* d_work = nullptr;  // [rbinv n_tiles][rend n_tiles][stats n_species*4 u64][err u64]
  FilterTables* d_filt = nullptr;
  bool filt_fits16 = true;       // every entry of the tables fits 16 bits (the long-overhang instantiation of the direct kernel)
  uint32_t* d_orig = nullptr;   // device record -> input index (for error reports)
  uint32_t* d_key = nullptr;    // device record -> tile << 7 | reach << 2 | class (input of the index kernel)
  uint32_t* d_items = nullptr;  // [n_items][4] work items {tile, part, n_parts, 0} of the pileup kernel
  size_t items_cap = 0;         // words d_items can hold
  uint32_t* d_ticket = nullptr; // [n_tiles] arrival counters of split tiles
  uint32_t* d_wg_begin = nullptr;   // [n_stream_wgs + 1] first tile of every workgroup of the streaming kernel
  uint8_t* d_tile_split = nullptr;  // [n_tiles] 1 = processed as parts (only allocated when a tile is split)
  std::vector<uint32_t> h_wg_begin;
  std::vector<uint8_t> h_tile_split;
  bool any_split = false;
  bool has_high_qual = false;   // the payload holds a quality above kMaxPackedQual: a baseq above it cannot be served
  int64_t n_items = 0, n_whole_items = 0;
  std::vector<std::pair<size_t, size_t>> zero_ranges;   // (first site, sites) of split tiles: counts zeroed before a run
  FilterTables h_filt;
  bool filt_valid = false;
  double filt_mapid = 0, filt_aln_cov = 0;
  int32_t max_l_seq = 0;
  size_t work_bytes = 0;
  uint32_t* d_counts = nullptr;
  uint8_t* d_allele = nullptr;
  // facts
  int64_t n_reads = 0, n_sites = 0, n_tiles = 0, blob_bytes = 0, alg_bytes = 0;
  int64_t n_records = 0;   // device records (match segments + reads that keep their CIGAR) >= n_reads
  int32_t n_contigs = 0, n_species = 0, lanes_per_read = 1, lane_bases = 31, tile_len = kTileSites;
  // ---- direct path (index_direct.hip + pileup_direct.hip): the pileup kernel reads the raw arrays above -----------------
  int path = MIDAS_SNPS_PATH_DIRECT;   // the path batch_run takes
  int path_auto = MIDAS_SNPS_PATH_DIRECT;   // what the batch's own numbers recommend
  bool packed_built = false;    // rec / blob / orig / key exist (the packed path's layout is built on first use)
  uint32_t* d_trange = nullptr; // [2 parities][tbegin n_tiles][tend n_tiles]
  DirectFacts* d_dfacts = nullptr;
  DirectBlockCursor* d_block_contig = nullptr;   // per workgroup of the direct path's index kernels: the contig of its first read
  DirectRec* d_drec = nullptr;         // [n_reads + 1] the direct layout: one 16-byte record per read ...
  uint8_t* d_dpay = nullptr;           // ... and its CIGAR / SEQ / QUAL bytes as one run (layout.h)
  unsigned long long* d_dunits = nullptr;   // (scratch of the layout's scan)
  int64_t direct_payload_bytes = 0;
  int32_t layout_build_us = 0;         // device time of the layout's three launches (batch_get_info)
  // reads that span more than the overhang (a long deletion, an N skip): listed once (facts pass), so that the ranges pass keeps
  // to the common span and only the chunks they touch are dealt tile by tile
  DirectOutlier* d_outliers = nullptr;
  uint32_t* d_n_outliers = nullptr;
  uint8_t* d_tile_flag = nullptr;
  uint8_t* d_chunk_ok = nullptr;       // per chunk of kDirectChunkTiles tiles (nullptr: every chunk carries its overhang)
  uint32_t outlier_cap = 0, n_outliers_listed = 0;
  int64_t n_outliers = 0;
  int32_t direct_reach_ranges = 1;     // what the ranges pass reaches back over: the common span when the outliers are listed, else direct_reach
  int32_t direct_overhang = kDirectOverhang;     // the batch's overhang: kDirectOverhangLong when its longest read asks for it (direct_prepare)
  bool direct_chunks = false;          // chunked tiles (position-sorted reads, outliers listed)
  unsigned long long* d_probe = nullptr;   // developer builds only (MIDAS_SNPS_DEBUG_BITS & 256)
  int64_t direct_general = 0;   // reads the pileup kernel walks op by op (facts pass)
  int32_t direct_reach = 1;     // longest reference span of a read: what a tile's range must reach back over
  int32_t direct_lane_bases = 30, direct_lanes_per_read = 1;
  int64_t direct_stream_reads = 0;   // sum over tiles of the positions their streams hold (>= n_reads: straddlers twice)
  int64_t direct_max_tile_reads = 0;
  int64_t direct_run_count = 0;
  bool pad_advances = false;    // the context's pad rule when the batch was created (MIDAS_SNPS_PAD_PYSAM)
  int64_t n_long = 0;           // reads beyond the fast paths' limits: > 0 and the batch runs on the long path only (pileup_long.hip)
  bool direct_sorted = false;   // every contig's reads in position order (found by the first index pass): tile ranges need no atomics
  // timing
  std::vector<hipEvent_t> ev;  // 3 per slot: before the index kernel, before and after the pileup kernel
  std::vector<hipEvent_t> pev; // 3 per slot: before the pack, before and after its scatter kernel
  int32_t timing_slots = 0;
  bool timing_pileup_only = false;
  int64_t timed_runs = 0, timed_packs = 0;
  int64_t run_count = 0;
  bool ran = false;
};

namespace midas {
// The reference's own expressions (midas/run/snps.py:148, 157), evaluated in IEEE fp64 exactly as
// Python does, tabulated over every length a batch can contain.  Both predicates are monotone in
// the integer being tested, so the least passing value is found by bisection.
//   min_match[a]: least x = align_len - NM in [-65535, a] with !(100*x/float(a) < mapid); a+1 if none
//   min_align[l]: least a in [1, l] with !(a/float(l) < aln_cov);                          l+1 if none
void build_filter_tables(double mapid, double aln_cov, int32_t max_l, FilterTables* t) {
  t->min_match[0] = 0;
  t->min_align[0] = 0;
  for (int32_t a = 1; a <= max_l && a <= kMaxLSeq; ++a) {
    auto pass_pid = [&](long long x) { return !((double)(100LL * x) / (double)a < mapid); };
    long long lo = -65536, hi = (long long)a + 1;  // invariant: pass(hi) (virtually) true, pass(lo) false
    if (pass_pid(-65535)) {
      t->min_match[a] = -65535;
    } else {
      lo = -65535;
      if (!pass_pid(a)) {
        t->min_match[a] = a + 1;
      } else {
        hi = a;
        while (hi - lo > 1) {
          const long long mid = lo + (hi - lo) / 2;
          if (pass_pid(mid)) hi = mid; else lo = mid;
        }
        t->min_match[a] = (int32_t)hi;
      }
    }
    const int32_t l = a;
    auto pass_cov = [&](long long x) { return !((double)x / (double)l < aln_cov); };
    if (pass_cov(1)) {
      t->min_align[l] = 1;
    } else if (!pass_cov(l)) {
      t->min_align[l] = l + 1;
    } else {
      long long lo2 = 1, hi2 = l;
      while (hi2 - lo2 > 1) {
        const long long mid = lo2 + (hi2 - lo2) / 2;
        if (pass_cov(mid)) hi2 = mid; else lo2 = mid;
      }
      t->min_align[l] = (int32_t)hi2;
    }
  }
}

}  // namespace midas

namespace {

int32_t fail(midas_snps_ctx* ctx, int32_t st, const std::string& msg) {
  if (ctx) {
    ctx->set_error(msg);
  }
  return st;
}

int32_t hip_fail(midas_snps_ctx* ctx, hipError_t e, const char* what) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  (void)hipGetLastError();
  return fail(ctx, e == hipErrorOutOfMemory ? MIDAS_SNPS_ERR_OUT_OF_MEMORY : MIDAS_SNPS_ERR_HIP, buf);
}

#define HIP_TRY(ctx, call)                                   \
  do {                                                       \
    hipError_t e__ = (call);                                 \
    if (e__ != hipSuccess) return hip_fail(ctx, e__, #call); \
  } while (0)

// Device -> host copy into caller memory.  Pinned destinations (midas_snps_host_alloc, or anything the caller registered
// with HIP) take one DMA; pageable ones go through the context's pinned ring, 32 MiB at a time, the DMA of chunk k + 1
// running while host threads move chunk k out (HIP's own pageable path reaches ~12 GB/s here, this one ~3x that).
void parallel_copy(uint8_t* dst, const uint8_t* src, size_t n) {
  const unsigned hw = (unsigned)midas::cpu_budget();
  size_t nt = hw >= 32 ? 12 : (hw >= 8 ? 4 : 1);
  if (n < ((size_t)4 << 20)) nt = 1;
  if (nt == 1) { memcpy(dst, src, n); return; }
  const size_t per = ((n + nt - 1) / nt + 63) & ~(size_t)63;
  std::atomic<size_t> next{0};
  midas::Workers::run((int)nt, [&] {
    for (;;) {
      const size_t lo = next.fetch_add(1) * per, hi = lo + per < n ? lo + per : n;
      if (lo >= hi) return;
      memcpy(dst + lo, src + lo, hi - lo);
    }
  });
}

// Results into page-locked host memory by the shader cores: stores to the mapped host pointer cross the link as posted
// writes.  The runtime's own device-to-host copy goes through the SDMA engines, which moved 16 GB/s here (255 MB of
// counts in 16 ms) where this kernel is limited by the link.
typedef uint32_t copy_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_out_kernel(copy_u32x4* __restrict__ dst, const copy_u32x4* __restrict__ src, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) __builtin_nontemporal_store(src[i], &dst[i]);
}

int32_t copy_to_host(midas_snps_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t on = nullptr) {
  if (bytes == 0) return MIDAS_SNPS_OK;
  hipStream_t s = on ? on : ctx->stream;
  hipPointerAttribute_t at;
  const bool pinned = hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeHost;
  (void)hipGetLastError();   // (an unregistered pointer is reported as an error: that is the pageable case)
#ifndef MIDAS_SNPS_NO_COPY_KERNEL
  if (pinned && bytes >= ((size_t)1 << 20) && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
    void* mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, dst, 0) == hipSuccess && mapped) {
      const size_t n16 = bytes / 16, tail = bytes - n16 * 16;
      hipLaunchKernelGGL(copy_out_kernel, dim3(2048), dim3(256), 0, s, static_cast<copy_u32x4*>(mapped), static_cast<const copy_u32x4*>(src), n16);
      HIP_TRY(ctx, hipGetLastError());
      if (tail) HIP_TRY(ctx, hipMemcpyAsync(static_cast<uint8_t*>(dst) + n16 * 16, static_cast<const uint8_t*>(src) + n16 * 16, tail, hipMemcpyDeviceToHost, s));
      HIP_TRY(ctx, hipStreamSynchronize(s));
      return MIDAS_SNPS_OK;
    }
    (void)hipGetLastError();
  }
#endif
  if (pinned || bytes < ((size_t)1 << 20)) {
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MIDAS_SNPS_OK;
  }
  constexpr size_t kChunk = midas_snps_ctx::kStageBytes;
  std::lock_guard<std::mutex> ring(ctx->copy_mutex);
  ctx->stage_join();
  for (int k = 0; k < midas_snps_ctx::kStageSlots; ++k) {
    if (!ctx->stage[k]) HIP_TRY(ctx, hipHostMalloc(&ctx->stage[k], kChunk, kHostAllocFlags));
    if (!ctx->stage_ev[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
  }
  const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
  auto issue = [&](size_t k) -> hipError_t {
    const size_t off = k * kChunk, n = std::min(kChunk, bytes - off);
    hipError_t e = hipSuccess;
    void* mapped = nullptr;
#ifndef MIDAS_SNPS_NO_COPY_KERNEL
    if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0 && hipHostGetDevicePointer(&mapped, ctx->stage[k & 1], 0) != hipSuccess) mapped = nullptr;
    (void)hipGetLastError();
#endif
    if (mapped && n >= 16) {
      const size_t n16 = n / 16, tail = n - n16 * 16;
      hipLaunchKernelGGL(copy_out_kernel, dim3(2048), dim3(256), 0, s, static_cast<copy_u32x4*>(mapped),
                         reinterpret_cast<const copy_u32x4*>(static_cast<const uint8_t*>(src) + off), n16);
      e = hipGetLastError();
      if (e == hipSuccess && tail)
        e = hipMemcpyAsync(static_cast<uint8_t*>(ctx->stage[k & 1]) + n16 * 16, static_cast<const uint8_t*>(src) + off + n16 * 16, tail, hipMemcpyDeviceToHost, s);
    } else {
      e = hipMemcpyAsync(ctx->stage[k & 1], static_cast<const uint8_t*>(src) + off, n, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipEventRecord(ctx->stage_ev[k & 1], s);
    return e;
  };
  HIP_TRY(ctx, issue(0));
  for (size_t k = 0; k < n_chunks; ++k) {
    if (k + 1 < n_chunks) HIP_TRY(ctx, issue(k + 1));      // slot (k + 1) & 1 was emptied in the previous round
    HIP_TRY(ctx, hipEventSynchronize(ctx->stage_ev[k & 1]));
    const size_t off = k * kChunk, n = std::min(kChunk, bytes - off);
    parallel_copy(static_cast<uint8_t*>(dst) + off, static_cast<const uint8_t*>(ctx->stage[k & 1]), n);
  }
  return MIDAS_SNPS_OK;
}

// Host bytes that are NOT page-locked (a mapped file, a malloc'd buffer) to the device through the context's pinned ring:
// several threads copy a chunk into a slot while the slot before it crosses the link.  The runtime's own pageable path
// stages through one thread: 14 GB/s out of a file mapping where this reaches the threads' copy rate.
// n bytes of a file into dst (a slot of the pinned ring) by several threads: the page cache's bytes, without a page of the
// file mapped for them (hostio.h, register_file_mapping)
bool parallel_pread(uint8_t* dst, int fd, size_t file_off, size_t n) {
  const unsigned hw = (unsigned)midas::cpu_budget();
  size_t nt = hw >= 32 ? 12 : (hw >= 16 ? 8 : (hw >= 8 ? 4 : (hw >= 2 ? 2 : 1)));
  static const int forced = [] { const char* e = getenv("MIDAS_SNPS_UPLOAD_THREADS"); return e ? atoi(e) : 0; }();
  if (forced > 0) nt = (size_t)std::min(forced, 64);
  if (n < ((size_t)4 << 20)) nt = 1;
  const size_t per = ((n + nt - 1) / nt + 4095) & ~(size_t)4095;
  std::atomic<int> bad{0};
  auto piece = [&](size_t k) {
    size_t off = k * per;
    const size_t end = std::min(n, off + per);
    while (off < end) {
      const ssize_t got = pread(fd, dst + off, end - off, (off_t)(file_off + off));
      if (got <= 0) { bad = 1; return; }
      off += (size_t)got;
    }
  };
  if (nt == 1) { piece(0); return bad == 0; }
  std::atomic<size_t> next{0};
  midas::Workers::run((int)nt, [&] {
    for (;;) {
      const size_t k = next.fetch_add(1);
      if (k >= nt) return;
      piece(k);
    }
  });
  return bad == 0;
}

int32_t copy_to_device_staged(midas_snps_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s) {
  constexpr size_t kChunk = midas_snps_ctx::kStageBytes;
  int src_fd = -1;
  size_t src_off = 0;
  const bool from_file = midas::file_of_mapping(src, bytes, &src_fd, &src_off);      // (a mapped BAM: read with pread, not through the mapping)
  if (bytes < 2 * kChunk && !from_file) {
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    return MIDAS_SNPS_OK;
  }
  std::lock_guard<std::mutex> ring(ctx->copy_mutex);
  ctx->stage_join();
  for (int k = 0; k < midas_snps_ctx::kStageSlots; ++k) {
    if (!ctx->stage[k]) HIP_TRY(ctx, hipHostMalloc(&ctx->stage[k], kChunk, kHostAllocFlags));
    if (!ctx->stage_ev[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
  }
  const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
  for (size_t k = 0; k < n_chunks; ++k) {
    const int slot = (int)(k % midas_snps_ctx::kStageSlots);
    if (k >= (size_t)midas_snps_ctx::kStageSlots) HIP_TRY(ctx, hipEventSynchronize(ctx->stage_ev[slot]));      // the slot's last chunk is over
    const size_t off = k * kChunk, n = std::min(kChunk, bytes - off);
    if (from_file) {
      if (!parallel_pread(static_cast<uint8_t*>(ctx->stage[slot]), src_fd, src_off + off, n)) return fail(ctx, MIDAS_SNPS_ERR_BAD_LAYOUT, "short read on the BAM file");
    } else {
      parallel_copy(static_cast<uint8_t*>(ctx->stage[slot]), static_cast<const uint8_t*>(src) + off, n);
    }
    HIP_TRY(ctx, hipMemcpyAsync(static_cast<uint8_t*>(dst) + off, ctx->stage[slot], n, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipEventRecord(ctx->stage_ev[slot], s));
  }
  // (the ring is free again when the caller next waits for the stream; the D2H path waits on the same events before it reuses a slot)
  for (int k = 0; k < midas_snps_ctx::kStageSlots; ++k) HIP_TRY(ctx, hipEventSynchronize(ctx->stage_ev[k]));
  return MIDAS_SNPS_OK;
}

// tile ranges are double-buffered by run parity: [rbinv0][rend0][rbinv1][rend1]
// (each tile has three ranges, slots 3t..3t+2: see index_reads.hip)
uint32_t* work_rbinv(midas_snps_batch* b, int par) { return reinterpret_cast<uint32_t*>(b->d_work) + (size_t)par * 6 * b->n_tiles; }
uint32_t* work_rend(midas_snps_batch* b, int par) { return work_rbinv(b, par) + 3 * b->n_tiles; }
unsigned long long* work_stats(midas_snps_batch* b) {
  size_t off = ((size_t)b->n_tiles * 48 + 15) & ~(size_t)15;
  return reinterpret_cast<unsigned long long*>(b->d_work + off);
}
unsigned long long* work_err(midas_snps_batch* b) { return work_stats(b) + (size_t)b->n_species * MIDAS_STATS; }

const char* read_err_name(int32_t st) {
  switch (st) {
    case MIDAS_SNPS_ERR_READ_NO_SEQ: return "record has no SEQ (reference: TypeError in keep_read)";
    case MIDAS_SNPS_ERR_READ_NO_NM: return "record has no NM tag (reference: KeyError 'NM' in keep_read)";
    case MIDAS_SNPS_ERR_READ_ZERO_ALIGN: return "aligned length is 0 (reference: ZeroDivisionError in keep_read)";
    case MIDAS_SNPS_ERR_READ_NO_QUAL: return "record has no QUAL (reference: TypeError in np.mean)";
    case MIDAS_SNPS_ERR_READ_CIGAR_OVERRUN: return "CIGAR consumes more query than SEQ holds (reference: IndexError)";
    case MIDAS_SNPS_ERR_READ_BAD_CIGAR_OP: return "bad CIGAR op";
    default: return "unknown";
  }
}

}  // namespace

extern "C" {

int32_t midas_snps_abi_version(void) { return MIDAS_SNPS_ABI_VERSION; }

int32_t midas_snps_cpu_budget(void) { return (int32_t)midas::cpu_budget(); }

const char* midas_snps_status_string(int32_t st) {
  switch (st) {
    case MIDAS_SNPS_OK: return "ok";
    case MIDAS_SNPS_ERR_INVALID_ARG: return "invalid argument";
    case MIDAS_SNPS_ERR_NO_DEVICE: return "no usable gfx950 device";
    case MIDAS_SNPS_ERR_HIP: return "HIP runtime error";
    case MIDAS_SNPS_ERR_OUT_OF_MEMORY: return "out of device memory";
    case MIDAS_SNPS_ERR_UNSUPPORTED: return "unsupported input";
    case MIDAS_SNPS_ERR_BAD_LAYOUT: return "inconsistent input layout";
    default: return (st >= 1 && st <= 6) ? read_err_name(st) : "unknown status";
  }
}

int32_t midas_snps_create(int32_t device_ordinal, midas_snps_ctx** out_ctx) {
  if (!out_ctx) return MIDAS_SNPS_ERR_INVALID_ARG;
  *out_ctx = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return MIDAS_SNPS_ERR_NO_DEVICE;
  }
  if (device_ordinal < 0 || device_ordinal >= n) return MIDAS_SNPS_ERR_INVALID_ARG;
  midas_snps_ctx* ctx = new (std::nothrow) midas_snps_ctx();
  if (!ctx) return MIDAS_SNPS_ERR_OUT_OF_MEMORY;
  ctx->device = device_ordinal;
  if (hipSetDevice(device_ordinal) != hipSuccess ||
      hipGetDeviceProperties(&ctx->prop, device_ordinal) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    delete ctx;
    return MIDAS_SNPS_ERR_NO_DEVICE;
  }
  // The kernels are built for gfx950 only; anything else cannot run them.
  if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return MIDAS_SNPS_ERR_NO_DEVICE;
  }
  ctx->stream = ctx->own_stream;
  const char* coder = getenv("MIDAS_SNPS_ROW_CODER");
  ctx->row_coder = (coder && strcmp(coder, "host") == 0) ? MIDAS_SNPS_ROWS_HOST : MIDAS_SNPS_ROWS_DEVICE;
  // the pinned staging ring, page-locked beside whatever the caller does next (ctx_internal.h: stage_thread)
  ctx->stage_thread = std::thread([ctx] {
    if (hipSetDevice(ctx->device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (int k = 0; k < midas_snps_ctx::kStageSlots; ++k)
      if (hipHostMalloc(&ctx->stage[k], midas_snps_ctx::kStageBytes, kHostAllocFlags) != hipSuccess) { (void)hipGetLastError(); ctx->stage[k] = nullptr; }
  });
  *out_ctx = ctx;
  return MIDAS_SNPS_OK;
}

void midas_snps_destroy(midas_snps_ctx* ctx) {
  if (!ctx) return;
  ctx->stage_join();
  (void)hipSetDevice(ctx->device);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  ctx->row_buffers.clear();
  if (ctx->arena) ctx->arena->close();
  for (int k = 0; k < midas_snps_ctx::kStageSlots; ++k) {
    if (ctx->stage[k]) (void)hipHostFree(ctx->stage[k]);
    if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
  }
  delete ctx;
}

void* midas_snps_host_alloc(int64_t bytes) {
  void* p = nullptr;
  if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, kHostAllocFlags) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void midas_snps_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

// The text is copied out under the context's error lock into a buffer of the calling thread (valid until that thread asks again):
// the table writers of one context run on several host threads.
const char* midas_snps_last_error(const midas_snps_ctx* ctx) {
  if (!ctx) return "NULL context";
  static thread_local std::string text;
  text = const_cast<midas_snps_ctx*>(ctx)->error_text();
  return text.c_str();
}

int64_t midas_snps_last_error_read(const midas_snps_ctx* ctx) { return ctx ? ctx->err_read : -1; }

int32_t midas_snps_set_stream(midas_snps_ctx* ctx, void* hip_stream) {
  if (!ctx) return MIDAS_SNPS_ERR_INVALID_ARG;
  ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return MIDAS_SNPS_OK;
}

// Device-to-device copy rate of THIS device, with the library's own copy kernel (16 bytes per lane, streaming stores): the
// practical HBM ceiling a roofline fraction can be held against (the guide's 6.29 TB/s figure comes from such a kernel).
int32_t midas_snps_copy_rate(midas_snps_ctx* ctx, int64_t bytes, int32_t reps, double* out_gbps) {
  if (!ctx || !out_gbps || bytes < (1 << 20) || reps < 1) return MIDAS_SNPS_ERR_INVALID_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n16 = (size_t)bytes / 16;
  copy_u32x4 *src = nullptr, *dst = nullptr;
  HIP_TRY(ctx, hipMalloc(&src, n16 * 16));
  if (hipMalloc(&dst, n16 * 16) != hipSuccess) { (void)hipFree(src); return fail(ctx, MIDAS_SNPS_ERR_OUT_OF_MEMORY, "copy_rate: out of device memory"); }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t s = ctx->stream;
  int32_t st = MIDAS_SNPS_OK;
  float ms = 0.f;
  const int grid = ctx->prop.multiProcessorCount * 16;
  if (hipMemsetAsync(src, 0x5A, n16 * 16, s) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) st = MIDAS_SNPS_ERR_HIP;
  if (st == MIDAS_SNPS_OK) {
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(copy_out_kernel, dim3(grid), dim3(256), 0, s, dst, src, n16);
    (void)hipEventRecord(e0, s);
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(copy_out_kernel, dim3(grid), dim3(256), 0, s, dst, src, n16);
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) st = MIDAS_SNPS_ERR_HIP;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(src);
  (void)hipFree(dst);
  if (st != MIDAS_SNPS_OK) { (void)hipGetLastError(); return fail(ctx, st, "copy_rate: HIP runtime error"); }
  *out_gbps = 2.0 * (double)(n16 * 16) * reps / ((double)ms * 1e-3) / 1e9;
  return MIDAS_SNPS_OK;
}

namespace {
// midas::BlockInflater over a context: the streams go to the device, one thread inflates each (bgzf_inflate.hip), the
// inflated bytes come back through the staging ring.
struct InflateUser {
  midas_snps_ctx* ctx;
  bool keep = false;            // leave the inflated stream on the device: `kept` (the caller frees it)
  void* kept = nullptr;         // the arena; the inflated stream is at its start
  size_t kept_bytes = 0;
  uint8_t* scratch = nullptr;   // what lies behind the stream in the arena: dead once the call returns, the caller's to reuse
  size_t scratch_bytes = 0;
};
int32_t device_inflate(void* user, const InflateSegment* segs, size_t n_segs, const InflateJob* jobs, size_t n_jobs, uint8_t* out,
                       size_t out_bytes, int64_t* bad_job, char* err256) {
  InflateUser* iu = static_cast<InflateUser*>(user);
  midas_snps_ctx* ctx = iu->ctx;
  if (bad_job) *bad_job = -1;
  if (n_jobs == 0) return MIDAS_SNPS_OK;
  size_t comp_bytes = 0;
  for (size_t k = 0; k < n_segs; ++k) comp_bytes += segs[k].n;
  for (size_t k = 0; k < n_jobs; ++k) {
    if (jobs[k].cpos + jobs[k].clen > comp_bytes || jobs[k].upos + jobs[k].ulen > out_bytes) {
      if (err256) snprintf(err256, 256, "device inflate: stream %lld lies outside the buffers", (long long)k);
      return MIDAS_SNPS_ERR_INVALID_ARG;
    }
  }
  auto hip_err = [&](hipError_t e, const char* what) {
    if (err256) snprintf(err256, 256, "device inflate: %s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? MIDAS_SNPS_ERR_OUT_OF_MEMORY : MIDAS_SNPS_ERR_HIP;
  };
#define INF_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_err(e__, #call); } while (0)
  std::lock_guard<std::mutex> g(ctx->device_mutex);
  const bool trace = getenv("MIDAS_SNPS_TRACE") != nullptr;       // where the call spends its time, on stderr
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[device inflate] %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  INF_TRY(hipSetDevice(ctx->device));
  // ONE allocation for everything the call needs (a hipMalloc / hipFree pair costs tens of milliseconds per gigabyte here, and
  // the match lists' worst-case room alone is 2.7 bytes per output byte): | inflated | compressed | blocks | status | matches |
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  // Room for a stream's tokens and literals (bgzf_inflate.hip: a dword per match from the front, the literals from the end): as
  // many bytes as the stream inflates to -- a BAM's block needs ~0.7 of that (8 000 tokens + 13 000 literals for 64 KiB), the
  // bound is 4/3 (a match per three bytes) -- because device memory costs ~17 ms a gigabyte to allocate here and this is the
  // second largest buffer.  A stream that needs more says so (kInflateMatchRoom) and is decoded again, with the bound's room,
  // in a second small launch.  (InflateBlock counts the room in 8-byte units.)
  std::vector<InflateBlock> blocks(n_jobs);
  unsigned long long n_match_room = 0;
  for (size_t k = 0; k < n_jobs; ++k) {
    const uint32_t cap = jobs[k].ulen / 8u + 16u;
    blocks[k] = InflateBlock{jobs[k].cpos, jobs[k].upos, n_match_room, jobs[k].clen, jobs[k].ulen, cap, 0u};
    n_match_room += cap;
  }
  bool check = n_jobs > 0;       // the streams' CRC-32 is verified when every one of them brings it (BGZF blocks do)
  for (size_t k = 0; k < n_jobs; ++k) check = check && jobs[k].check_crc != 0u;
  const size_t at_out = 0, at_comp = up(out_bytes + 64), at_blocks = at_comp + up(comp_bytes + 512),
               at_status = at_blocks + up(n_jobs * sizeof(InflateBlock)), at_crc = at_status + up(n_jobs * 8),
               at_matches = at_crc + up(n_jobs * 4), arena_bytes = at_matches + up((size_t)n_match_room * 8);
  struct Buf { void* p = nullptr; ~Buf() { if (p) (void)hipFree(p); } } arena;
  INF_TRY(hipMalloc(&arena.p, arena_bytes));
  uint8_t* const base = static_cast<uint8_t*>(arena.p);
  struct View { void* p; } d_out{base + at_out}, d_comp{base + at_comp}, d_blocks{base + at_blocks}, d_status{base + at_status},
      d_crc{base + at_crc}, d_matches{base + at_matches};
  hipStream_t s = ctx->stream;
  size_t at = 0;
  for (size_t k = 0; k < n_segs; ++k) {
    if (segs[k].n) INF_TRY(hipMemcpyAsync(static_cast<uint8_t*>(d_comp.p) + at, segs[k].p, segs[k].n, hipMemcpyHostToDevice, s));
    at += segs[k].n;
  }
  INF_TRY(hipMemsetAsync(static_cast<uint8_t*>(d_comp.p) + comp_bytes, 0, 512, s));
  if (trace) { INF_TRY(hipStreamSynchronize(s)); lap("hipMalloc + streams up"); }
  INF_TRY(hipMemcpyAsync(d_blocks.p, blocks.data(), n_jobs * sizeof(InflateBlock), hipMemcpyHostToDevice, s));
  InflateParams ip;
  ip.comp = static_cast<const uint8_t*>(d_comp.p);
  ip.blocks = static_cast<const InflateBlock*>(d_blocks.p);
  ip.n_blocks = (long long)n_jobs;
  ip.out = static_cast<uint8_t*>(d_out.p);
  ip.status = static_cast<uint32_t*>(d_status.p);
  ip.n_matches = static_cast<uint32_t*>(d_status.p) + n_jobs;
  ip.matches = static_cast<unsigned long long*>(d_matches.p);
  ip.want_crc = nullptr;
  std::vector<uint32_t> want;
  if (check) {
    want.resize(n_jobs);
    for (size_t k = 0; k < n_jobs; ++k) want[k] = jobs[k].crc;
    INF_TRY(hipMemcpyAsync(d_crc.p, want.data(), n_jobs * 4, hipMemcpyHostToDevice, s));
    ip.want_crc = static_cast<const uint32_t*>(d_crc.p);
  }
  if (trace) {
    INF_TRY(launch_bgzf_inflate(ip, s, 1));
    INF_TRY(hipStreamSynchronize(s));
    lap("decode kernel");
    INF_TRY(launch_bgzf_inflate(ip, s, 2));
    INF_TRY(hipStreamSynchronize(s));
    lap("resolve kernel");
  } else {
    INF_TRY(launch_bgzf_inflate(ip, s));
  }
  std::vector<uint32_t> status(n_jobs);
  INF_TRY(hipMemcpyAsync(status.data(), d_status.p, n_jobs * 4, hipMemcpyDeviceToHost, s));
  INF_TRY(hipStreamSynchronize(s));
  lap("kernel");
  {   // the streams whose matches did not fit: again, with the bound's room
    std::vector<size_t> again;
    for (size_t k = 0; k < n_jobs; ++k)
      if (status[k] == kInflateMatchRoom) again.push_back(k);
    if (!again.empty()) {
      std::vector<InflateBlock> b2(again.size());
      unsigned long long room2 = 0;
      for (size_t j = 0; j < again.size(); ++j) {
        const InflateJob& q = jobs[again[j]];
        const uint32_t cap = q.ulen / 3u + 1u;
        b2[j] = InflateBlock{q.cpos, q.upos, room2, q.clen, q.ulen, cap, 0u};
        room2 += cap;
      }
      Buf d_b2, d_s2, d_m2, d_c2;
      std::vector<uint32_t> want2(again.size());
      for (size_t j = 0; j < again.size(); ++j) want2[j] = jobs[again[j]].crc;
      if (check) {
        INF_TRY(hipMalloc(&d_c2.p, again.size() * 4));
        INF_TRY(hipMemcpyAsync(d_c2.p, want2.data(), again.size() * 4, hipMemcpyHostToDevice, s));
      }
      INF_TRY(hipMalloc(&d_b2.p, b2.size() * sizeof(InflateBlock)));
      INF_TRY(hipMalloc(&d_s2.p, b2.size() * 8));
      INF_TRY(hipMalloc(&d_m2.p, (size_t)room2 * 8));
      INF_TRY(hipMemcpyAsync(d_b2.p, b2.data(), b2.size() * sizeof(InflateBlock), hipMemcpyHostToDevice, s));
      InflateParams ip2 = ip;
      ip2.blocks = static_cast<const InflateBlock*>(d_b2.p);
      ip2.n_blocks = (long long)b2.size();
      ip2.status = static_cast<uint32_t*>(d_s2.p);
      ip2.n_matches = static_cast<uint32_t*>(d_s2.p) + b2.size();
      ip2.matches = static_cast<unsigned long long*>(d_m2.p);
      ip2.want_crc = check ? static_cast<const uint32_t*>(d_c2.p) : nullptr;
      INF_TRY(launch_bgzf_inflate(ip2, s));
      std::vector<uint32_t> st2(b2.size());
      INF_TRY(hipMemcpyAsync(st2.data(), d_s2.p, b2.size() * 4, hipMemcpyDeviceToHost, s));
      INF_TRY(hipStreamSynchronize(s));
      for (size_t j = 0; j < again.size(); ++j) status[again[j]] = st2[j];
      lap("streams decoded again");
    }
  }
#undef INF_TRY
  for (size_t k = 0; k < n_jobs; ++k) {
    if (status[k] != 0u) {
      if (bad_job) *bad_job = (int64_t)k;
      if (err256) snprintf(err256, 256, status[k] == kInflateCrc ? "CRC-32 mismatch (stream %lld: the inflated bytes are not the ones that were compressed)"
                                                                 : "corrupt deflate data (stream %lld: code %u)", (long long)k, status[k]);
      return MIDAS_SNPS_ERR_BAD_LAYOUT;
    }
  }
  const int32_t st = copy_to_host(ctx, out, d_out.p, out_bytes);
  if (st != MIDAS_SNPS_OK && err256) snprintf(err256, 256, "device inflate: results to host: %s", ctx->error_text().c_str());
  lap("inflated bytes down");
  if (st == MIDAS_SNPS_OK && iu->keep) {     // the caller takes the arena: the inflated stream, and everything behind it as scratch
    iu->kept = arena.p; iu->kept_bytes = out_bytes; iu->scratch = base + at_comp; iu->scratch_bytes = arena_bytes - at_comp;
    arena.p = nullptr;
  }
  return st;
}
}  // namespace

int32_t midas_snps_inflate_blocks(midas_snps_ctx* ctx, const uint8_t* comp, int64_t comp_bytes, int64_t n_blocks,
                                  const int64_t* cpos, const int32_t* clen, const int64_t* upos, const int32_t* ulen,
                                  const uint32_t* crc, uint8_t* out, int64_t out_bytes, int64_t* bad_block) {
  if (!ctx || comp_bytes < 0 || n_blocks < 0 || out_bytes < 0 || (n_blocks > 0 && (!comp || !cpos || !clen || !upos || !ulen || !out)))
    return MIDAS_SNPS_ERR_INVALID_ARG;
  std::vector<InflateJob> jobs((size_t)n_blocks);
  for (int64_t k = 0; k < n_blocks; ++k) {
    if (cpos[k] < 0 || clen[k] < 0 || upos[k] < 0 || ulen[k] < 0) return fail(ctx, MIDAS_SNPS_ERR_INVALID_ARG, "inflate_blocks: negative offset or size");
    jobs[(size_t)k] = InflateJob{(uint64_t)cpos[k], (uint64_t)upos[k], (uint32_t)clen[k], (uint32_t)ulen[k], crc ? crc[k] : 0u, crc ? 1u : 0u};
  }
  const InflateSegment seg{comp, (size_t)comp_bytes};
  char err[256] = {0};
  InflateUser iu{ctx};
  const int32_t st = device_inflate(&iu, &seg, 1, jobs.data(), jobs.size(), out, (size_t)out_bytes, bad_block, err);
  if (st != MIDAS_SNPS_OK) return fail(ctx, st, err);
  return MIDAS_SNPS_OK;
}

int32_t midas_bam_open_device(const char* path, midas_snps_ctx* ctx, midas_bam** out, char* err256) {
  if (!ctx) return MIDAS_SNPS_ERR_INVALID_ARG;
  InflateUser iu{ctx};
  const BlockInflater inf{&iu, device_inflate};
  return bam_open_with(path, &inf, out, err256);
}

namespace {
void device_free(void* p) { (void)hipFree(p); }
}

// The decode with the host walking the records (the inflated stream comes down for that): what midas_bam_load_device falls back
// to when the device cannot settle the record boundaries.
static int32_t bam_load_device_host_walk(const char* path, midas_snps_ctx* ctx, midas_bam** out, int64_t* n_reads, int64_t* seq_bytes,
                                         int64_t* qual_bytes, int64_t* n_cigar, char* err256) {
  if (!ctx || !path || !out) return MIDAS_SNPS_ERR_INVALID_ARG;
  *out = nullptr;
  InflateUser iu{ctx};
  iu.keep = true;
  const bool trace = getenv("MIDAS_SNPS_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[device decode] %-26s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  struct Kept { InflateUser* u; ~Kept() { if (u->kept) (void)hipFree(u->kept); } } kept{&iu};       // (freed on every way out)
  const BlockInflater inf{&iu, device_inflate};
  midas_bam* b = nullptr;
  int32_t st = bam_open_with(path, &inf, &b, err256);
  if (st != MIDAS_SNPS_OK) return st;
  struct Handle { midas_bam* b; ~Handle() { if (b) midas_bam_close(b); } } handle{b};
  lap("open (map, inflate, header)");
  bam_keep_payload_on_device(b);
  int64_t n = 0, sb = 0, qb = 0, nc = 0;
  st = midas_bam_load(b, &n, &sb, &qb, &nc, err256);        // the host walks the records and decodes the small columns
  if (st != MIDAS_SNPS_OK) return st;
  lap("host walk + small columns");
  size_t n_off = 0;
  const uint64_t* rec_off = bam_record_offsets(b, &n_off);
  const int64_t *seq_off, *qual_off, *cigar_off;
  bam_offsets(b, &seq_off, &qual_off, &cigar_off);
  auto hip_err = [&](hipError_t e, const char* what) {
    if (err256) snprintf(err256, 256, "device decode: %s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? MIDAS_SNPS_ERR_OUT_OF_MEMORY : MIDAS_SNPS_ERR_HIP;
  };
#define DEC_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_err(e__, #call); } while (0)
  {
    std::lock_guard<std::mutex> g(ctx->device_mutex);
    DEC_TRY(hipSetDevice(ctx->device));
    // the columns and the offsets the cut needs go where the compressed bytes and the match lists were (the arena's scratch is
    // 2.7 x the stream, the columns 0.9 x): no second allocation
    struct View { void* p = nullptr; } d_rec, d_so, d_qo, d_co, d_seq, d_qual, d_cig;
    const size_t n1 = (size_t)n + 1;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t at = 0;
    auto take = [&](View& v, size_t bytes) { v.p = iu.scratch + at; at += up(bytes); };
    take(d_seq, (size_t)sb + 64); take(d_qual, (size_t)qb + 64); take(d_cig, (size_t)nc * 4 + 64);
    take(d_rec, n1 * 8); take(d_so, n1 * 8); take(d_qo, n1 * 8); take(d_co, n1 * 8);
    struct Own { void* p = nullptr; ~Own() { if (p) (void)hipFree(p); } } own;
    if (at > iu.scratch_bytes) {      // (very short reads: more offsets than the scratch has room for -- a buffer of their own)
      DEC_TRY(hipMalloc(&own.p, at));
      const ptrdiff_t shift = static_cast<uint8_t*>(own.p) - iu.scratch;
      for (View* v : {&d_seq, &d_qual, &d_cig, &d_rec, &d_so, &d_qo, &d_co}) v->p = static_cast<uint8_t*>(v->p) + shift;
    }
    hipStream_t s = ctx->stream;
    lap("hipMalloc of the columns");
    if (n > 0) DEC_TRY(hipMemcpyAsync(d_rec.p, rec_off, (size_t)n * 8, hipMemcpyHostToDevice, s));
    DEC_TRY(hipMemcpyAsync(d_so.p, seq_off, n1 * 8, hipMemcpyHostToDevice, s));
    DEC_TRY(hipMemcpyAsync(d_qo.p, qual_off, n1 * 8, hipMemcpyHostToDevice, s));
    DEC_TRY(hipMemcpyAsync(d_co.p, cigar_off, n1 * 8, hipMemcpyHostToDevice, s));
    DEC_TRY(hipMemsetAsync(static_cast<uint8_t*>(d_cig.p) + (size_t)nc * 4, 0, 64, s));
    if (trace) { DEC_TRY(hipStreamSynchronize(s)); lap("offsets up"); }
    PayloadParams pp;
    pp.stream = static_cast<const uint8_t*>(iu.kept);
    pp.rec_off = static_cast<const unsigned long long*>(d_rec.p);
    pp.n_records = n;
    pp.seq_off = static_cast<const long long*>(d_so.p);
    pp.qual_off = static_cast<const long long*>(d_qo.p);
    pp.cigar_off = static_cast<const long long*>(d_co.p);
    pp.seq4 = static_cast<uint8_t*>(d_seq.p);
    pp.qual = static_cast<uint8_t*>(d_qual.p);
    pp.cigar = static_cast<uint32_t*>(d_cig.p);
    DEC_TRY(launch_bam_payload(pp, ctx->prop.multiProcessorCount, s));
    DEC_TRY(hipStreamSynchronize(s));
    lap("payload kernel");
    if (own.p) {                      // (the columns have their own buffer: the arena goes now)
      bam_set_device_payload(b, d_seq.p, d_qual.p, d_cig.p, own.p, device_free);
      own.p = nullptr;
    } else {
      bam_set_device_payload(b, d_seq.p, d_qual.p, d_cig.p, iu.kept, device_free);     // (the arena lives as long as the columns)
      iu.kept = nullptr;
    }
  }
  lap("free scratch");
#undef DEC_TRY
  if (n_reads) *n_reads = n;
  if (seq_bytes) *seq_bytes = sb;
  if (qual_bytes) *qual_bytes = qb;
  if (n_cigar) *n_cigar = nc;
  *out = b;
  handle.b = nullptr;
  return MIDAS_SNPS_OK;
}


namespace {
struct ArenaLoan { std::shared_ptr<midas_arena_pool> pool; void* p; };
void arena_loan_free(void* v) {
  ArenaLoan* l = static_cast<ArenaLoan*>(v);
  if (l) { l->pool->drop_twins(l->p); l->pool->give(l->p); delete l; }
}

// The streams of a decode whose matches did not fit their room (status kInflateMatchRoom): again, with the bound's room (a match
// is at least three bytes), into the same output; their statuses replace the first pass's.
hipError_t inflate_again(const InflateParams& ip, const std::vector<InflateBlock>& blocks, const std::vector<uint32_t>& want, std::vector<uint32_t>& status,
                         hipStream_t s, bool* ran) {
  *ran = false;
  std::vector<size_t> again;
  for (size_t k = 0; k < status.size(); ++k)
    if (status[k] == kInflateMatchRoom) again.push_back(k);
  if (again.empty()) return hipSuccess;
  *ran = true;
  std::vector<InflateBlock> b2(again.size());
  std::vector<uint32_t> want2(again.size());
  unsigned long long room2 = 0;
  for (size_t j = 0; j < again.size(); ++j) {
    const InflateBlock& q = blocks[again[j]];
    const uint32_t cap = q.ulen / 3u + 1u;
    b2[j] = InflateBlock{q.cpos, q.upos, room2, q.clen, q.ulen, cap, 0u};
    want2[j] = want[again[j]];
    room2 += cap;
  }
  struct Buf { void* p = nullptr; ~Buf() { if (p) (void)hipFree(p); } } d_b2, d_s2, d_m2, d_c2;
  hipError_t e;
#define AG_TRY(call) do { e = (call); if (e != hipSuccess) return e; } while (0)
  AG_TRY(hipMalloc(&d_b2.p, b2.size() * sizeof(InflateBlock)));
  AG_TRY(hipMalloc(&d_s2.p, b2.size() * 8));
  AG_TRY(hipMalloc(&d_m2.p, (size_t)room2 * 8));
  AG_TRY(hipMalloc(&d_c2.p, b2.size() * 4));
  AG_TRY(hipMemcpyAsync(d_b2.p, b2.data(), b2.size() * sizeof(InflateBlock), hipMemcpyHostToDevice, s));
  AG_TRY(hipMemcpyAsync(d_c2.p, want2.data(), b2.size() * 4, hipMemcpyHostToDevice, s));
  InflateParams ip2 = ip;
  ip2.blocks = static_cast<const InflateBlock*>(d_b2.p);
  ip2.n_blocks = (long long)b2.size();
  ip2.status = static_cast<uint32_t*>(d_s2.p);
  ip2.n_matches = static_cast<uint32_t*>(d_s2.p) + b2.size();
  ip2.matches = static_cast<unsigned long long*>(d_m2.p);
  ip2.want_crc = static_cast<const uint32_t*>(d_c2.p);
  AG_TRY(launch_bgzf_inflate(ip2, s));
  std::vector<uint32_t> st2(b2.size());
  AG_TRY(hipMemcpyAsync(st2.data(), d_s2.p, b2.size() * 4, hipMemcpyDeviceToHost, s));
  AG_TRY(hipStreamSynchronize(s));
#undef AG_TRY
  for (size_t j = 0; j < again.size(); ++j) status[again[j]] = st2[j];
  return hipSuccess;
}

// ---- the streamed decode ------------------------------------------------------------------------------------------------------
// A BAM of several device-fills of blocks (bgzf_inflate_wave_blocks), decoded RESIDENT group by group: the reference's loop over the
// file (midas/run/snps.py:186-199 iterates the alignments as htslib inflates them, a block at a time) at the device's granularity.
//   * a GROUP is a run of whole BGZF blocks -- one wave of the decoder's workgroups by default -- plus a few blocks behind it for
//     the record that straddles its end; it wants the records that START inside it.  Where the chain of group g ends (the first
//     record start at or behind its last wanted byte) is the exact first record of group g + 1: nothing is guessed behind group 0.
//   * a group lives in a SLOT (inflated bytes | compressed bytes | block tables | match lists, the walk's tables over the dead
//     ones); an uploader thread fills slot (g + 1) % S through the pinned ring on a stream of its own while the kernels of group g
//     run on the context's -- the link and the decoder work at the same time.
//   * what STAYS is written where it stays: every group's columns continue the ones before it (BamColumnsParams::base: the offset
//     scans start at what the earlier groups came to), its records and their [cigar][seq][qual] runs go straight behind theirs
//     in the direct layout.  Those arrays are sized from the first group's records per inflated byte (+ 3 %) and grown (a copy on
//     the device) if a later group proves the estimate short.
// Device memory: S slots of ~2.5 x a group's inflated bytes + the result (~1.1 x the file's inflated bytes), against ~2.3 x the
// file's inflated bytes in one arena -- bounded by the group, not by the file, in everything but the result itself.
// No inflated stream is kept: a handle decoded this way cuts its raw columns, if somebody asks for them, out of the direct layout
// (PayloadParams::drec).  kStreamFallback: this BAM is not for the streamed decode (a record longer than the blocks a group
// keeps behind its end) -- the caller decodes it in one arena.
constexpr int32_t kStreamFallback = -1000;
struct TwoBuffers { void* a; void* b; };
void two_buffers_free(void* v) {
  TwoBuffers* t = static_cast<TwoBuffers*>(v);
  if (t->a) (void)hipFree(t->a);
  if (t->b) (void)hipFree(t->b);
  delete t;
}
struct StreamColumns {       // the result's record arrays in ONE allocation, for `cap` records (+ 2: the offsets' last entry, the sentinel record)
  uint8_t* p = nullptr;
  size_t cap = 0, bytes = 0;
  size_t at[11] = {0};        // rec, refid, pos, nm, l_seq, mapq, flag, seq_off, qual_off, cigar_off, unit_off
  static constexpr size_t width(int k) { return k == 0 ? 16 : (k <= 4 ? 4 : (k == 5 ? 1 : (k == 6 ? 2 : 8))); }
  void lay(size_t cap_records) {
    cap = cap_records;
    size_t o = 0;
    for (int k = 0; k < 11; ++k) { at[k] = o; o += ((cap + 2) * width(k) + 255) & ~(size_t)255; }
    bytes = o;
  }
  uint8_t* colp(int k) const { return p + at[k]; }
};

int32_t device_decode_stream(midas_snps_ctx* ctx, const uint8_t* comp_base, const InflateJob* jobs, DecodeSegment& sg, size_t group_blocks, int n_slots,
                             const int64_t* ref_lens, int32_t n_ref, HostColumns (*alloc)(void*, int64_t), void* sink, DeviceDecodeResult* res,
                             int64_t* bad_job, int64_t* bad_record, char* err256) {
  auto hip_err = [&](hipError_t e, const char* what) {
    if (err256) snprintf(err256, 256, "device decode (streamed): %s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? MIDAS_SNPS_ERR_OUT_OF_MEMORY : MIDAS_SNPS_ERR_HIP;
  };
#define DS_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_err(e__, #call); } while (0)
  const bool trace = getenv("MIDAS_SNPS_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto t_last = t_begin;
  double laps[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // waited for the upload, inflate, walk, stitch, columns, direct, grow, tables
  auto lap = [&](int k) {
    const auto t = std::chrono::steady_clock::now();
    laps[k] += std::chrono::duration<double, std::milli>(t - t_last).count();
    t_last = t;
  };
  hipStream_t s = ctx->stream;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  constexpr size_t kTail = 8;                    // blocks kept behind a group's last: the record that straddles its end lies in them
  const size_t j_lo = sg.job_lo, j_hi = sg.job_hi;
  const size_t K = (j_hi - j_lo + group_blocks - 1) / group_blocks;
  const uint64_t seg_limit = jobs[j_hi - 1].upos + jobs[j_hi - 1].ulen;
  const uint64_t seg_stop = sg.stop < seg_limit ? sg.stop : seg_limit;
  struct Group { size_t b_lo, b_hi, b_ext; uint64_t u_lo, stop; size_t comp, infl, room; };
  std::vector<Group> groups(K);
  size_t slot_bytes = 0;
  for (size_t g = 0; g < K; ++g) {
    Group& G = groups[g];
    G.b_lo = j_lo + g * group_blocks;
    G.b_hi = std::min(j_hi, G.b_lo + group_blocks);
    G.b_ext = std::min(j_hi, G.b_hi + kTail);
    G.u_lo = jobs[G.b_lo].upos;
    G.stop = G.b_hi == j_hi ? seg_stop : std::min<uint64_t>(seg_stop, jobs[G.b_hi].upos);
    G.comp = (size_t)(jobs[G.b_ext - 1].cpos + jobs[G.b_ext - 1].clen + 8 - jobs[G.b_lo].cpos);
    G.infl = (size_t)(jobs[G.b_ext - 1].upos + jobs[G.b_ext - 1].ulen - G.u_lo);
    G.room = 0;
    for (size_t j = G.b_lo; j < G.b_ext; ++j) {
      if (jobs[j].cpos < jobs[G.b_lo].cpos || jobs[j].upos < G.u_lo) { if (err256) snprintf(err256, 256, "device decode: block %lld lies outside the buffers", (long long)j); return MIDAS_SNPS_ERR_INVALID_ARG; }
      G.room += jobs[j].ulen / 8u + 16u;
    }
    const size_t nj = G.b_ext - G.b_lo;
    // (behind the inflated bytes: the dead compressed bytes, tables and match lists hold the walk's tables and the record offsets --
    // 8 bytes a record of >= 36: a quarter of the inflated bytes at most)
    const size_t behind = std::max(up(G.comp + 512) + up(nj * sizeof(InflateBlock)) + up(nj * 8) + up(nj * 4) + up(G.room * 8) + 256, G.infl / 3 + ((size_t)4 << 20) + up((size_t)(n_ref > 0 ? n_ref : 1) * 8));      // (slot_layout's regions, each rounded up by itself)
    slot_bytes = std::max(slot_bytes, up(G.infl + 64) + behind);
  }
  if (n_slots > (int)K) n_slots = (int)K;
  bool pooled = false;
  void* arena_p = ctx->arena->take(slot_bytes * (size_t)n_slots, &pooled);
  if (!arena_p) { if (err256) snprintf(err256, 256, "device decode (streamed): out of device memory (%.1f GB of slots)", (double)(slot_bytes * (size_t)n_slots) / 1e9); return MIDAS_SNPS_ERR_OUT_OF_MEMORY; }
  struct Loan { std::shared_ptr<midas_arena_pool> pool; void* p; ~Loan() { if (p) pool->give(p); } } loan{ctx->arena, arena_p};
  uint8_t* const arena = static_cast<uint8_t*>(arena_p);
  const double slots_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (trace) fprintf(stderr, "[device decode] streamed: %zu groups of <= %zu blocks, %d slots of %.2f GB (allocated in %.1f ms)\n", K, group_blocks, n_slots,
                     (double)slot_bytes / 1e9, slots_ms);
  t_last = std::chrono::steady_clock::now();

  // ---- the uploader: a group's bytes and tables up on its own stream, then the group's decoder / resolver / CRC kernels on one of
  // two streams (groups alternate: the decoder is latency-bound -- ~25 ms a launch however few blocks -- and the next group's
  // workgroups fill the CUs that this group's stragglers leave idle), its statuses down into pinned memory, an event behind them --
  struct Pipe {
    std::mutex m;
    std::condition_variable cv;
    long long launched = 0, decoded = 0;
    bool abort = false;
    int32_t status = MIDAS_SNPS_OK;
    hipError_t hip = hipSuccess;
    double busy_ms = 0;
  } pipe;
  struct GroupHost { std::vector<InflateBlock> blocks; std::vector<uint32_t> want; };
  std::vector<GroupHost> host(K);
  struct Streams {
    hipStream_t up = nullptr, inf[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev;
    uint32_t* status = nullptr;       // pinned: every group's block statuses
    ~Streams() {
      for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
      if (up) (void)hipStreamDestroy(up);
      for (hipStream_t q : inf) if (q) (void)hipStreamDestroy(q);
      if (status) (void)hipHostFree(status);
    }
  } st;
  DS_TRY(hipStreamCreateWithFlags(&st.up, hipStreamNonBlocking));
  DS_TRY(hipStreamCreateWithFlags(&st.inf[0], hipStreamNonBlocking));
  DS_TRY(hipStreamCreateWithFlags(&st.inf[1], hipStreamNonBlocking));
  st.ev.assign(K, nullptr);
  for (size_t g = 0; g < K; ++g) DS_TRY(hipEventCreateWithFlags(&st.ev[g], hipEventDisableTiming));
  std::vector<size_t> status_at(K + 1, 0);
  for (size_t g = 0; g < K; ++g) status_at[g + 1] = status_at[g] + (groups[g].b_ext - groups[g].b_lo);
  DS_TRY(hipHostMalloc(reinterpret_cast<void**>(&st.status), status_at[K] * 4 + 64, kHostAllocFlags));
  auto slot_layout = [&](const Group& G, size_t* at_comp, size_t* at_blocks, size_t* at_status, size_t* at_crc, size_t* at_matches) {
    const size_t nj = G.b_ext - G.b_lo;
    *at_comp = up(G.infl + 64); *at_blocks = *at_comp + up(G.comp + 512); *at_status = *at_blocks + up(nj * sizeof(InflateBlock));
    *at_crc = *at_status + up(nj * 8); *at_matches = *at_crc + up(nj * 4);
  };
  auto inflate_params = [&](const Group& G, uint8_t* slot) {
    size_t at_comp, at_blocks, at_status, at_crc, at_matches;
    slot_layout(G, &at_comp, &at_blocks, &at_status, &at_crc, &at_matches);
    const size_t nj = G.b_ext - G.b_lo;
    InflateParams ip;
    ip.comp = slot + at_comp;
    ip.blocks = reinterpret_cast<const InflateBlock*>(slot + at_blocks);
    ip.n_blocks = (long long)nj;
    ip.out = slot;
    ip.status = reinterpret_cast<uint32_t*>(slot + at_status);
    ip.n_matches = reinterpret_cast<uint32_t*>(slot + at_status) + nj;
    ip.matches = reinterpret_cast<unsigned long long*>(slot + at_matches);
    ip.want_crc = reinterpret_cast<const uint32_t*>(slot + at_crc);
    return ip;
  };
  std::thread uploader([&] {
    (void)hipSetDevice(ctx->device);
    for (size_t g = 0; g < K; ++g) {
      {
        std::unique_lock<std::mutex> lk(pipe.m);
        pipe.cv.wait(lk, [&] { return pipe.abort || (long long)g < pipe.decoded + n_slots; });
        if (pipe.abort) return;
      }
      const Group& G = groups[g];
      uint8_t* slot = arena + (g % (size_t)n_slots) * slot_bytes;
      const auto t0 = std::chrono::steady_clock::now();
      const size_t nj = G.b_ext - G.b_lo;
      GroupHost& H = host[g];
      H.blocks.resize(nj);
      H.want.resize(nj);
      {
        unsigned long long room = 0;
        const uint64_t c0 = jobs[G.b_lo].cpos;
        for (size_t j = 0; j < nj; ++j) {
          const InflateJob& q = jobs[G.b_lo + j];
          const uint32_t cap = q.ulen / 8u + 16u;
          H.blocks[j] = InflateBlock{(unsigned long long)(q.cpos - c0), (unsigned long long)(q.upos - G.u_lo), room, q.clen, q.ulen, cap, 0u};
          H.want[j] = q.crc;
          room += cap;
        }
      }
      size_t at_comp, at_blocks, at_status, at_crc, at_matches;
      slot_layout(G, &at_comp, &at_blocks, &at_status, &at_crc, &at_matches);
      int32_t ust = copy_to_device_staged(ctx, slot + at_comp, comp_base + jobs[G.b_lo].cpos, G.comp, st.up);
      hipError_t e = hipSuccess;
      if (ust == MIDAS_SNPS_OK) {
        e = hipMemcpyAsync(slot + at_blocks, H.blocks.data(), nj * sizeof(InflateBlock), hipMemcpyHostToDevice, st.up);
        if (e == hipSuccess) e = hipMemcpyAsync(slot + at_crc, H.want.data(), nj * 4, hipMemcpyHostToDevice, st.up);
        if (e == hipSuccess) e = hipMemsetAsync(slot + at_comp + G.comp, 0, 512, st.up);
        if (e == hipSuccess) e = hipStreamSynchronize(st.up);
        if (e == hipSuccess) {
          hipStream_t q = st.inf[g & 1];
          const InflateParams ip = inflate_params(G, slot);
          e = launch_bgzf_inflate(ip, q);
          if (e == hipSuccess) e = hipMemcpyAsync(st.status + status_at[g], ip.status, nj * 4, hipMemcpyDeviceToHost, q);
          if (e == hipSuccess) e = hipEventRecord(st.ev[g], q);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); ust = e == hipErrorOutOfMemory ? MIDAS_SNPS_ERR_OUT_OF_MEMORY : MIDAS_SNPS_ERR_HIP; }
      }
      {
        std::lock_guard<std::mutex> lk(pipe.m);
        pipe.busy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ust != MIDAS_SNPS_OK) { pipe.status = ust; pipe.hip = e; pipe.abort = true; }
        else pipe.launched = (long long)g + 1;
      }
      pipe.cv.notify_all();
      if (ust != MIDAS_SNPS_OK) return;
    }
  });
  struct Join {       // (every way out: the uploader is told to stop and waited for, the streams' work too)
    Pipe& pipe; std::thread& t; Streams& st;
    ~Join() {
      { std::lock_guard<std::mutex> lk(pipe.m); pipe.abort = true; }
      pipe.cv.notify_all();
      if (t.joinable()) t.join();
      for (hipStream_t q : st.inf) if (q) (void)hipStreamSynchronize(q);
      (void)hipGetLastError();
    }
  } join{pipe, uploader, st};

  // ---- the result's arrays ----------------------------------------------------------------------------------------------------------
  StreamColumns cols;
  struct Pay { uint8_t* p = nullptr; size_t cap_units = 0; } pay;
  struct Owned { StreamColumns& c; Pay& y; bool keep = false; ~Owned() { if (!keep) { if (c.p) (void)hipFree(c.p); if (y.p) (void)hipFree(y.p); } } } owned{cols, pay};
  long long N = 0;                                  // records so far
  long long base[4] = {0, 0, 0, 0};                 // what the offset columns came to so far: SEQ bytes, QUAL bytes, CIGAR ops, payload units
  const double span_total = (double)(seg_stop > sg.from ? seg_stop - sg.from : 1);
  auto grow_columns = [&](size_t need) -> int32_t {       // room for `need` records
    if (cols.p && need <= cols.cap) return MIDAS_SNPS_OK;
    StreamColumns nc;
    nc.lay(need);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&nc.p), nc.bytes);
    if (e != hipSuccess) return hip_err(e, "the records' arrays");
    if (cols.p) {
      for (int k = 0; k < 11; ++k) {
        const size_t n = (size_t)N + (k == 0 || k >= 7 ? 1 : 0);
        e = hipMemcpyAsync(nc.p + nc.at[k], cols.p + cols.at[k], n * StreamColumns::width(k), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { (void)hipFree(nc.p); return hip_err(e, "the records' arrays moved"); }
      }
      e = hipStreamSynchronize(s);
      (void)hipFree(cols.p);
      if (e != hipSuccess) { (void)hipFree(nc.p); return hip_err(e, "the records' arrays moved"); }
    }
    cols = nc;
    return MIDAS_SNPS_OK;
  };
  auto grow_payload = [&](size_t need_units) -> int32_t {
    if (pay.p && need_units <= pay.cap_units) return MIDAS_SNPS_OK;
    uint8_t* q = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&q), need_units * 8 + 64);
    if (e != hipSuccess) return hip_err(e, "the reads' payload");
    if (pay.p) {
      e = base[3] > 0 ? hipMemcpyAsync(q, pay.p, (size_t)base[3] * 8, hipMemcpyDeviceToDevice, s) : hipSuccess;
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      (void)hipFree(pay.p);
      if (e != hipSuccess) { (void)hipFree(q); return hip_err(e, "the reads' payload moved"); }
    }
    pay.p = q;
    pay.cap_units = need_units;
    return MIDAS_SNPS_OK;
  };

  // ---- group by group -----------------------------------------------------------------------------------------------------------------
  sg.first = ~0ull; sg.n_records = 0; sg.n_unmapped = 0; sg.first_unmapped = ~0ull; sg.end = seg_stop;
  unsigned long long cur = sg.exact ? sg.from : ~0ull;       // GLOBAL buffer offset of the next record (~0: group 0 guesses it)
  int regrown = 0, rounds_total = 0;
  const unsigned long long kChunk = 32768ull;
  for (size_t g = 0; g < K; ++g) {
    const Group& G = groups[g];
    uint8_t* const slot = arena + (g % (size_t)n_slots) * slot_bytes;
    const size_t nj = G.b_ext - G.b_lo;
    size_t at_comp, at_blocks, at_status, at_crc, at_matches;
    slot_layout(G, &at_comp, &at_blocks, &at_status, &at_crc, &at_matches);
    lap(7);
    {
      std::unique_lock<std::mutex> lk(pipe.m);
      pipe.cv.wait(lk, [&] { return pipe.abort || pipe.launched > (long long)g; });
      if (pipe.launched <= (long long)g) {
        if (pipe.hip != hipSuccess) return hip_err(pipe.hip, "a group's blocks to the device and its decoder's launch");
        if (err256) snprintf(err256, 256, "device decode (streamed): blocks to the device: %s", ctx->error_text().c_str());
        return pipe.status != MIDAS_SNPS_OK ? pipe.status : MIDAS_SNPS_ERR_HIP;
      }
    }
    lap(0);
    DS_TRY(hipEventSynchronize(st.ev[g]));        // the group is inflated, resolved, checked; its statuses are down
    const InflateParams ip = inflate_params(G, slot);
    const std::vector<InflateBlock>& blocks = host[g].blocks;
    const std::vector<uint32_t>& want = host[g].want;
    std::vector<uint32_t> status(st.status + status_at[g], st.status + status_at[g] + nj);
    {
      bool again = false;
      const hipError_t ae = inflate_again(ip, blocks, want, status, s, &again);
      if (ae != hipSuccess) return hip_err(ae, "streams decoded again");
    }
    for (size_t k = 0; k < nj; ++k) {
      if (status[k] != 0u) {
        *bad_job = (int64_t)(G.b_lo + k);
        if (err256) snprintf(err256, 256, "corrupt BGZF block %lld (code %u)", (long long)(G.b_lo + k), status[k]);
        return MIDAS_SNPS_ERR_BAD_LAYOUT;
      }
    }
    lap(1);
    // ---- the record walk of the group: chunks over [from, stop) in the slot's own offsets -----------------------------------------
    const unsigned long long l_limit = G.infl;
    const unsigned long long l_stop = G.stop > G.u_lo ? (unsigned long long)(G.stop - G.u_lo) : 0ull;
    unsigned long long l_from;
    bool exact;
    if (cur != ~0ull) { exact = true; l_from = cur >= G.u_lo ? cur - G.u_lo : 0ull; }
    else { exact = false; l_from = g == 0 && sg.from >= G.u_lo ? (unsigned long long)(sg.from - G.u_lo) : 0ull; }
    if (cur != ~0ull && cur < G.u_lo) {        // (cannot be: the chain of the group before ended at or behind this group's first byte)
      if (err256) snprintf(err256, 256, "device decode (streamed): the record chain fell behind group %zu", g);
      return MIDAS_SNPS_ERR_UNSUPPORTED;
    }
    std::vector<unsigned long long> h_lo, h_hi, h_stop, h_limit, h_start;
    std::vector<uint8_t> h_forced;
    for (unsigned long long lo = l_from; lo < l_stop; lo += kChunk) {
      h_lo.push_back(lo); h_hi.push_back(lo + kChunk < l_stop ? lo + kChunk : l_stop); h_stop.push_back(l_stop); h_limit.push_back(l_limit);
      const bool first = lo == l_from;
      h_forced.push_back(first && exact ? 1 : 0);
      h_start.push_back(first && exact ? l_from : ~0ull);
    }
    const long long n_chunks = (long long)h_lo.size();
    uint8_t* const scratch = slot + at_comp;
    const size_t scratch_bytes = slot_bytes - at_comp;
    size_t at = 0;
    auto take = [&](size_t bytes) -> uint8_t* { uint8_t* q = scratch + at; at += up(bytes); return at <= scratch_bytes ? q : nullptr; };
    const size_t nc1 = (size_t)(n_chunks > 0 ? n_chunks : 1);
    BamWalkParams wp;
    wp.d = slot; wp.n_ref = n_ref; wp.n_chunks = n_chunks;
    long long* d_ref_lens = reinterpret_cast<long long*>(take((size_t)(n_ref > 0 ? n_ref : 1) * 8));
    unsigned long long* d_lo = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
    unsigned long long* d_hi = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
    unsigned long long* d_stop = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
    unsigned long long* d_limit = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
    uint8_t* d_forced = take(nc1);
    wp.start = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
    wp.end = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
    wp.kept = reinterpret_cast<uint32_t*>(take(nc1 * 4));
    wp.unmapped = reinterpret_cast<uint32_t*>(take(nc1 * 4));
    wp.first_unmapped = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
    wp.bad = reinterpret_cast<uint32_t*>(take(nc1 * 4));
    unsigned long long* d_base = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
    long long* d_list = reinterpret_cast<long long*>(take(4096 * 8));
    if (!d_list) { if (err256) snprintf(err256, 256, "device decode (streamed): a slot is too small for the walk"); return MIDAS_SNPS_ERR_OUT_OF_MEMORY; }
    wp.lo = d_lo; wp.hi = d_hi; wp.stop = d_stop; wp.limit = d_limit; wp.forced = d_forced; wp.ref_lens = d_ref_lens;
    if (n_ref > 0) DS_TRY(hipMemcpyAsync(d_ref_lens, ref_lens, (size_t)n_ref * 8, hipMemcpyHostToDevice, s));
    std::vector<unsigned long long> h_end(nc1), h_base(nc1), h_fu(nc1);
    std::vector<uint32_t> h_kept(nc1), h_bad(nc1), h_unm(nc1);
    if (n_chunks > 0) {
      DS_TRY(hipMemcpyAsync(d_lo, h_lo.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
      DS_TRY(hipMemcpyAsync(d_hi, h_hi.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
      DS_TRY(hipMemcpyAsync(d_stop, h_stop.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
      DS_TRY(hipMemcpyAsync(d_limit, h_limit.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
      DS_TRY(hipMemcpyAsync(d_forced, h_forced.data(), (size_t)n_chunks, hipMemcpyHostToDevice, s));
      DS_TRY(hipMemcpyAsync(wp.start, h_start.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
      DS_TRY(launch_bam_walk(wp, nullptr, 0, s));
      DS_TRY(hipMemcpyAsync(h_start.data(), wp.start, (size_t)n_chunks * 8, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(h_end.data(), wp.end, (size_t)n_chunks * 8, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(h_kept.data(), wp.kept, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(h_unm.data(), wp.unmapped, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(h_fu.data(), wp.first_unmapped, (size_t)n_chunks * 8, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(h_bad.data(), wp.bad, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, s));
      DS_TRY(hipStreamSynchronize(s));
    }
    lap(2);
    // stitch in order (device_decode_run's loop, one segment); a record that overruns the bytes the group keeps behind its end: not for
    // this decode
    unsigned long long lcur = exact ? l_from : ~0ull;
    unsigned long long g_first = ~0ull;
    long long g_records = 0;
    for (size_t c = 0; c < (size_t)n_chunks;) {
      if (lcur == ~0ull) {
        if (h_start[c] == ~0ull) { h_kept[c] = 0u; h_unm[c] = 0u; ++c; continue; }
        lcur = h_start[c];
      }
      if (lcur >= h_hi[c] || lcur + 4 > h_limit[c]) { h_kept[c] = 0u; h_unm[c] = 0u; h_start[c] = ~0ull; ++c; continue; }
      if (h_start[c] == lcur) {
        if (h_bad[c]) {
          if (G.b_ext < j_hi) return kStreamFallback;       // (the group's own bytes end where the record goes on: the one-arena decode holds it whole)
          *bad_record = -2;
          return MIDAS_SNPS_ERR_BAD_LAYOUT;
        }
        if (g_first == ~0ull) g_first = lcur;
        g_records += h_kept[c];
        if (h_unm[c] && sg.first_unmapped == ~0ull) sg.first_unmapped = h_fu[c] + G.u_lo;
        sg.n_unmapped += h_unm[c];
        lcur = h_end[c];
        ++c;
        continue;
      }
      if (++rounds_total > 4096) {
        if (err256) snprintf(err256, 256, "device decode: the record boundaries did not settle");
        return MIDAS_SNPS_ERR_UNSUPPORTED;
      }
      const long long one = (long long)c;
      DS_TRY(hipMemcpyAsync(wp.start + c, &lcur, 8, hipMemcpyHostToDevice, s));
      DS_TRY(hipMemcpyAsync(d_list, &one, 8, hipMemcpyHostToDevice, s));
      DS_TRY(launch_bam_walk(wp, d_list, 1, s));
      DS_TRY(hipMemcpyAsync(&h_end[c], wp.end + c, 8, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(&h_kept[c], wp.kept + c, 4, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(&h_unm[c], wp.unmapped + c, 4, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(&h_fu[c], wp.first_unmapped + c, 8, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(&h_bad[c], wp.bad + c, 4, hipMemcpyDeviceToHost, s));
      DS_TRY(hipStreamSynchronize(s));
      h_start[c] = lcur;
    }
    if (lcur != ~0ull) {      // (else: a group that had to guess found no record boundary: the next one guesses too)
      cur = lcur + G.u_lo;
      sg.end = cur;
    }
    if (g_first != ~0ull && sg.first == ~0ull) sg.first = g_first + G.u_lo;
    unsigned long long n_rec = 0;
    for (long long c = 0; c < n_chunks; ++c) { h_base[(size_t)c] = n_rec; n_rec += h_kept[(size_t)c]; }
    lap(3);
    const long long n = (long long)n_rec;
    if (n > 0) {
      DS_TRY(hipMemcpyAsync(wp.start, h_start.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
      DS_TRY(hipMemcpyAsync(wp.kept, h_kept.data(), (size_t)n_chunks * 4, hipMemcpyHostToDevice, s));
      DS_TRY(hipMemcpyAsync(d_base, h_base.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
      // room in the result: from the records per inflated byte so far, + 3 %
      if (!cols.p || (size_t)(N + n) > cols.cap) {
        const double done = (double)(cur != ~0ull && cur > sg.from ? cur - sg.from : 1);
        const double est = (double)(N + n) * std::max(1.0, span_total / done) * 1.03 + 4096.0;
        if (cols.p) ++regrown;
        const int32_t gst = grow_columns(std::max((size_t)(N + n), (size_t)est));
        if (gst != MIDAS_SNPS_OK) return gst;
        lap(6);
      }
      const size_t n1 = (size_t)n + 1;
      unsigned long long* d_rec = reinterpret_cast<unsigned long long*>(take(n1 * 8));
      unsigned long long* d_badrec = reinterpret_cast<unsigned long long*>(take(8));
      long long* d_scan = reinterpret_cast<long long*>(take(bam_scan_scratch_bytes(n)));
      if (!d_scan) { if (err256) snprintf(err256, 256, "device decode (streamed): a slot is too small for the record offsets"); return MIDAS_SNPS_ERR_OUT_OF_MEMORY; }
      BamColumnsParams cp;
      cp.d = slot; cp.rec_off = d_rec; cp.n = n; cp.n_ref = n_ref;
      cp.refid = reinterpret_cast<int32_t*>(cols.colp(1)) + N; cp.pos = reinterpret_cast<int32_t*>(cols.colp(2)) + N; cp.nm = reinterpret_cast<int32_t*>(cols.colp(3)) + N; cp.l_seq = reinterpret_cast<int32_t*>(cols.colp(4)) + N;
      cp.mapq = reinterpret_cast<uint8_t*>(cols.colp(5)) + N; cp.flag = reinterpret_cast<uint16_t*>(cols.colp(6)) + N;
      cp.seq_off = reinterpret_cast<long long*>(cols.colp(7)) + N; cp.qual_off = reinterpret_cast<long long*>(cols.colp(8)) + N; cp.cigar_off = reinterpret_cast<long long*>(cols.colp(9)) + N;
      cp.unit_off = reinterpret_cast<long long*>(cols.colp(10)) + N;
      cp.span = nullptr;
      cp.bad_record = d_badrec;
      for (int k = 0; k < 4; ++k) cp.base[k] = base[k];
      DS_TRY(hipMemsetAsync(cp.bad_record, 0xFF, 8, s));
      DS_TRY(launch_bam_offsets(wp, d_base, d_rec, s));
      DS_TRY(launch_bam_columns(cp, d_scan, s));
      unsigned long long h_bad_record = ~0ull;
      long long ends[4] = {0, 0, 0, 0};
      DS_TRY(hipMemcpyAsync(&h_bad_record, cp.bad_record, 8, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(&ends[0], cp.seq_off + n, 8, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(&ends[1], cp.qual_off + n, 8, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(&ends[2], cp.cigar_off + n, 8, hipMemcpyDeviceToHost, s));
      DS_TRY(hipMemcpyAsync(&ends[3], cp.unit_off + n, 8, hipMemcpyDeviceToHost, s));
      DS_TRY(hipStreamSynchronize(s));
      lap(4);
      if (h_bad_record != ~0ull) { *bad_record = (int64_t)h_bad_record + N; return MIDAS_SNPS_ERR_BAD_LAYOUT; }
      if ((unsigned long long)ends[3] > kMaxDirectPayloadUnits) {
        if (err256) snprintf(err256, 256, "device decode: %llu bytes of read payload exceed the 32 GiB the direct layout addresses", (unsigned long long)ends[3] * 8ull);
        return MIDAS_SNPS_ERR_UNSUPPORTED;
      }
      if (!pay.p || (size_t)ends[3] > pay.cap_units) {
        const double done = (double)(cur != ~0ull && cur > sg.from ? cur - sg.from : 1);
        const double est = (double)ends[3] * std::max(1.0, span_total / done) * 1.03 + 65536.0;
        if (pay.p) ++regrown;
        const int32_t gst = grow_payload(std::max((size_t)ends[3], (size_t)std::min(est, (double)kMaxDirectPayloadUnits)));
        if (gst != MIDAS_SNPS_OK) return gst;
        lap(6);
      }
      BamDirectParams dp;
      dp.stream = slot; dp.rec_off = d_rec; dp.n_records = n;
      dp.pos = cp.pos; dp.nm = cp.nm; dp.unit_off = cp.unit_off;
      dp.rec = reinterpret_cast<DirectRec*>(cols.colp(0)) + N; dp.payload = pay.p;
      DS_TRY(launch_bam_direct(dp, ctx->prop.multiProcessorCount, s));
      DS_TRY(hipStreamSynchronize(s));
      lap(5);
      N += n;
      for (int k = 0; k < 4; ++k) base[k] = ends[k];
    }
    {
      std::lock_guard<std::mutex> lk(pipe.m);
      pipe.decoded = (long long)g + 1;
    }
    pipe.cv.notify_all();
    if (cur != ~0ull && cur >= seg_stop) break;       // (the wanted records end here: nothing of the groups behind is needed)
  }
  sg.n_records = N;
  if (N == 0) {        // (no record at all: the caller's columns are empty; nothing stays on the device)
    const HostColumns hc = alloc(sink, 0);
    (void)hc;
    res->n_records = 0; res->seq_bytes = 0; res->qual_bytes = 0; res->n_cigar = 0;
    const int32_t g0 = grow_columns(1);
    if (g0 != MIDAS_SNPS_OK) return g0;
    const int32_t g1 = grow_payload(8);
    if (g1 != MIDAS_SNPS_OK) return g1;
    DS_TRY(hipMemsetAsync(cols.p, 0, cols.bytes, s));
  }
  DS_TRY(hipMemsetAsync(pay.p + (size_t)base[3] * 8, 0, 64, s));      // (a lane's 16-byte loads may overhang the last read)
  {       // the uploader has nothing left to do: the ring is the copy-down's again
    { std::lock_guard<std::mutex> lk(pipe.m); pipe.abort = true; }
    pipe.cv.notify_all();
    if (uploader.joinable()) uploader.join();
    for (hipStream_t q : st.inf) DS_TRY(hipStreamSynchronize(q));
  }
  if (N > 0) {
    const HostColumns hc = alloc(sink, N);
    if (!hc.refid) { DS_TRY(hipStreamSynchronize(s)); return MIDAS_SNPS_OK; }
    const int32_t dst = copy_to_host(ctx, hc.refid, reinterpret_cast<int32_t*>(cols.colp(1)), (size_t)N * 4);
    if (dst != MIDAS_SNPS_OK) { if (err256) snprintf(err256, 256, "device decode: columns to host: %s", ctx->error_text().c_str()); return dst; }
  }
  DS_TRY(hipStreamSynchronize(s));
  if (trace) {
    const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    fprintf(stderr, "[device decode] streamed: %lld records in %.1f ms: waited for a group's launch %.1f (the uploader worked %.1f), for its decoder %.1f, walk %.1f, stitch %.1f, "
                    "columns %.1f, direct layout %.1f, result grown %.1f (%d times), tables %.1f ms; %d chunk(s) walked again; result %.2f GB\n",
            N, total_ms, laps[0], pipe.busy_ms, laps[1], laps[2], laps[3], laps[4], laps[5], laps[6], regrown, laps[7], rounds_total,
            (double)(cols.bytes + pay.cap_units * 8) / 1e9);
  }
  res->n_records = N; res->seq_bytes = base[0]; res->qual_bytes = base[1]; res->n_cigar = base[2];
  ResidentReads& rr = res->resident;
  rr.rec = reinterpret_cast<DirectRec*>(cols.colp(0)); rr.payload = pay.p; rr.refid = reinterpret_cast<int32_t*>(cols.colp(1)); rr.pos = reinterpret_cast<int32_t*>(cols.colp(2)); rr.nm = reinterpret_cast<int32_t*>(cols.colp(3));
  rr.l_seq = reinterpret_cast<int32_t*>(cols.colp(4)); rr.mapq = reinterpret_cast<uint8_t*>(cols.colp(5)); rr.flag = reinterpret_cast<uint16_t*>(cols.colp(6));
  rr.seq_off = reinterpret_cast<int64_t*>(cols.colp(7)); rr.qual_off = reinterpret_cast<int64_t*>(cols.colp(8)); rr.cigar_off = reinterpret_cast<int64_t*>(cols.colp(9)); rr.unit_off = reinterpret_cast<int64_t*>(cols.colp(10));
  rr.stream = nullptr; rr.rec_off = nullptr;       // (no inflated stream is kept: raw columns come out of the direct layout)
  rr.payload_units = base[3];
  res->dev_owner = new TwoBuffers{cols.p, pay.p};
  res->dev_free = two_buffers_free;
  owned.keep = true;
#undef DS_TRY
  return MIDAS_SNPS_OK;
}

// DeviceDecoder::run (hostio.h): BGZF blocks of a BAM -- the whole file's, a rank's slice, or the runs that hold a rank's contigs --
// decoded on the device: up, inflated, resolved and CRC-checked (bgzf_inflate.hip), records found and decoded (bam_walk.hip), SEQ /
// QUAL / CIGAR cut out where the stream lies; the small columns are all that comes down.
int32_t device_decode_run(void* user, const uint8_t* comp_base, const InflateJob* jobs, size_t n_jobs, uint64_t total, DecodeSegment* segs,
                          size_t n_segs, const int64_t* ref_lens, int32_t n_ref, int payload, int extra, HostColumns (*alloc)(void*, int64_t),
                          void* sink, DeviceDecodeResult* res, int64_t* bad_job, int64_t* bad_record, char* err256) {
  midas_snps_ctx* ctx = static_cast<midas_snps_ctx*>(user);
  *bad_job = -1;
  *bad_record = -1;
  auto hip_err = [&](hipError_t e, const char* what) {
    if (err256) snprintf(err256, 256, "device decode: %s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? MIDAS_SNPS_ERR_OUT_OF_MEMORY : MIDAS_SNPS_ERR_HIP;
  };
#define DEC_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_err(e__, #call); } while (0)
  std::lock_guard<std::mutex> g(ctx->device_mutex);
  const bool trace = getenv("MIDAS_SNPS_TRACE") != nullptr;       // where the call spends its time, on stderr
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[device decode] %-30s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  DEC_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  // ---- a resident decode of ONE run of blocks that fills the device several times over: group by group (device_decode_stream) ----
  if (payload == 2 && !extra && n_segs == 1 && segs[0].job_lo < segs[0].job_hi && segs[0].job_hi <= n_jobs) {
    const size_t nb = segs[0].job_hi - segs[0].job_lo;
    size_t group = 0;
    int slots = 3;
    const char* on = getenv("MIDAS_SNPS_DECODE_STREAM");
    if (!on || atoi(on) != 0) {
      // (half a device-fill of the decoder's workgroups a group: two groups' kernels run side by side, on alternating streams)
      group = (size_t)std::max(64ll, bgzf_inflate_wave_blocks(ctx->prop.multiProcessorCount) / 2);
      if (const char* e = getenv("MIDAS_SNPS_DECODE_GROUP_BLOCKS")) group = (size_t)std::max(1ll, atoll(e));
      if (const char* e = getenv("MIDAS_SNPS_DECODE_SLOT_MB")) {        // a slot is ~2.5 x its blocks' inflated bytes (<= 64 KiB each)
        const size_t cap_blocks = (size_t)std::max(16ll, atoll(e) * (1ll << 20) / (160ll << 10));
        group = std::min(group, cap_blocks);
      }
      if (const char* e = getenv("MIDAS_SNPS_DECODE_SLOTS")) slots = std::max(1, std::min(4, atoi(e)));
    }
    if (group && nb > group + group / 4) {
      const size_t K = (nb + group - 1) / group;
      group = (nb + K - 1) / K;        // (equal groups, none above a wave)
      const int32_t sst = device_decode_stream(ctx, comp_base, jobs, segs[0], group, slots, ref_lens, n_ref, alloc, sink, res, bad_job, bad_record, err256);
      if (sst != kStreamFallback) return sst;
      if (trace) fprintf(stderr, "[device decode] streamed decode gave up (a record longer than what a group keeps behind its end): one arena\n");
      *bad_job = -1;
      *bad_record = -1;
    }
  }
  // ---- the compressed bytes: every segment's blocks are consecutive in the file, the segments go up back to back ----------
  std::vector<size_t> seg_at(n_segs + 1, 0);      // where segment k's bytes start in the device's copy
  for (size_t k = 0; k < n_segs; ++k) {
    const DecodeSegment& sg = segs[k];
    if (sg.job_lo >= sg.job_hi || sg.job_hi > n_jobs) { if (err256) snprintf(err256, 256, "device decode: empty segment"); return MIDAS_SNPS_ERR_INVALID_ARG; }
    const size_t bytes = (size_t)(jobs[sg.job_hi - 1].cpos + jobs[sg.job_hi - 1].clen + 8 - jobs[sg.job_lo].cpos);
    seg_at[k + 1] = seg_at[k] + bytes;
  }
  const size_t comp_bytes = seg_at[n_segs];
  // ---- the arena: | inflated bytes | compressed bytes | blocks | status | crc | match lists |; everything behind the inflated
  // bytes is scratch once the blocks are resolved, and the columns are laid over it
  std::vector<InflateBlock> blocks(n_jobs);
  std::vector<uint32_t> want(n_jobs);
  unsigned long long n_match_room = 0;
  for (size_t k = 0; k < n_segs; ++k) {
    const uint64_t c0 = jobs[segs[k].job_lo].cpos;
    for (size_t j = segs[k].job_lo; j < segs[k].job_hi; ++j) {
      if (jobs[j].upos + jobs[j].ulen > total || jobs[j].cpos < c0) {
        if (err256) snprintf(err256, 256, "device decode: block %lld lies outside the buffers", (long long)j);
        return MIDAS_SNPS_ERR_INVALID_ARG;
      }
      const uint32_t cap = jobs[j].ulen / 8u + 16u;
      blocks[j] = InflateBlock{(unsigned long long)(seg_at[k] + (jobs[j].cpos - c0)), jobs[j].upos, n_match_room, jobs[j].clen, jobs[j].ulen, cap, 0u};
      want[j] = jobs[j].crc;
      n_match_room += cap;
    }
  }
  const size_t at_comp = up((size_t)total + 64), at_blocks = at_comp + up(comp_bytes + 512),
               at_status = at_blocks + up(n_jobs * sizeof(InflateBlock)), at_crc = at_status + up(n_jobs * 8),
               at_matches = at_crc + up(n_jobs * 4);
  // (a BAM's columns are ~0.95 of its inflated bytes, the offsets and small columns ~0.2: the scratch must hold them too)
  const size_t scratch_need = std::max(up(comp_bytes + 512) + up(n_jobs * sizeof(InflateBlock)) + up(n_jobs * 12) + up((size_t)n_match_room * 8),
                                       (size_t)total + (size_t)total / 3 + ((size_t)16 << 20));
  const size_t arena_bytes = at_comp + scratch_need;
  bool pooled = false;
  void* arena_p = ctx->arena->take(arena_bytes, &pooled);
  if (!arena_p) { if (err256) snprintf(err256, 256, "device decode: out of device memory (%.1f GB)", (double)arena_bytes / 1e9); return MIDAS_SNPS_ERR_OUT_OF_MEMORY; }
  struct Loan { std::shared_ptr<midas_arena_pool> pool; void* p; ~Loan() { if (p) pool->give(p); } } loan{ctx->arena, arena_p};
  uint8_t* const base = static_cast<uint8_t*>(arena_p);
  lap("arena");
  // ---- blocks up, inflate, resolve, check ------------------------------------------------------------------------------------
  DEC_TRY(hipMemcpyAsync(base + at_blocks, blocks.data(), n_jobs * sizeof(InflateBlock), hipMemcpyHostToDevice, s));
  DEC_TRY(hipMemcpyAsync(base + at_crc, want.data(), n_jobs * 4, hipMemcpyHostToDevice, s));
  DEC_TRY(hipMemsetAsync(base + at_comp + comp_bytes, 0, 512, s));
  InflateParams ip;
  ip.comp = base + at_comp;
  ip.blocks = reinterpret_cast<const InflateBlock*>(base + at_blocks);
  ip.n_blocks = (long long)n_jobs;
  ip.out = base;
  ip.status = reinterpret_cast<uint32_t*>(base + at_status);
  ip.n_matches = reinterpret_cast<uint32_t*>(base + at_status) + n_jobs;
  ip.matches = reinterpret_cast<unsigned long long*>(base + at_matches);
  ip.want_crc = reinterpret_cast<const uint32_t*>(base + at_crc);
  auto upload = [&](size_t dev_lo, size_t dev_hi) -> int32_t {      // the bytes [dev_lo, dev_hi) of the device's copy, segment by segment
    for (size_t k = 0; k < n_segs; ++k) {
      const size_t lo = std::max(dev_lo, seg_at[k]), hi = std::min(dev_hi, seg_at[k + 1]);
      if (lo >= hi) continue;
      const int32_t cst = copy_to_device_staged(ctx, base + at_comp + lo, comp_base + jobs[segs[k].job_lo].cpos + (lo - seg_at[k]), hi - lo, s);
      if (cst != MIDAS_SNPS_OK) { if (err256) snprintf(err256, 256, "device decode: blocks to the device: %s", ctx->error_text().c_str()); return cst; }
    }
    return MIDAS_SNPS_OK;
  };
  // (Inflating a first group of blocks while the next group's bytes go up -- four groups, a stream each -- was built and
  // measured: 216 ms against 188 ms for the whole decode of configs[2]'s BAM on the same box.  The copy threads and the
  // link are slowed by the running kernels by more than the overlap wins.  One upload, one launch.)
  {
    const int32_t ust = upload(0, comp_bytes);
    if (ust != MIDAS_SNPS_OK) return ust;
  }
  if (trace) {      // (phase by phase, each waited for)
    DEC_TRY(hipStreamSynchronize(s)); lap("blocks up");
    DEC_TRY(launch_bgzf_inflate(ip, s, 1)); DEC_TRY(hipStreamSynchronize(s)); lap("  inflate kernel");
    DEC_TRY(launch_bgzf_inflate(ip, s, 2 | 8)); DEC_TRY(hipStreamSynchronize(s)); lap("  resolve kernel");
    DEC_TRY(launch_bgzf_inflate(ip, s, 4)); DEC_TRY(hipStreamSynchronize(s)); lap("  crc kernel");
  } else {
    DEC_TRY(launch_bgzf_inflate(ip, s));
  }
  std::vector<uint32_t> status(n_jobs);
  DEC_TRY(hipMemcpyAsync(status.data(), ip.status, n_jobs * 4, hipMemcpyDeviceToHost, s));
  DEC_TRY(hipStreamSynchronize(s));
  lap("blocks up, inflate, resolve, crc");
  {   // the streams whose matches did not fit: again, with the bound's room
    bool again = false;
    const hipError_t ae = inflate_again(ip, blocks, want, status, s, &again);
    if (ae != hipSuccess) return hip_err(ae, "streams decoded again");
    if (again) lap("streams decoded again");
  }
  for (size_t k = 0; k < n_jobs; ++k) {
    if (status[k] != 0u) {
      *bad_job = (int64_t)k;
      if (err256) snprintf(err256, 256, "corrupt BGZF block %lld (code %u)", (long long)k, status[k]);
      return MIDAS_SNPS_ERR_BAD_LAYOUT;
    }
  }
  // ---- the record walk: chunks of at most 32 KiB, laid out segment by segment over [from, stop) ---------------------------
  const unsigned long long kChunk = 32768ull;
  std::vector<unsigned long long> h_lo, h_hi, h_stop, h_limit, h_start;
  std::vector<uint8_t> h_forced;
  std::vector<size_t> seg_chunk(n_segs + 1, 0);
  for (size_t k = 0; k < n_segs; ++k) {
    const DecodeSegment& sg = segs[k];
    const unsigned long long limit = jobs[sg.job_hi - 1].upos + jobs[sg.job_hi - 1].ulen;
    const unsigned long long stop = sg.stop < limit ? sg.stop : limit;
    for (unsigned long long lo = sg.from; lo < stop; lo += kChunk) {
      h_lo.push_back(lo); h_hi.push_back(lo + kChunk < stop ? lo + kChunk : stop); h_stop.push_back(stop); h_limit.push_back(limit);
      const bool first = lo == sg.from;
      h_forced.push_back(first && sg.exact ? 1 : 0);
      h_start.push_back(first && sg.exact ? sg.from : ~0ull);
    }
    seg_chunk[k + 1] = h_lo.size();
  }
  const long long n_chunks = (long long)h_lo.size();
  uint8_t* const scratch = base + at_comp;
  const size_t scratch_bytes = arena_bytes - at_comp;
  size_t at = 0;
  auto take = [&](size_t bytes) -> uint8_t* { uint8_t* q = scratch + at; at += up(bytes); return at <= scratch_bytes ? q : nullptr; };
  const size_t nc1 = (size_t)(n_chunks > 0 ? n_chunks : 1);
  BamWalkParams wp;
  wp.d = base; wp.n_ref = n_ref; wp.n_chunks = n_chunks;
  long long* d_ref_lens = reinterpret_cast<long long*>(take((size_t)(n_ref > 0 ? n_ref : 1) * 8));
  unsigned long long* d_lo = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
  unsigned long long* d_hi = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
  unsigned long long* d_stop = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
  unsigned long long* d_limit = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
  uint8_t* d_forced = take(nc1);
  wp.start = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
  wp.end = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
  wp.kept = reinterpret_cast<uint32_t*>(take(nc1 * 4));
  wp.unmapped = reinterpret_cast<uint32_t*>(take(nc1 * 4));
  wp.first_unmapped = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
  wp.bad = reinterpret_cast<uint32_t*>(take(nc1 * 4));
  unsigned long long* d_base = reinterpret_cast<unsigned long long*>(take(nc1 * 8));
  long long* d_list = reinterpret_cast<long long*>(take(4096 * 8));
  if (!d_list) { if (err256) snprintf(err256, 256, "device decode: the arena is too small for the walk"); return MIDAS_SNPS_ERR_OUT_OF_MEMORY; }
  wp.lo = d_lo; wp.hi = d_hi; wp.stop = d_stop; wp.limit = d_limit; wp.forced = d_forced; wp.ref_lens = d_ref_lens;
  if (n_ref > 0) DEC_TRY(hipMemcpyAsync(d_ref_lens, ref_lens, (size_t)n_ref * 8, hipMemcpyHostToDevice, s));
  std::vector<unsigned long long> h_end(nc1), h_base(nc1), h_fu(nc1);
  std::vector<uint32_t> h_kept(nc1), h_bad(nc1), h_unm(nc1);
  if (n_chunks > 0) {
    DEC_TRY(hipMemcpyAsync(d_lo, h_lo.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
    DEC_TRY(hipMemcpyAsync(d_hi, h_hi.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
    DEC_TRY(hipMemcpyAsync(d_stop, h_stop.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
    DEC_TRY(hipMemcpyAsync(d_limit, h_limit.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
    DEC_TRY(hipMemcpyAsync(d_forced, h_forced.data(), (size_t)n_chunks, hipMemcpyHostToDevice, s));
    DEC_TRY(hipMemcpyAsync(wp.start, h_start.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
    DEC_TRY(launch_bam_walk(wp, nullptr, 0, s));
    DEC_TRY(hipMemcpyAsync(h_start.data(), wp.start, (size_t)n_chunks * 8, hipMemcpyDeviceToHost, s));
    DEC_TRY(hipMemcpyAsync(h_end.data(), wp.end, (size_t)n_chunks * 8, hipMemcpyDeviceToHost, s));
    DEC_TRY(hipMemcpyAsync(h_kept.data(), wp.kept, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, s));
    DEC_TRY(hipMemcpyAsync(h_unm.data(), wp.unmapped, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, s));
    DEC_TRY(hipMemcpyAsync(h_fu.data(), wp.first_unmapped, (size_t)n_chunks * 8, hipMemcpyDeviceToHost, s));
    DEC_TRY(hipMemcpyAsync(h_bad.data(), wp.bad, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, s));
    DEC_TRY(hipStreamSynchronize(s));
  }
  lap("walk (guesses)");
  // stitch every segment in order; a chunk whose guess the chain does not hit is walked again from where the chain stands
  int rounds = 0;
  for (size_t k = 0; k < n_segs; ++k) {
    DecodeSegment& sg = segs[k];
    sg.first = ~0ull; sg.n_records = 0; sg.n_unmapped = 0; sg.first_unmapped = ~0ull;
    unsigned long long cur = sg.exact ? sg.from : ~0ull;
    const unsigned long long limit = jobs[sg.job_hi - 1].upos + jobs[sg.job_hi - 1].ulen;
    sg.end = sg.stop < limit ? sg.stop : limit;
    for (size_t c = seg_chunk[k]; c < seg_chunk[k + 1];) {
      if (cur == ~0ull) {            // (a guessed first record: the first chunk that found a boundary gives it)
        if (h_start[c] == ~0ull) { h_kept[c] = 0u; h_unm[c] = 0u; ++c; continue; }
        cur = h_start[c];
      }
      if (cur >= h_hi[c] || cur + 4 > h_limit[c]) { h_kept[c] = 0u; h_unm[c] = 0u; h_start[c] = ~0ull; ++c; continue; }   // no record starts in this chunk
      if (h_start[c] == cur) {
        if (h_bad[c]) { *bad_record = -2; return MIDAS_SNPS_ERR_BAD_LAYOUT; }
        if (sg.first == ~0ull) sg.first = cur;
        sg.n_records += h_kept[c];
        if (h_unm[c] && sg.first_unmapped == ~0ull) sg.first_unmapped = h_fu[c];
        sg.n_unmapped += h_unm[c];
        cur = h_end[c];
        ++c;
        continue;
      }
      if (++rounds > 4096) {
        if (err256) snprintf(err256, 256, "device decode: the record boundaries did not settle");
        return MIDAS_SNPS_ERR_UNSUPPORTED;
      }
      const long long one = (long long)c;
      DEC_TRY(hipMemcpyAsync(wp.start + c, &cur, 8, hipMemcpyHostToDevice, s));
      DEC_TRY(hipMemcpyAsync(d_list, &one, 8, hipMemcpyHostToDevice, s));
      DEC_TRY(launch_bam_walk(wp, d_list, 1, s));
      DEC_TRY(hipMemcpyAsync(&h_end[c], wp.end + c, 8, hipMemcpyDeviceToHost, s));
      DEC_TRY(hipMemcpyAsync(&h_kept[c], wp.kept + c, 4, hipMemcpyDeviceToHost, s));
      DEC_TRY(hipMemcpyAsync(&h_unm[c], wp.unmapped + c, 4, hipMemcpyDeviceToHost, s));
      DEC_TRY(hipMemcpyAsync(&h_fu[c], wp.first_unmapped + c, 8, hipMemcpyDeviceToHost, s));
      DEC_TRY(hipMemcpyAsync(&h_bad[c], wp.bad + c, 4, hipMemcpyDeviceToHost, s));
      DEC_TRY(hipStreamSynchronize(s));
      h_start[c] = cur;
    }
    if (cur != ~0ull) sg.end = cur;
  }
  if (rounds && trace) fprintf(stderr, "[device decode] %d chunk(s) walked again\n", rounds);
  unsigned long long n_rec = 0;
  for (long long c = 0; c < n_chunks; ++c) { h_base[(size_t)c] = n_rec; n_rec += h_kept[(size_t)c]; }
  if (n_chunks > 0) {
    DEC_TRY(hipMemcpyAsync(wp.start, h_start.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
    DEC_TRY(hipMemcpyAsync(wp.kept, h_kept.data(), (size_t)n_chunks * 4, hipMemcpyHostToDevice, s));
    DEC_TRY(hipMemcpyAsync(d_base, h_base.data(), (size_t)n_chunks * 8, hipMemcpyHostToDevice, s));
  }
  lap("stitch");
  const long long n = (long long)n_rec;
  const size_t n1 = (size_t)n + 1;
  unsigned long long* d_rec = reinterpret_cast<unsigned long long*>(take(n1 * 8));
  BamColumnsParams cp;
  cp.d = base; cp.rec_off = d_rec; cp.n = n; cp.n_ref = n_ref;
  cp.refid = reinterpret_cast<int32_t*>(take(n1 * 4)); cp.pos = reinterpret_cast<int32_t*>(take(n1 * 4));
  cp.nm = reinterpret_cast<int32_t*>(take(n1 * 4)); cp.l_seq = reinterpret_cast<int32_t*>(take(n1 * 4));
  cp.mapq = take(n1); cp.flag = reinterpret_cast<uint16_t*>(take(n1 * 2));
  cp.seq_off = reinterpret_cast<long long*>(take(n1 * 8)); cp.qual_off = reinterpret_cast<long long*>(take(n1 * 8));
  cp.cigar_off = reinterpret_cast<long long*>(take(n1 * 8));
  cp.span = extra ? reinterpret_cast<int32_t*>(take(n1 * 4)) : nullptr;
  cp.unit_off = payload == 2 ? reinterpret_cast<long long*>(take(n1 * 8)) : nullptr;
  cp.bad_record = reinterpret_cast<unsigned long long*>(take(8));
  long long* d_scan = reinterpret_cast<long long*>(take(bam_scan_scratch_bytes(n)));
  if (!d_scan) { if (err256) snprintf(err256, 256, "device decode: the arena is too small for the columns"); return MIDAS_SNPS_ERR_OUT_OF_MEMORY; }
  DEC_TRY(hipMemsetAsync(cp.bad_record, 0xFF, 8, s));
  DEC_TRY(launch_bam_offsets(wp, d_base, d_rec, s));
  DEC_TRY(launch_bam_columns(cp, d_scan, s));
  unsigned long long h_bad_record = ~0ull;
  long long ends[4] = {0, 0, 0, 0};
  DEC_TRY(hipMemcpyAsync(&h_bad_record, cp.bad_record, 8, hipMemcpyDeviceToHost, s));
  DEC_TRY(hipMemcpyAsync(&ends[0], cp.seq_off + n, 8, hipMemcpyDeviceToHost, s));
  DEC_TRY(hipMemcpyAsync(&ends[1], cp.qual_off + n, 8, hipMemcpyDeviceToHost, s));
  DEC_TRY(hipMemcpyAsync(&ends[2], cp.cigar_off + n, 8, hipMemcpyDeviceToHost, s));
  if (cp.unit_off) DEC_TRY(hipMemcpyAsync(&ends[3], cp.unit_off + n, 8, hipMemcpyDeviceToHost, s));
  DEC_TRY(hipStreamSynchronize(s));
  lap("offsets, columns, scans");
  if (h_bad_record != ~0ull) { *bad_record = (int64_t)h_bad_record; return MIDAS_SNPS_ERR_BAD_LAYOUT; }
  const int64_t sb = ends[0], qb = ends[1], nc = ends[2];
  if (payload == 2) {
    // ---- resident: the records in the pileup kernel's own layout, ONE copy of every record's [cigar][seq][qual] run; every
    // column stays where it was decoded and only refID comes down (the host groups the records by contig with it) -------------
    const unsigned long long units = (unsigned long long)ends[3];
    if (units > kMaxDirectPayloadUnits) {
      if (err256) snprintf(err256, 256, "device decode: %llu bytes of read payload exceed the 32 GiB the direct layout addresses", units * 8ull);
      return MIDAS_SNPS_ERR_UNSUPPORTED;
    }
    DirectRec* d_drec = reinterpret_cast<DirectRec*>(take((n1 + 1) * sizeof(DirectRec)));
    uint8_t* d_pay = take((size_t)units * 8 + 64);
    if (!d_pay) { if (err256) snprintf(err256, 256, "device decode: the arena is too small for the direct layout"); return MIDAS_SNPS_ERR_OUT_OF_MEMORY; }
    DEC_TRY(hipMemsetAsync(d_pay + (size_t)units * 8, 0, 64, s));       // (a lane's 16-byte loads may overhang the last read)
    BamDirectParams dp;
    dp.stream = base; dp.rec_off = d_rec; dp.n_records = n;
    dp.pos = cp.pos; dp.nm = cp.nm; dp.unit_off = cp.unit_off;
    dp.rec = d_drec; dp.payload = d_pay;
    DEC_TRY(launch_bam_direct(dp, ctx->prop.multiProcessorCount, s));
    const HostColumns hc = alloc(sink, n);
    if (!hc.refid) { DEC_TRY(hipStreamSynchronize(s)); return MIDAS_SNPS_OK; }      // (the caller reports its own out-of-memory)
    if (n > 0) {
      const int32_t dst = copy_to_host(ctx, hc.refid, cp.refid, (size_t)n * 4);
      if (dst != MIDAS_SNPS_OK) { if (err256) snprintf(err256, 256, "device decode: columns to host: %s", ctx->error_text().c_str()); return dst; }
    }
    DEC_TRY(hipStreamSynchronize(s));
    lap("direct layout, refID down");
    res->n_records = n; res->seq_bytes = sb; res->qual_bytes = qb; res->n_cigar = nc;
    ResidentReads& rr = res->resident;
    rr.rec = d_drec; rr.payload = d_pay; rr.refid = cp.refid; rr.pos = cp.pos; rr.nm = cp.nm; rr.l_seq = cp.l_seq; rr.mapq = cp.mapq; rr.flag = cp.flag;
    rr.seq_off = reinterpret_cast<int64_t*>(cp.seq_off); rr.qual_off = reinterpret_cast<int64_t*>(cp.qual_off);
    rr.cigar_off = reinterpret_cast<int64_t*>(cp.cigar_off); rr.unit_off = reinterpret_cast<int64_t*>(cp.unit_off);
    rr.stream = base; rr.rec_off = reinterpret_cast<const uint64_t*>(d_rec);
    rr.payload_units = (int64_t)units;
    res->dev_owner = new ArenaLoan{loan.pool, loan.p};       // everything lives in the arena: it stays lent until the handle is closed
    res->dev_free = arena_loan_free;
    loan.p = nullptr;
    return MIDAS_SNPS_OK;
  }
  // ---- SEQ / QUAL / CIGAR cut out of the stream, behind everything taken so far ---------------------------------------------
  uint8_t *d_seq = nullptr, *d_qual = nullptr, *d_cig = nullptr;
  struct Own { void* p = nullptr; ~Own() { if (p) (void)hipFree(p); } } own;
  if (payload) {
    d_seq = take((size_t)sb + 64);
    d_qual = take((size_t)qb + 64);
    d_cig = take((size_t)nc * 4 + 64);
    if (!d_cig) {       // (the scratch cannot hold them: a buffer of their own)
      const size_t need = up((size_t)sb + 64) + up((size_t)qb + 64) + up((size_t)nc * 4 + 64);
      DEC_TRY(hipMalloc(&own.p, need));
      d_seq = static_cast<uint8_t*>(own.p);
      d_qual = d_seq + up((size_t)sb + 64);
      d_cig = d_qual + up((size_t)qb + 64);
    }
    DEC_TRY(hipMemsetAsync(d_cig + (size_t)nc * 4, 0, 64, s));
    PayloadParams pp;
    pp.stream = base;
    pp.rec_off = d_rec;
    pp.n_records = n;
    pp.seq_off = cp.seq_off; pp.qual_off = cp.qual_off; pp.cigar_off = cp.cigar_off;
    pp.seq4 = d_seq; pp.qual = d_qual; pp.cigar = reinterpret_cast<uint32_t*>(d_cig);
    // (Launched on a stream of its own so that the small columns go down the link while it runs: measured, 24.5 ms against
    // 20.3 ms one behind the other -- the copy kernels and this one slow each other by more than the overlap wins.)
    DEC_TRY(launch_bam_payload(pp, ctx->prop.multiProcessorCount, s));
  }
  // ---- the small columns down ---------------------------------------------------------------------------------------------
  const HostColumns hc = alloc(sink, n);
  if (!hc.cigar_off) { DEC_TRY(hipStreamSynchronize(s)); return MIDAS_SNPS_OK; }      // (the caller reports its own out-of-memory)
  auto down = [&](void* dst, const void* src, size_t bytes) -> int32_t { return bytes && dst ? copy_to_host(ctx, dst, src, bytes) : MIDAS_SNPS_OK; };
  int32_t st = MIDAS_SNPS_OK;
  const std::pair<void*, std::pair<const void*, size_t>> cols[] = {
      {hc.refid, {cp.refid, (size_t)n * 4}}, {hc.pos, {cp.pos, (size_t)n * 4}}, {hc.nm, {cp.nm, (size_t)n * 4}}, {hc.l_seq, {cp.l_seq, (size_t)n * 4}},
      {hc.mapq, {cp.mapq, (size_t)n}}, {hc.flag, {cp.flag, (size_t)n * 2}}, {hc.seq_off, {cp.seq_off, n1 * 8}}, {hc.qual_off, {cp.qual_off, n1 * 8}},
      {hc.cigar_off, {cp.cigar_off, n1 * 8}}, {extra ? hc.span : nullptr, {cp.span, (size_t)n * 4}}, {extra ? hc.rec_off : nullptr, {d_rec, (size_t)n * 8}}};
  for (const auto& c : cols) {
    st = down(c.first, c.second.first, c.second.second);
    if (st != MIDAS_SNPS_OK) { if (err256) snprintf(err256, 256, "device decode: columns to host: %s", ctx->error_text().c_str()); return st; }
  }
  DEC_TRY(hipStreamSynchronize(s));
  lap("payload cut, small columns down");
#undef DEC_TRY
  res->n_records = n; res->seq_bytes = sb; res->qual_bytes = qb; res->n_cigar = nc;
  res->dev_seq = d_seq; res->dev_qual = d_qual; res->dev_cigar = d_cig;
  if (!payload) return MIDAS_SNPS_OK;       // (the arena goes back with `loan`)
  if (own.p) {        // the columns have a buffer of their own: the arena goes back now
    res->dev_owner = own.p;
    res->dev_free = device_free;
    own.p = nullptr;
  } else {            // the columns live in the arena: it stays lent until the handle is closed
    // (and with them the small columns the host has just been given: a batch made from those host arrays need not send them up
    // again -- midas_arena_pool::find_twin, midas_snps_batch_create)
    for (const auto& c : cols)
      if (c.first && c.second.first != static_cast<const void*>(d_rec)) loan.pool->add_twin(c.first, c.second.first, c.second.second, loan.p);
    res->dev_owner = new ArenaLoan{loan.pool, loan.p};
    res->dev_free = arena_loan_free;
    loan.p = nullptr;
  }
  return MIDAS_SNPS_OK;
}
}  // namespace

int32_t midas_bam_load_device(const char* path, midas_snps_ctx* ctx, midas_bam** out, int64_t* n_reads, int64_t* seq_bytes,
                              int64_t* qual_bytes, int64_t* n_cigar, char* err256) {
  if (!ctx || !path || !out) return MIDAS_SNPS_ERR_INVALID_ARG;
  const DeviceDecoder dec{ctx, device_decode_run};
  const int32_t st = bam_decode_on_device(path, &dec, out, n_reads, seq_bytes, qual_bytes, n_cigar, err256);
  if (st != MIDAS_SNPS_ERR_UNSUPPORTED) return st;
  return bam_load_device_host_walk(path, ctx, out, n_reads, seq_bytes, qual_bytes, n_cigar, err256);      // (boundaries not settled: the host walks)
}

int32_t midas_bam_load_resident(const char* path, midas_snps_ctx* ctx, midas_bam** out, int64_t* n_reads, int64_t* sum_l_seq, char* err256) {
  if (!ctx || !path || !out) return MIDAS_SNPS_ERR_INVALID_ARG;
  const DeviceDecoder dec{ctx, device_decode_run};
  int64_t qb = 0;
  const int32_t st = bam_decode_on_device(path, &dec, out, n_reads, nullptr, &qb, nullptr, err256, 2);
  if (sum_l_seq) *sum_l_seq = qb;       // (QUAL holds one byte per base)
  return st;
}

int32_t midas_bam_load_ranges_resident(midas_bam* bam, midas_snps_ctx* ctx, int32_t n_ranges, const int64_t* range_begin,
                                       const int64_t* range_end, int64_t* n_reads, int64_t* sum_l_seq, char* err256) {
  if (!bam || !ctx) return MIDAS_SNPS_ERR_INVALID_ARG;
  const DeviceDecoder dec{ctx, device_decode_run};
  int64_t qb = 0;
  const int32_t st = bam_load_ranges_on_device(bam, &dec, n_ranges, range_begin, range_end, n_reads, nullptr, &qb, nullptr, err256, 2);
  if (sum_l_seq) *sum_l_seq = qb;
  return st;
}

int32_t midas_bam_is_resident(const midas_bam* bam) { return bam_resident(bam, nullptr, nullptr, nullptr, nullptr) ? 1 : 0; }

// The fall-back of a resident handle: the three payload columns cut out of the inflated stream it still holds (a buffer of
// their own), the small columns brought down -- afterwards the handle answers midas_bam_columns as after midas_bam_load_device.
int32_t midas_bam_resident_to_columns(midas_bam* bam, midas_snps_ctx* ctx, int64_t* seq_bytes, int64_t* qual_bytes, int64_t* n_cigar, char* err256) {
  if (!bam || !ctx) return MIDAS_SNPS_ERR_INVALID_ARG;
  int64_t n = 0, sb = 0, qb = 0, nc = 0;
  const midas::ResidentReads* rr = bam_resident(bam, &n, &sb, &qb, &nc);
  if (!rr) { if (err256) snprintf(err256, 256, "the BAM handle holds no device-resident records"); return MIDAS_SNPS_ERR_INVALID_ARG; }
  if (seq_bytes) *seq_bytes = sb;
  if (qual_bytes) *qual_bytes = qb;
  if (n_cigar) *n_cigar = nc;
  if (midas_bam_payload_on_device(bam)) return MIDAS_SNPS_OK;         // (done before)
  auto hip_err = [&](hipError_t e, const char* what) {
    if (err256) snprintf(err256, 256, "resident BAM to columns: %s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? MIDAS_SNPS_ERR_OUT_OF_MEMORY : MIDAS_SNPS_ERR_HIP;
  };
#define RC_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_err(e__, #call); } while (0)
  std::lock_guard<std::mutex> g(ctx->device_mutex);
  RC_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t at_q = up((size_t)sb + 64), at_c = at_q + up((size_t)qb + 64), bytes = at_c + up((size_t)nc * 4 + 64);
  struct Own { void* p = nullptr; ~Own() { if (p) (void)hipFree(p); } } own;
  RC_TRY(hipMalloc(&own.p, bytes));
  uint8_t* base = static_cast<uint8_t*>(own.p);
  RC_TRY(hipMemsetAsync(base + sb, 0, 64, s));
  RC_TRY(hipMemsetAsync(base + at_q + qb, 0, 64, s));
  RC_TRY(hipMemsetAsync(base + at_c + (size_t)nc * 4, 0, 64, s));
  PayloadParams pp;
  pp.stream = rr->stream;
  pp.rec_off = reinterpret_cast<const unsigned long long*>(rr->rec_off);
  if (!rr->stream) { pp.stream = rr->payload; pp.drec = static_cast<const DirectRec*>(rr->rec); }      // (a streamed decode: out of the direct layout)
  pp.n_records = n;
  pp.seq_off = reinterpret_cast<const long long*>(rr->seq_off); pp.qual_off = reinterpret_cast<const long long*>(rr->qual_off);
  pp.cigar_off = reinterpret_cast<const long long*>(rr->cigar_off);
  pp.seq4 = base; pp.qual = base + at_q; pp.cigar = reinterpret_cast<uint32_t*>(base + at_c);
  RC_TRY(launch_bam_payload(pp, ctx->prop.multiProcessorCount, s));
  HostColumns hc{};
  if (!bam_alloc_host_columns(bam, n, &hc)) { if (err256) snprintf(err256, 256, "resident BAM to columns: out of host memory"); return MIDAS_SNPS_ERR_OUT_OF_MEMORY; }
  const size_t n1 = (size_t)n + 1;
  const std::pair<void*, std::pair<const void*, size_t>> cols[] = {
      {hc.pos, {rr->pos, (size_t)n * 4}}, {hc.nm, {rr->nm, (size_t)n * 4}}, {hc.l_seq, {rr->l_seq, (size_t)n * 4}}, {hc.mapq, {rr->mapq, (size_t)n}},
      {hc.flag, {rr->flag, (size_t)n * 2}}, {hc.seq_off, {rr->seq_off, n1 * 8}}, {hc.qual_off, {rr->qual_off, n1 * 8}}, {hc.cigar_off, {rr->cigar_off, n1 * 8}}};
  for (const auto& c : cols) {
    if (!c.second.second) continue;
    const int32_t st = copy_to_host(ctx, c.first, c.second.first, c.second.second);
    if (st != MIDAS_SNPS_OK) { if (err256) snprintf(err256, 256, "resident BAM to columns: %s", ctx->error_text().c_str()); return st; }
  }
  // (refID is in the handle's host memory already, and stays there: the caller holds views of it)
  RC_TRY(hipStreamSynchronize(s));
#undef RC_TRY
  bam_resident_became_columns(bam, base, base + at_q, base + at_c, own.p, device_free);
  own.p = nullptr;
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_copy_from_device(midas_snps_ctx* ctx, void* dst, const void* src, int64_t bytes) {
  if (!ctx || bytes < 0 || (bytes > 0 && (!dst || !src))) return MIDAS_SNPS_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(ctx->device_mutex);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return copy_to_host(ctx, dst, src, (size_t)bytes);
}

int32_t midas_bam_load_ranges_device(midas_bam* bam, midas_snps_ctx* ctx, int32_t n_ranges, const int64_t* range_begin,
                                     const int64_t* range_end, int64_t* n_reads, int64_t* seq_bytes, int64_t* qual_bytes,
                                     int64_t* n_cigar, char* err256) {
  if (!ctx) return MIDAS_SNPS_ERR_INVALID_ARG;
  const DeviceDecoder dec{ctx, device_decode_run};
  const int32_t st = bam_load_ranges_on_device(bam, &dec, n_ranges, range_begin, range_end, n_reads, seq_bytes, qual_bytes, n_cigar, err256);
  if (st != MIDAS_SNPS_ERR_UNSUPPORTED) return st;
  InflateUser iu{ctx};       // (boundaries not settled on the device: its inflater, the host's walk)
  const BlockInflater inf{&iu, device_inflate};
  return bam_load_ranges_with(bam, &inf, n_ranges, range_begin, range_end, n_reads, seq_bytes, qual_bytes, n_cigar, err256);
}

int32_t midas_bam_open_slice_device(const char* path, int32_t slice, int32_t n_slices, midas_snps_ctx* ctx, midas_bam** out, char* err256) {
  if (!ctx) return MIDAS_SNPS_ERR_INVALID_ARG;
  const DeviceDecoder dec{ctx, device_decode_run};
  return bam_open_slice_with(path, slice, n_slices, &dec, out, err256);
}

int32_t midas_snps_set_row_coder(midas_snps_ctx* ctx, int32_t coder) {
  if (!ctx || (coder != MIDAS_SNPS_ROWS_DEVICE && coder != MIDAS_SNPS_ROWS_HOST)) return MIDAS_SNPS_ERR_INVALID_ARG;
  ctx->row_coder = coder;
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_set_pad_rule(midas_snps_ctx* ctx, int32_t rule) {
  if (!ctx || (rule != MIDAS_SNPS_PAD_SPEC && rule != MIDAS_SNPS_PAD_PYSAM)) return MIDAS_SNPS_ERR_INVALID_ARG;
  ctx->pad_rule = rule;
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_set_default_path(midas_snps_ctx* ctx, int32_t path) {
  if (!ctx || (path != MIDAS_SNPS_PATH_AUTO && path != MIDAS_SNPS_PATH_DIRECT && path != MIDAS_SNPS_PATH_PACKED && path != MIDAS_SNPS_PATH_LONG))
    return MIDAS_SNPS_ERR_INVALID_ARG;
  ctx->default_path = path;
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_device_info(const midas_snps_ctx* ctx, char* name256, int32_t* n_cu, int64_t* hbm_bytes) {
  if (!ctx) return MIDAS_SNPS_ERR_INVALID_ARG;
  if (name256) snprintf(name256, 256, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
  if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
  return MIDAS_SNPS_OK;
}

void midas_snps_batch_destroy(midas_snps_batch* b) {
  if (!b) return;
  if (b->ctx) (void)hipSetDevice(b->ctx->device);
  if (b->resident) {      // (borrowed from the BAM handle, or biased into raw_owned)
    b->d_pos = nullptr; b->d_mapq = nullptr; b->d_nm = nullptr; b->d_lseq = nullptr; b->d_seq_off = nullptr; b->d_qual_off = nullptr;
    b->d_cigar_off = nullptr; b->d_seq4 = nullptr; b->d_qual = nullptr; b->d_cigar = nullptr; b->d_drec = nullptr; b->d_dpay = nullptr;
    (void)hipFree(b->raw_owned);
  }
  void* dev[] = {b->d_pos, b->d_mapq, b->d_nm, b->d_lseq, b->d_seq_off, b->d_qual_off, b->d_cigar_off, b->d_seq4, b->d_qual,
                 b->d_cigar, b->d_pack_reads, b->d_pack_recs, b->d_sort_tmp, b->d_rec, b->d_blob, b->d_ref, b->d_tiles,
                 b->d_contig_read_begin, b->d_contig_tile_base, b->d_contig_len, b->d_work, b->d_items, b->d_ticket, b->d_filt,
                 b->d_wg_begin, b->d_tile_split, b->d_trange, b->d_dfacts, b->d_block_contig, b->d_probe, b->d_drec, b->d_dpay, b->d_dunits,
                 b->d_outliers, b->d_n_outliers, b->d_tile_flag, b->d_chunk_ok,
                 b->d_orig, b->d_key, b->d_counts, b->d_allele};
  for (void* q : dev) (void)hipFree(q);
  if (b->h_tile_reads) (void)hipHostFree(b->h_tile_reads);
  for (auto& e : b->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : b->pev)
    if (e) (void)hipEventDestroy(e);
  delete b;
}

}  // extern "C"

namespace {

// Work items of the pileup kernel from the reads every tile will see: one per tile; a tile with very many reads (a
// coverage hot spot) is cut into parts that different workgroups process concurrently and merge with atomics -- otherwise
// one workgroup would walk it alone while the rest of the chip idles.  Whole tiles and parts are launched as two
// instantiations of the kernel.  Uploaded only when the plan differs from the one on the device.
int32_t plan_work_items(midas_snps_batch* b, const uint32_t* tile_reads) {
  midas_snps_ctx* ctx = b->ctx;
  // split only what would otherwise leave the chip idle: a tile holding more than twice a workgroup's fair share
  // of all reads (and at least 2048), cut into parts of about one fair share (at least 1024 reads)
  const size_t nt = (size_t)b->n_tiles;
  int64_t total_reads = 0;
  for (size_t t = 0; t < nt; ++t) total_reads += tile_reads[t];
  const int64_t fair = total_reads / (kWorkgroupsPerCU * (int64_t)ctx->prop.multiProcessorCount) + 1;
  int64_t split_reads = std::max<int64_t>(2048, 2 * fair), part_reads = std::max<int64_t>(1024, fair);
#ifdef MIDAS_SNPS_SPLIT_READS   // developer variants only (tools/build_variant.sh)
  split_reads = MIDAS_SNPS_SPLIT_READS;
  part_reads = split_reads > 1 ? split_reads / 2 : 1;
#endif
  std::vector<uint32_t> items;
  items.reserve(nt * 4 + 64);
  std::vector<std::pair<size_t, size_t>> zero_ranges;
  size_t n_split = 0;
  for (size_t t = 0; t < nt; ++t) {     // whole tiles first ...
    if ((int64_t)tile_reads[t] > split_reads) { ++n_split; continue; }
    items.push_back((uint32_t)t); items.push_back(0u); items.push_back(1u); items.push_back(0u);
  }
  const int64_t n_whole = (int64_t)(items.size() / 4);
  if (n_split) {
    std::vector<Tile> tiles(nt);
    HIP_TRY(ctx, hipMemcpy(tiles.data(), b->d_tiles, nt * sizeof(Tile), hipMemcpyDeviceToHost));
    for (size_t t = 0; t < nt; ++t) {     // ... then the parts of split tiles
      if ((int64_t)tile_reads[t] <= split_reads) continue;
      int64_t np = ((int64_t)tile_reads[t] + part_reads - 1) / part_reads;
      if (np > 4096) np = 4096;
      for (int64_t j = 0; j < np; ++j) { items.push_back((uint32_t)t); items.push_back((uint32_t)j); items.push_back((uint32_t)np); items.push_back(0u); }
      const size_t first = (size_t)tiles[t].site_base, cnt = (size_t)tiles[t].len;
      if (!zero_ranges.empty() && zero_ranges.back().first + zero_ranges.back().second == first) zero_ranges.back().second += cnt;
      else zero_ranges.emplace_back(first, cnt);
    }
  }
  b->n_whole_items = n_whole;
  b->n_items = (int64_t)(items.size() / 4);
  b->zero_ranges.swap(zero_ranges);
#ifdef MIDAS_SNPS_STREAM_KERNEL
  // Runs of whole tiles for the streaming kernel: one workgroup per CU, cuts where the cumulative cost (bytes of the
  // reads a tile will see + bytes of its rows) crosses a multiple of the fair share.  Split tiles cost nothing there.
  {
    const int n_wg = ctx->prop.multiProcessorCount;
    std::vector<uint8_t> split(nt, 0);
    std::vector<double> cost(nt);
    double total = 0;
    for (size_t t = 0; t < nt; ++t) {
      split[t] = (int64_t)tile_reads[t] > split_reads ? 1 : 0;
      cost[t] = split[t] ? 0.0 : 245.0 * (double)tile_reads[t] + 17.0 * (double)b->tile_len + 2000.0;
      total += cost[t];
    }
    std::vector<uint32_t> wg(n_wg + 1, (uint32_t)nt);
    wg[0] = 0;
    double acc = 0;
    int k = 1;
    for (size_t t = 0; t < nt && k < n_wg; ++t) {
      acc += cost[t];
      while (k < n_wg && acc >= total * k / n_wg) wg[k++] = (uint32_t)(t + 1);
    }
    if (wg != b->h_wg_begin) {
      if (!b->d_wg_begin) HIP_TRY(ctx, hipMalloc(&b->d_wg_begin, wg.size() * 4));
      HIP_TRY(ctx, hipMemcpyAsync(b->d_wg_begin, wg.data(), wg.size() * 4, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      b->h_wg_begin.swap(wg);
    }
    b->any_split = n_split > 0;
    if (n_split > 0 && split != b->h_tile_split) {
      if (!b->d_tile_split) HIP_TRY(ctx, hipMalloc(&b->d_tile_split, nt));
      HIP_TRY(ctx, hipMemcpyAsync(b->d_tile_split, split.data(), nt, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      b->h_tile_split.swap(split);
    }
  }
#endif
  // with no split tile the list is the identity and the kernel never reads it
  const bool need_upload = n_split > 0 && items != b->h_items;
  if (need_upload) {
    if (b->items_cap < items.size()) {
      (void)hipFree(b->d_items);
      b->d_items = nullptr;
      b->items_cap = 0;
      HIP_TRY(ctx, hipMalloc(&b->d_items, items.size() * 4));
      b->items_cap = items.size();
    }
    HIP_TRY(ctx, hipMemcpyAsync(b->d_items, items.data(), items.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    b->h_items.swap(items);
  }
  return MIDAS_SNPS_OK;
}

template <class T>
T* carve(uint8_t*& cur, size_t n) {
  T* r = reinterpret_cast<T*>(cur);
  cur += (n * sizeof(T) + 255) & ~(size_t)255;
  return r;
}
size_t carved(size_t n, size_t elem) { return (n * elem + 255) & ~(size_t)255; }

int32_t pack_status_to_error(midas_snps_ctx* ctx, unsigned long long st) {
  const long long rd = (long long)(st >> 8);
  char buf[256];
  ctx->err_read = rd;
  if ((st & 0xFF) == kPackUnsupported) {
    snprintf(buf, sizeof buf, "read %lld: l_seq > %d or n_cigar/NM > %d is not supported", rd, kMaxLSeq, kMaxField16);
    return fail(ctx, MIDAS_SNPS_ERR_UNSUPPORTED, buf);
  }
  snprintf(buf, sizeof buf, "read %lld: negative size or CSR offsets shorter than l_seq", rd);
  return fail(ctx, MIDAS_SNPS_ERR_BAD_LAYOUT, buf);
}

// The device packer over the resident raw reads, once the buffers exist: plan, keys, order, scatter on the context's stream.
// The per-tile read counts come back over a pinned buffer while the scatter kernel runs; the host derives the hot-spot
// plan from them (identical for identical input: nothing is uploaded then).
int32_t run_pack(midas_snps_batch* b, hipEvent_t* ev) {
  midas_snps_ctx* ctx = b->ctx;
  hipStream_t s = ctx->stream;
  if (ev) HIP_TRY(ctx, hipEventRecord(ev[0], s));
  HIP_TRY(ctx, launch_pack_plan(b->pk, b->d_sort_tmp, b->sort_tmp_bytes, s));
  HIP_TRY(ctx, launch_pack_keys(b->pk, s));
  HIP_TRY(ctx, launch_pack_order(b->pk, b->d_sort_tmp, b->sort_tmp_bytes, b->key_bits, s));
  if (b->n_tiles > 0)
    HIP_TRY(ctx, hipMemcpyAsync(b->h_tile_reads, b->pk.tile_reads, (size_t)b->n_tiles * 4, hipMemcpyDeviceToHost, s));
  if (ev) HIP_TRY(ctx, hipEventRecord(ev[1], s));
  HIP_TRY(ctx, launch_pack_scatter(b->pk, s));
  if (ev) HIP_TRY(ctx, hipEventRecord(ev[2], s));
  b->pack_count += 1;
  return MIDAS_SNPS_OK;
}


uint32_t* trange_begin(midas_snps_batch* b, int par) { return b->d_trange + (size_t)par * 2 * (b->n_tiles > 0 ? b->n_tiles : 1); }
uint32_t* trange_end(midas_snps_batch* b, int par) { return trange_begin(b, par) + (b->n_tiles > 0 ? b->n_tiles : 1); }

// SEQ / QUAL / CIGAR of a resident batch as the three columns the packed and the long path read: cut out of the BAM handle's
// inflated stream on first use (bam_payload_kernel over the batch's records), into one buffer of the batch's own; the
// pointers are biased by the first read's offsets, so the BAM's absolute CSR offsets index them.
int32_t ensure_raw_payload(midas_snps_batch* b) {
  if (!b->resident || b->raw_owned || b->n_reads <= 0) return MIDAS_SNPS_OK;
  midas_snps_ctx* ctx = b->ctx;
  hipStream_t s = ctx->stream;
  long long lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  const int64_t* offs[3] = {b->d_seq_off, b->d_qual_off, b->d_cigar_off};
  for (int k = 0; k < 3; ++k) {
    HIP_TRY(ctx, hipMemcpyAsync(&lo[k], offs[k], 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(&hi[k], offs[k] + b->n_reads, 8, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(ctx, hipStreamSynchronize(s));
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t sb = (size_t)(hi[0] - lo[0]), qb = (size_t)(hi[1] - lo[1]), cb = (size_t)(hi[2] - lo[2]) * 4;
  const size_t at_q = up(sb + 64), at_c = at_q + up(qb + 64), bytes = at_c + up(cb + 64);
  HIP_TRY(ctx, hipMalloc(&b->raw_owned, bytes));
  uint8_t* base = static_cast<uint8_t*>(b->raw_owned);
  HIP_TRY(ctx, hipMemsetAsync(base + sb, 0, 64, s));
  HIP_TRY(ctx, hipMemsetAsync(base + at_q + qb, 0, 64, s));
  HIP_TRY(ctx, hipMemsetAsync(base + at_c + cb, 0, 64, s));
  b->d_seq4 = base - lo[0];
  b->d_qual = base + at_q - lo[1];
  b->d_cigar = reinterpret_cast<uint32_t*>(base + at_c) - lo[2];
  PayloadParams pp;
  pp.stream = b->rr.stream;
  pp.rec_off = reinterpret_cast<const unsigned long long*>(b->rr.rec_off) + b->rr_first;
  if (!b->rr.stream) { pp.stream = b->rr.payload; pp.rec_off = nullptr; pp.drec = b->d_drec; }      // (a streamed decode: out of the direct layout)
  pp.n_records = b->n_reads;
  pp.seq_off = reinterpret_cast<const long long*>(b->d_seq_off); pp.qual_off = reinterpret_cast<const long long*>(b->d_qual_off);
  pp.cigar_off = reinterpret_cast<const long long*>(b->d_cigar_off);
  pp.seq4 = b->d_seq4; pp.qual = b->d_qual; pp.cigar = b->d_cigar;
  if (getenv("MIDAS_SNPS_TRACE"))
    fprintf(stderr, "[batch] raw columns cut out of %s: %lld reads, %zu + %zu + %zu bytes\n", pp.drec ? "the direct layout" : "the handle's inflated stream",
            (long long)b->n_reads, sb, qb, cb);
  HIP_TRY(ctx, launch_bam_payload(pp, ctx->prop.multiProcessorCount, s));
  return MIDAS_SNPS_OK;
}

void fill_direct_index(midas_snps_batch* b, DirectIndexParams* ip) {
  const int par = (int)(b->direct_run_count & 1);
  ip->pos = b->d_pos; ip->nm = b->d_nm; ip->l_seq = b->d_lseq;
  ip->seq_off = b->d_seq_off; ip->qual_off = b->d_qual_off; ip->cigar_off = b->d_cigar_off;
  ip->cigar = b->d_cigar;
  ip->rec = b->resident ? b->d_drec : nullptr;          // (a resident batch has no CIGAR column: the facts pass reads the ops in the payload)
  ip->payload = b->resident ? b->d_dpay : nullptr;
  ip->seq_bytes = b->seq_bytes; ip->qual_bytes = b->qual_bytes; ip->n_cigar = b->n_cigar;
  ip->n_reads = (int32_t)b->n_reads;
  ip->contig_read_begin = b->d_contig_read_begin; ip->contig_tile_base = b->d_contig_tile_base; ip->contig_len = b->d_contig_len;
  ip->n_contigs = b->n_contigs; ip->n_tiles = (int32_t)b->n_tiles; ip->tile_shift = kTileShift;
  ip->tbegin = trange_begin(b, par); ip->tend = trange_end(b, par);
  ip->tbegin_next = trange_begin(b, par ^ 1); ip->tend_next = trange_end(b, par ^ 1);
  ip->sorted = b->direct_sorted ? 1 : 0;
  ip->reach = b->direct_reach_ranges;
  ip->overhang = b->direct_overhang;
  ip->outliers = b->d_outliers; ip->n_outliers_listed = b->d_n_outliers; ip->outlier_cap = b->outlier_cap;
  ip->tile_flag = b->d_tile_flag;
  ip->facts = b->d_dfacts;
  ip->block_contig = b->d_block_contig;
  ip->stats = b->d_work ? work_stats(b) : nullptr; ip->err = b->d_work ? work_err(b) : nullptr;
  ip->n_stat_words = b->d_work ? b->n_species * MIDAS_STATS : 0;
}

// Buffers of the direct path, its facts pass (once: every read validated on the device -- the statuses of the packer --, the
// batch's totals, the longest reference span of a read) and one ranges pass: the numbers the choice of path rests on.
int32_t direct_prepare(midas_snps_batch* b) {
  midas_snps_ctx* ctx = b->ctx;
  hipStream_t s = ctx->stream;
  const size_t nt = (size_t)(b->n_tiles > 0 ? b->n_tiles : 1);
  HIP_TRY(ctx, hipMalloc(&b->d_trange, nt * 4 * 4));
  HIP_TRY(ctx, hipMalloc(&b->d_dfacts, sizeof(DirectFacts) * kDirectFactSlots));
  HIP_TRY(ctx, hipMalloc(&b->d_block_contig, (size_t)direct_index_blocks(b->n_reads) * sizeof(DirectBlockCursor)));
  HIP_TRY(ctx, hipMemsetAsync(b->d_block_contig, 0, (size_t)direct_index_blocks(b->n_reads) * sizeof(DirectBlockCursor), s));
  // tile bounds start clean (every pass resets the other parity's); counters at zero; status words at "no error"
  HIP_TRY(ctx, hipMemsetAsync(b->d_trange, 0, nt * 16, s));
  HIP_TRY(ctx, hipMemsetAsync(trange_begin(b, 0), 0xFF, nt * 4, s));
  HIP_TRY(ctx, hipMemsetAsync(trange_begin(b, 1), 0xFF, nt * 4, s));
  HIP_TRY(ctx, hipMemsetAsync(b->d_dfacts, 0, sizeof(DirectFacts) * kDirectFactSlots, s));
  for (int k = 0; k < kDirectFactSlots; ++k) HIP_TRY(ctx, hipMemsetAsync(&b->d_dfacts[k].status, 0xFF, 8, s));
  b->outlier_cap = (uint32_t)std::max<int64_t>(4096, b->n_reads / 64);
  HIP_TRY(ctx, hipMalloc(&b->d_outliers, (size_t)b->outlier_cap * sizeof(DirectOutlier)));
  HIP_TRY(ctx, hipMalloc(&b->d_n_outliers, 4));
  HIP_TRY(ctx, hipMalloc(&b->d_tile_flag, nt));
  HIP_TRY(ctx, hipMemsetAsync(b->d_n_outliers, 0, 4, s));
  HIP_TRY(ctx, hipMemsetAsync(b->d_tile_flag, 0, nt, s));
  DirectIndexParams ip;
  std::vector<DirectFacts> facts(kDirectFactSlots);
  unsigned long long status = kNoError, alg = 0, n_general = 0, n_long = 0, n_outliers = 0;
  uint32_t max_l = 0, max_span = 0, unsorted = 0, max_common = 0;
  auto facts_pass = [&]() -> int32_t {
    fill_direct_index(b, &ip);
    if (b->n_reads > 0) HIP_TRY(ctx, launch_direct_facts(ip, s));
    HIP_TRY(ctx, hipMemcpyAsync(facts.data(), b->d_dfacts, sizeof(DirectFacts) * kDirectFactSlots, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    status = kNoError; alg = 0; n_general = 0; n_long = 0; n_outliers = 0;
    max_l = 0; max_span = 0; unsorted = 0; max_common = 0;
    for (const DirectFacts& f : facts) {
      n_outliers += f.n_outliers;
      max_common = std::max(max_common, f.max_span_common);
      status = std::min(status, f.status);
      alg += f.alg_bytes;
      n_general += f.n_general;
      n_long += f.n_long;
      max_l = std::max(max_l, f.max_l);
      max_span = std::max(max_span, f.max_span);
      unsorted |= f.unsorted;
    }
    return MIDAS_SNPS_OK;
  };
  b->direct_overhang = kDirectOverhang;
  {
    const int32_t fst = facts_pass();
    if (fst != MIDAS_SNPS_OK) return fst;
  }
  if (status != kNoError) return pack_status_to_error(ctx, status);
  // Reads longer than the common overhang (250 bp reads), position-sorted: the batch takes the kernel's instantiation with the
  // long overhang -- and which reads are OUTLIERS is a question of that overhang: the facts pass again (once per batch, a few ms).
  static const int overhang_max = [] { const char* e = getenv("MIDAS_SNPS_OVERHANG_MAX"); return e ? atoi(e) : kDirectOverhangLong; }();
  if (kDirectChunkTiles > 1 && unsorted == 0 && n_long == 0 && max_l > (uint32_t)kDirectOverhang && max_l <= (uint32_t)kDirectOverhangLong &&
      overhang_max >= kDirectOverhangLong) {
    b->direct_overhang = kDirectOverhangLong;
    const size_t nt2 = (size_t)(b->n_tiles > 0 ? b->n_tiles : 1);
    HIP_TRY(ctx, hipMemsetAsync(b->d_block_contig, 0, (size_t)direct_index_blocks(b->n_reads) * sizeof(DirectBlockCursor), s));
    HIP_TRY(ctx, hipMemsetAsync(b->d_dfacts, 0, sizeof(DirectFacts) * kDirectFactSlots, s));
    for (int k = 0; k < kDirectFactSlots; ++k) HIP_TRY(ctx, hipMemsetAsync(&b->d_dfacts[k].status, 0xFF, 8, s));
    HIP_TRY(ctx, hipMemsetAsync(b->d_n_outliers, 0, 4, s));
    HIP_TRY(ctx, hipMemsetAsync(b->d_tile_flag, 0, nt2, s));
    const int32_t fst = facts_pass();
    if (fst != MIDAS_SNPS_OK) return fst;
    if (status != kNoError) return pack_status_to_error(ctx, status);
  }
  b->max_l_seq = (int32_t)max_l;
  b->direct_sorted = unsorted == 0;
  b->direct_general = (int64_t)n_general;
  b->direct_reach = (int32_t)std::max<uint32_t>(1u, std::max(max_l, max_span));
  b->direct_reach_ranges = b->direct_reach;
  b->n_outliers = (int64_t)n_outliers;
  // Chunks (a workgroup takes kDirectChunkTiles consecutive tiles and carries a tile's overhang on): position-sorted reads.  A
  // read that spans more than the overhang does not switch them off for the batch: the outliers are listed -- the ranges pass
  // then reaches back over the COMMON span only and direct_outliers_kernel adds them to the tiles they reach -- and only the
  // chunks they touch are dealt tile by tile.  Too many of them for the list (reads longer than the overhang, say): as before
  // round 6 -- the full reach, no chunks.
  b->direct_chunks = false;
  if (kDirectChunkTiles > 1 && unsorted == 0 && max_l <= (uint32_t)b->direct_overhang) {
    uint32_t listed = 0;
    HIP_TRY(ctx, hipMemcpy(&listed, b->d_n_outliers, 4, hipMemcpyDeviceToHost));
    if (listed <= b->outlier_cap) {
      b->direct_chunks = true;
      b->n_outliers_listed = listed;
      if (n_outliers > 0) {
        b->direct_reach_ranges = (int32_t)std::max<uint32_t>(1u, std::max(max_l, max_common));
        const int64_t n_chunked = (b->n_tiles - b->n_tiles / kDirectTailDiv) / kDirectChunkTiles * kDirectChunkTiles;
        std::vector<uint8_t> flag(nt), ok((size_t)std::max<int64_t>(1, n_chunked / kDirectChunkTiles), 1);
        HIP_TRY(ctx, hipMemcpy(flag.data(), b->d_tile_flag, nt, hipMemcpyDeviceToHost));
        for (int64_t t = 0; t < n_chunked; ++t)
          if (flag[(size_t)t]) ok[(size_t)(t / kDirectChunkTiles)] = 0;
        HIP_TRY(ctx, hipMalloc(&b->d_chunk_ok, ok.size()));
        HIP_TRY(ctx, hipMemcpy(b->d_chunk_ok, ok.data(), ok.size(), hipMemcpyHostToDevice));
      }
    }
  }
  b->alg_bytes = (int64_t)alg + 17 * b->n_sites;
  b->n_long = (int64_t)n_long;
  if (n_long > 0) {      // a read beyond the fast paths' limits: the whole batch takes the long path, nothing else is built
    b->path_auto = b->path = MIDAS_SNPS_PATH_LONG;
    return MIDAS_SNPS_OK;
  }
  b->direct_lane_bases = direct_lane_bases(b->max_l_seq);
  b->direct_lanes_per_read = b->max_l_seq <= b->direct_lane_bases ? 1 : (b->max_l_seq + b->direct_lane_bases - 1) / b->direct_lane_bases;
  // the direct layout: one 16-byte record per read and its CIGAR / SEQ / QUAL bytes as one run of the payload (every read
  // was validated above).  A resident batch has it already: the device decoder wrote it (bam_walk.hip, bam_direct_kernel).
  if (!b->resident) {
    const size_t n1 = (size_t)(b->n_reads > 0 ? b->n_reads : 1);
    const int nb = direct_index_blocks(b->n_reads);
    HIP_TRY(ctx, hipMalloc(&b->d_drec, (n1 + 1) * sizeof(DirectRec)));
    HIP_TRY(ctx, hipMemsetAsync(b->d_drec, 0, (n1 + 1) * sizeof(DirectRec), s));
    HIP_TRY(ctx, hipMalloc(&b->d_dunits, ((size_t)nb + 1) * 8));
    DirectLayoutParams lp;
    lp.pos = b->d_pos; lp.mapq = b->d_mapq; lp.nm = b->d_nm; lp.l_seq = b->d_lseq;
    lp.seq_off = b->d_seq_off; lp.qual_off = b->d_qual_off; lp.cigar_off = b->d_cigar_off;
    lp.seq4 = b->d_seq4; lp.qual = b->d_qual; lp.cigar = b->d_cigar;
    lp.n_reads = b->n_reads;
    lp.block_units = b->d_dunits;
    lp.rec = b->d_drec;
    lp.payload = nullptr;
    unsigned long long units = 0;
    hipEvent_t lev[2] = {nullptr, nullptr};       // (how long the layout takes on the device: reported beside the step it feeds)
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int k = 0; k < 2; ++k) if (e[k]) (void)hipEventDestroy(e[k]); } } ev_guard{lev};
    HIP_TRY(ctx, hipEventCreate(&lev[0]));
    HIP_TRY(ctx, hipEventCreate(&lev[1]));
    HIP_TRY(ctx, hipEventRecord(lev[0], s));
    if (b->n_reads > 0) {
      HIP_TRY(ctx, launch_direct_layout_sizes(lp, s));
      HIP_TRY(ctx, hipMemcpyAsync(&units, b->d_dunits + nb, 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(ctx, hipStreamSynchronize(s));
    }
    if (units > kMaxDirectPayloadUnits) {
      char buf[160];
      snprintf(buf, sizeof buf, "read payload of %llu bytes exceeds the 32 GiB a batch can address", units * 8ull);
      return fail(ctx, MIDAS_SNPS_ERR_UNSUPPORTED, buf);
    }
    b->direct_payload_bytes = (int64_t)(units * 8ull);
    constexpr size_t kPaySlack = 64;            // a lane's 16-byte loads may overhang the last read
    HIP_TRY(ctx, hipMalloc(&b->d_dpay, (size_t)b->direct_payload_bytes + kPaySlack));
    HIP_TRY(ctx, hipMemsetAsync(b->d_dpay + b->direct_payload_bytes, 0, kPaySlack, s));
    lp.payload = b->d_dpay;
    if (b->n_reads > 0) HIP_TRY(ctx, launch_direct_layout_fill(lp, s));
    HIP_TRY(ctx, hipEventRecord(lev[1], s));
    HIP_TRY(ctx, hipEventSynchronize(lev[1]));
    float lms = 0.f;
    if (hipEventElapsedTime(&lms, lev[0], lev[1]) == hipSuccess) b->layout_build_us = (int32_t)(lms * 1000.f + 0.5f);
    (void)hipGetLastError();
  }
  if (b->resident) b->direct_payload_bytes = b->rr.payload_units * 8;
  // the tile ranges once, to see how well the reads are ordered: a tile's stream holds every read between the first and the
  // last that can touch it
  fill_direct_index(b, &ip);
  HIP_TRY(ctx, launch_direct_ranges(ip, s));
  if (b->direct_reach_ranges != b->direct_reach) HIP_TRY(ctx, launch_direct_outliers(b->d_outliers, b->n_outliers_listed, ip.tbegin, s));
  std::vector<uint32_t> tb(nt), te(nt);
  HIP_TRY(ctx, hipMemcpyAsync(tb.data(), ip.tbegin, nt * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(te.data(), ip.tend, nt * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  b->direct_run_count += 1;
  int64_t stream = 0, worst = 0;
  for (size_t k = 0; k < (size_t)b->n_tiles; ++k) {
    const int64_t n0 = te[k] > tb[k] ? (int64_t)te[k] - (int64_t)tb[k] : 0;
    stream += n0;
    worst = std::max(worst, n0);
  }
  b->direct_stream_reads = stream;
  b->direct_max_tile_reads = worst;
  // Coordinate-sorted input makes the streams add up to the reads plus the few that straddle a tile border.  Much more than
  // that (unsorted input, or one read with a reference span of many tiles), or one tile holding far more than a workgroup's
  // fair share (a coverage hot spot: the packed path cuts such a tile into parts), and the packed path is the faster one.
  const int64_t fair = stream / (kWorkgroupsPerCU * (int64_t)ctx->prop.multiProcessorCount) + 1;
  const bool ordered = stream <= b->n_reads + b->n_reads / 2 + 4096;
  const bool hot = worst > std::max<int64_t>(2048, 2 * fair);
  b->path_auto = (ordered && !hot) ? MIDAS_SNPS_PATH_DIRECT : MIDAS_SNPS_PATH_PACKED;
  b->path = ctx->default_path == MIDAS_SNPS_PATH_AUTO ? b->path_auto : ctx->default_path;
#if MIDAS_SNPS_DEBUG_BITS & 256
  HIP_TRY(ctx, hipMalloc(&b->d_probe, (size_t)ctx->prop.multiProcessorCount * kWorkgroupsPerCU * (kPileupBlock / 64) * 8 * 8));
  HIP_TRY(ctx, hipMemset(b->d_probe, 0, (size_t)ctx->prop.multiProcessorCount * kWorkgroupsPerCU * (kPileupBlock / 64) * 8 * 8));
#endif
  return MIDAS_SNPS_OK;
}

// The packed path's device layout (layout.h), built on first use: plan, keys, order, scatter over the resident raw reads.
int32_t ensure_packed(midas_snps_batch* b) {
  if (b->packed_built) return MIDAS_SNPS_OK;
  midas_snps_ctx* ctx = b->ctx;
  if (b->n_long > 0) return fail(ctx, MIDAS_SNPS_ERR_UNSUPPORTED, "the batch holds reads beyond the packed layout's limits: it has the long path only");
  {
    const int32_t rst = ensure_raw_payload(b);
    if (rst != MIDAS_SNPS_OK) return rst;
  }
  hipStream_t s = ctx->stream;
  char ebuf[256] = {0};
  const int64_t n = b->n_reads, n_sites = b->n_sites;
  const size_t n1 = (size_t)(n > 0 ? n : 1);
  const size_t nt = (size_t)(b->n_tiles > 0 ? b->n_tiles : 1);
  const int64_t seq_bytes = b->seq_bytes, qual_bytes = b->qual_bytes, n_cigar = b->n_cigar;
  auto lap = [](const char*) {};
#define P_TRY(call)                                          \
  do {                                                       \
    hipError_t e__ = (call);                                 \
    if (e__ != hipSuccess) return hip_fail(ctx, e__, #call); \
  } while (0)
#define P_ST(expr)                          \
  do {                                      \
    int32_t s__ = (expr);                   \
    if (s__ != MIDAS_SNPS_OK) return s__;   \
  } while (0)
  // ---- packer, step 1: plan (validates every read on the device; the sizes come back) ----------------------------
  const size_t nbins = (size_t)b->n_tiles * kPackBinsPerTile + 1;
  {
    const size_t bytes = carved(n1, 1) + 2 * carved(n1 + 1, 4) + carved(kPackFactSlots, sizeof(PackFacts)) + carved(nbins, 4) + 2 * carved(nt, 4);
    P_TRY(hipMalloc(&b->d_pack_reads, bytes));
    uint8_t* cur = b->d_pack_reads;
    PackParams& k = b->pk;
    memset(&k, 0, sizeof k);
    k.nseg = carve<uint8_t>(cur, n1);
    k.cnt = carve<uint32_t>(cur, n1 + 1);
    k.first = carve<uint32_t>(cur, n1 + 1);
    k.facts = carve<PackFacts>(cur, kPackFactSlots);
    k.bin_start = carve<uint32_t>(cur, nbins);
    k.tile_extra = carve<uint32_t>(cur, nt);
    k.tile_reads = carve<uint32_t>(cur, nt);
    k.pos = b->d_pos; k.mapq = b->d_mapq; k.nm = b->d_nm; k.l_seq = b->d_lseq;
    k.seq_off = b->d_seq_off; k.qual_off = b->d_qual_off; k.cigar_off = b->d_cigar_off;
    k.seq4 = b->d_seq4; k.qual = b->d_qual; k.cigar = b->d_cigar;
    k.seq_bytes = seq_bytes; k.qual_bytes = qual_bytes; k.n_cigar = n_cigar;
    k.n_reads = (int32_t)n;
    k.contig_read_begin = b->d_contig_read_begin; k.contig_tile_base = b->d_contig_tile_base; k.contig_len = b->d_contig_len;
    k.n_contigs = b->n_contigs; k.n_tiles = (int32_t)b->n_tiles; k.tile_len = b->tile_len; k.tile_shift = kTileShift;
    k.pad_advances = b->pad_advances ? 1 : 0;
  }
  b->key_bits = pack_key_bits((int32_t)b->n_tiles);
  // the first scan needs scratch before the record count is known: size it for the reads, regrow below for the records
  b->sort_tmp_bytes = pack_sort_temp_bytes((int64_t)n1 + 1, b->key_bits);
  P_TRY(hipMalloc(&b->d_sort_tmp, b->sort_tmp_bytes));
  P_TRY(launch_pack_plan(b->pk, b->d_sort_tmp, b->sort_tmp_bytes, s));
  PackFacts facts;
  std::vector<PackFacts> slots(kPackFactSlots);
  auto fetch_facts = [&]() -> hipError_t {   // the workgroups' partial sums, one slot per cache line
    hipError_t e = hipMemcpyAsync(slots.data(), b->pk.facts, sizeof(PackFacts) * kPackFactSlots, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    facts = slots[0];
    for (int k = 1; k < kPackFactSlots; ++k) {
      facts.alg_bytes += slots[k].alg_bytes;
      facts.blob_bytes += slots[k].blob_bytes;
      facts.n_records += slots[k].n_records;
      facts.max_l = std::max(facts.max_l, slots[k].max_l);
    }
    return e;
  };
  P_TRY(fetch_facts());
  lap("pack: plan");
  if (facts.status != kNoError) P_ST(pack_status_to_error(ctx, facts.status));
  if (facts.n_records > 2000000000ull) {
    snprintf(ebuf, sizeof ebuf, "%llu device records exceed the supported range", facts.n_records);
    P_ST(fail(ctx, MIDAS_SNPS_ERR_UNSUPPORTED, ebuf));
  }
  const int64_t m = (int64_t)facts.n_records;
  b->n_records = m;
  b->max_l_seq = (int32_t)facts.max_l;
  b->alg_bytes = (int64_t)facts.alg_bytes + 17 * n_sites;
  b->lane_bases = lane_bases_for(b->max_l_seq);
  b->lanes_per_read = b->max_l_seq <= b->lane_bases ? 1 : (b->max_l_seq + b->lane_bases - 1) / b->lane_bases;

  // ---- packer, step 2: keys + sizes ------------------------------------------------------------------------------
  const size_t m1 = (size_t)(m > 0 ? m : 1);
  {
    const size_t bytes = 6 * carved(m1, 4) + 2 * carved(m1 + 1, 4) + carved(m1, 32);
    P_TRY(hipMalloc(&b->d_pack_recs, bytes));
    uint8_t* cur = b->d_pack_recs;
    PackParams& k = b->pk;
    k.sort_key = carve<uint32_t>(cur, m1);
    k.sort_val = carve<uint32_t>(cur, m1);
    k.key_sorted = carve<uint32_t>(cur, m1);
    k.val_sorted = carve<uint32_t>(cur, m1);
    k.bytes8 = carve<uint32_t>(cur, m1);
    k.dest = carve<uint32_t>(cur, m1);
    k.desc = carve<uint4>(cur, 2 * m1);
    k.bytes8_dev = carve<uint32_t>(cur, m1 + 1);
    k.off8 = carve<uint32_t>(cur, m1 + 1);
    k.n_records = (int32_t)m;
    k.lane_bases = b->lane_bases;
    k.lanes_per_read = b->lanes_per_read;
  }
  {
    const size_t need = pack_sort_temp_bytes((int64_t)std::max(m1, n1) + 1, b->key_bits);
    if (need > b->sort_tmp_bytes) {
      (void)hipFree(b->d_sort_tmp);
      b->d_sort_tmp = nullptr;
      P_TRY(hipMalloc(&b->d_sort_tmp, need));
      b->sort_tmp_bytes = need;
    }
  }
  P_TRY(launch_pack_keys(b->pk, s));
  P_TRY(fetch_facts());
  lap("pack: keys");
  b->blob_bytes = (int64_t)facts.blob_bytes;
  if (facts.blob_bytes / 8 > 0xFFFFFFFFull) {
    snprintf(ebuf, sizeof ebuf, "packed payload %llu bytes exceeds the 32 GiB a batch can address", facts.blob_bytes);
    P_ST(fail(ctx, MIDAS_SNPS_ERR_UNSUPPORTED, ebuf));
  }

  // ---- packer, steps 3 + 4: device order, then records + payload ---------------------------------------------------
  const size_t blob_alloc = (size_t)b->blob_bytes + 64;  // slack: the last lane's 16-byte load may overhang
  P_TRY(hipMalloc(&b->d_rec, (size_t)(m + 1) * sizeof(ReadRec)));  // + sentinel
  P_TRY(hipMalloc(&b->d_blob, blob_alloc));
  P_TRY(hipMalloc(&b->d_orig, m1 * 4));
  P_TRY(hipMalloc(&b->d_key, m1 * 4));
  P_TRY(hipMemsetAsync(b->d_blob + b->blob_bytes, 0, 64, s));
  b->pk.rec = b->d_rec; b->pk.blob = b->d_blob; b->pk.orig = b->d_orig; b->pk.key_out = b->d_key;
  P_TRY(hipHostMalloc(reinterpret_cast<void**>(&b->h_tile_reads), nt * 4, kHostAllocFlags));
  P_TRY(launch_pack_order(b->pk, b->d_sort_tmp, b->sort_tmp_bytes, b->key_bits, s));
  if (b->n_tiles > 0)
    P_TRY(hipMemcpyAsync(b->h_tile_reads, b->pk.tile_reads, (size_t)b->n_tiles * 4, hipMemcpyDeviceToHost, s));
  P_TRY(launch_pack_scatter(b->pk, s));
  P_TRY(fetch_facts());
  b->has_high_qual = slots[0].high_qual != 0;
  b->pack_count = 1;
  lap("pack: order + scatter");
  P_ST(plan_work_items(b, b->h_tile_reads));
  b->packed_built = true;
#undef P_TRY
#undef P_ST
  return MIDAS_SNPS_OK;
}
}  // namespace

extern "C" {

}  // extern "C"

namespace {
// midas_snps_batch_create (reads: the caller's arrays, uploaded) and midas_snps_batch_create_resident (rbam: a device-decoded
// BAM whose records [first, first + contigs->read_begin[n_contigs]) are the batch's reads, used where they lie)
int32_t batch_create_impl(midas_snps_ctx* ctx, const midas_snps_contigs* contigs, const midas_snps_reads* reads, const midas_bam* rbam,
                          int64_t first, midas_snps_batch** out_batch) {
  if (!ctx || !contigs || (!reads && !rbam) || !out_batch) return MIDAS_SNPS_ERR_INVALID_ARG;
  *out_batch = nullptr;
  ctx->clear_error();
  ctx->err_read = -1;
  char ebuf[256] = {0};
  int64_t n_sites = 0;
#ifdef MIDAS_SNPS_TRACE   // developer variants only: where batch_create spends its time
  const bool trace = true;
#else
  const bool trace = false;
#endif
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t_prev = now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto t = now();
    fprintf(stderr, "[batch_create] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  const midas::ResidentReads* rr = nullptr;
  int64_t r_total = 0, r_sb = 0, r_qb = 0, r_nc = 0;
  if (rbam) {
    rr = bam_resident(rbam, &r_total, &r_sb, &r_qb, &r_nc);
    if (!rr) return fail(ctx, MIDAS_SNPS_ERR_INVALID_ARG, "the BAM handle holds no device-resident records (midas_bam_load_resident)");
    if (contigs->n_contigs < 0 || (contigs->n_contigs > 0 && !contigs->read_begin)) return fail(ctx, MIDAS_SNPS_ERR_INVALID_ARG, "bad contig table header");
  }
  const int64_t n = rbam ? (contigs->n_contigs > 0 ? contigs->read_begin[contigs->n_contigs] : 0) : reads->n_reads;
  int32_t st = validate_contigs(contigs, n, &n_sites, ebuf);
  if (st != MIDAS_SNPS_OK) return fail(ctx, st, ebuf);
  if (n < 0 || n > 2000000000LL) {
    snprintf(ebuf, sizeof ebuf, "n_reads %lld out of range", (long long)n);
    return fail(ctx, n < 0 ? MIDAS_SNPS_ERR_INVALID_ARG : MIDAS_SNPS_ERR_UNSUPPORTED, ebuf);
  }
  if (rbam && (first < 0 || first > r_total || n > r_total - first)) {
    snprintf(ebuf, sizeof ebuf, "records [%lld, %lld) lie outside the BAM handle's %lld", (long long)first, (long long)(first + n), (long long)r_total);
    return fail(ctx, MIDAS_SNPS_ERR_INVALID_ARG, ebuf);
  }
  if (!rbam && n > 0 && (!reads->pos || !reads->mapq || !reads->nm || !reads->l_seq || !reads->seq_off || !reads->qual_off ||
                         !reads->cigar_off || !reads->seq4 || !reads->qual || !reads->cigar))
    return fail(ctx, MIDAS_SNPS_ERR_INVALID_ARG, "NULL array in midas_snps_reads");
  // the CSR offsets' last entries are the array sizes (the device checks every read against them)
  const int64_t seq_bytes = rbam ? r_sb : (n > 0 ? reads->seq_off[n] : 0), qual_bytes = rbam ? r_qb : (n > 0 ? reads->qual_off[n] : 0),
                n_cigar = rbam ? r_nc : (n > 0 ? reads->cigar_off[n] : 0);
  if (!rbam && (seq_bytes < 0 || qual_bytes < 0 || n_cigar < 0 || (n > 0 && (reads->seq_off[0] < 0 || reads->qual_off[0] < 0 || reads->cigar_off[0] < 0))))
    return fail(ctx, MIDAS_SNPS_ERR_BAD_LAYOUT, "negative CSR offset in midas_snps_reads");

  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  midas_snps_batch* b = new (std::nothrow) midas_snps_batch();
  if (!b) return fail(ctx, MIDAS_SNPS_ERR_OUT_OF_MEMORY, "host allocation failed");
  b->ctx = ctx;
  b->pad_advances = ctx->pad_rule == MIDAS_SNPS_PAD_PYSAM;
  b->n_reads = n;
  b->n_sites = n_sites;
  b->n_contigs = contigs->n_contigs;
  b->n_species = contigs->n_species;
  b->seq_bytes = seq_bytes;
  b->qual_bytes = qual_bytes;
  b->n_cigar = n_cigar;

#define B_TRY(call)                              \
  do {                                           \
    hipError_t e__ = (call);                     \
    if (e__ != hipSuccess) {                     \
      int32_t s__ = hip_fail(ctx, e__, #call);   \
      midas_snps_batch_destroy(b);               \
      return s__;                                \
    }                                            \
  } while (0)
#define B_ST(expr)                               \
  do {                                           \
    int32_t s__ = (expr);                        \
    if (s__ != MIDAS_SNPS_OK) {                  \
      midas_snps_batch_destroy(b);               \
      return s__;                                \
    }                                            \
  } while (0)

  // tile length: the LDS budget (4096 sites x 16 B); shorter tiles were measured and are slower
  b->tile_len = kTileSites;
  const int64_t tile_len = b->tile_len;
  // ---- tile table -------------------------------------------------------------------------
  std::vector<Tile> tiles;
  std::vector<int32_t> tile_base(contigs->n_contigs + 1, 0), clen(contigs->n_contigs), rbeg(contigs->n_contigs + 1, 0);
  {
    int64_t site = 0;
    for (int32_t c = 0; c < contigs->n_contigs; ++c) {
      const int64_t len = contigs->length[c];
      tile_base[c] = (int32_t)tiles.size();
      clen[c] = (int32_t)len;
      rbeg[c] = (int32_t)contigs->read_begin[c];
      for (int64_t x = 0; x < len; x += tile_len) {
        Tile t;
        t.contig = c;
        t.start = (int32_t)x;
        t.len = (int32_t)((len - x) < tile_len ? (len - x) : tile_len);
        t.species = contigs->species[c];
        t.site_base = site + x;
        t.contig_len = (int32_t)len;
        t.halo = (contigs->origin && contigs->origin[c] > 0) ? 1 : 0;
        tiles.push_back(t);
      }
      site += len;
    }
    tile_base[contigs->n_contigs] = (int32_t)tiles.size();
    rbeg[contigs->n_contigs] = (int32_t)n;
    b->h_contig_site.assign((size_t)contigs->n_contigs + 1, 0);
    for (int32_t c = 0; c < contigs->n_contigs; ++c) b->h_contig_site[(size_t)c + 1] = b->h_contig_site[(size_t)c] + contigs->length[c];
    if (contigs->origin) b->h_origin.assign(contigs->origin, contigs->origin + contigs->n_contigs);
  }
  b->n_tiles = (int64_t)tiles.size();
  const size_t nt = (size_t)(b->n_tiles > 0 ? b->n_tiles : 1);
  B_TRY(hipMalloc(&b->d_tiles, nt * sizeof(Tile)));
  if (b->n_tiles > 0) B_TRY(hipMemcpy(b->d_tiles, tiles.data(), tiles.size() * sizeof(Tile), hipMemcpyHostToDevice));
  const size_t nc1 = (size_t)contigs->n_contigs + 1;
  B_TRY(hipMalloc(&b->d_contig_read_begin, nc1 * 4));
  B_TRY(hipMalloc(&b->d_contig_tile_base, nc1 * 4));
  B_TRY(hipMalloc(&b->d_contig_len, nc1 * 4));
  B_TRY(hipMemcpy(b->d_contig_read_begin, rbeg.data(), nc1 * 4, hipMemcpyHostToDevice));
  B_TRY(hipMemcpy(b->d_contig_tile_base, tile_base.data(), nc1 * 4, hipMemcpyHostToDevice));
  if (contigs->n_contigs > 0)
    B_TRY(hipMemcpy(b->d_contig_len, clen.data(), (size_t)contigs->n_contigs * 4, hipMemcpyHostToDevice));
  lap("validate + tile table");

  // ---- upload the reads as they are: the packer runs on the device -----------------------------------------------
  // (64 bytes of slack behind the byte arrays: the packer's 16-byte loads may overhang the last read)
  const size_t n1 = (size_t)(n > 0 ? n : 1);
  if (rbam) {
    // ... or use them where the device decoder left them: the columns and the direct layout of the handle's records from `first` on
    b->resident = true;
    b->rr = *rr;
    b->rr_first = first;
    b->d_pos = rr->pos + first; b->d_mapq = rr->mapq + first; b->d_nm = rr->nm + first; b->d_lseq = rr->l_seq + first;
    b->d_seq_off = rr->seq_off + first; b->d_qual_off = rr->qual_off + first; b->d_cigar_off = rr->cigar_off + first;
    b->d_drec = static_cast<DirectRec*>(rr->rec) + first;
    b->d_dpay = rr->payload;
  } else {
  B_TRY(hipMalloc(&b->d_pos, n1 * 4));
  B_TRY(hipMalloc(&b->d_mapq, n1));
  B_TRY(hipMalloc(&b->d_nm, n1 * 4));
  B_TRY(hipMalloc(&b->d_lseq, n1 * 4));
  B_TRY(hipMalloc(&b->d_seq_off, (n1 + 1) * 8));
  B_TRY(hipMalloc(&b->d_qual_off, (n1 + 1) * 8));
  B_TRY(hipMalloc(&b->d_cigar_off, (n1 + 1) * 8));
  B_TRY(hipMalloc(&b->d_seq4, (size_t)seq_bytes + 64));
  B_TRY(hipMalloc(&b->d_qual, (size_t)qual_bytes + 64));
  B_TRY(hipMalloc(&b->d_cigar, ((size_t)n_cigar + 16) * 4));
  if (n > 0) {
    // (a column that a device decode of this context copied down a moment ago and still holds -- the caller passes the decoder's
    // own read-only buffers, or a run of them -- is copied where it lies instead of going up the link again)
    auto column_up = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
      const void* twin = ctx->arena->find_twin(src, bytes);
      return twin ? hipMemcpyAsync(dst, twin, bytes, hipMemcpyDeviceToDevice, s) : hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
    };
    B_TRY(column_up(b->d_pos, reads->pos, (size_t)n * 4));
    B_TRY(column_up(b->d_mapq, reads->mapq, (size_t)n));
    B_TRY(column_up(b->d_nm, reads->nm, (size_t)n * 4));
    B_TRY(column_up(b->d_lseq, reads->l_seq, (size_t)n * 4));
    B_TRY(column_up(b->d_seq_off, reads->seq_off, (size_t)(n + 1) * 8));
    B_TRY(column_up(b->d_qual_off, reads->qual_off, (size_t)(n + 1) * 8));
    B_TRY(column_up(b->d_cigar_off, reads->cigar_off, (size_t)(n + 1) * 8));
    // (seq4 / qual / cigar may be DEVICE pointers -- midas_bam_load_device leaves these columns there: the kind is inferred)
    if (seq_bytes > 0) B_TRY(hipMemcpyAsync(b->d_seq4, reads->seq4, (size_t)seq_bytes, hipMemcpyDefault, s));
    if (qual_bytes > 0) B_TRY(hipMemcpyAsync(b->d_qual, reads->qual, (size_t)qual_bytes, hipMemcpyDefault, s));
    if (n_cigar > 0) B_TRY(hipMemcpyAsync(b->d_cigar, reads->cigar, (size_t)n_cigar * 4, hipMemcpyDefault, s));
  }
  B_TRY(hipMemsetAsync(b->d_seq4 + seq_bytes, 0, 64, s));
  B_TRY(hipMemsetAsync(b->d_qual + qual_bytes, 0, 64, s));
  B_TRY(hipMemsetAsync(b->d_cigar + n_cigar, 0, 64, s));
  }
  lap("H2D raw reads");

  // ---- reference letters, workspace, outputs --------------------------------------------------------------------
  const size_t ns = (size_t)(n_sites > 0 ? n_sites : 1);
  B_TRY(hipMalloc(&b->d_ref, ns));
  if (n_sites > 0) B_TRY(hipMemcpyAsync(b->d_ref, contigs->ref, (size_t)n_sites, hipMemcpyHostToDevice, s));
  b->work_bytes = (((size_t)b->n_tiles * 48 + 15) & ~(size_t)15) + ((size_t)b->n_species * MIDAS_STATS + 1) * 8;
  B_TRY(hipMalloc(&b->d_work, b->work_bytes));
  // tile ranges start clean and every pileup workgroup re-zeroes its own entry; the counters and the
  // error word are reset by the index kernel at the start of each run: no per-run memsets
  B_TRY(hipMemsetAsync(b->d_work, 0, b->work_bytes, s));
  B_TRY(hipMalloc(&b->d_filt, sizeof(FilterTables)));
  B_TRY(hipMalloc(&b->d_counts, ns * 16));
  B_TRY(hipMalloc(&b->d_allele, ns));
  B_TRY(hipMalloc(&b->d_ticket, (nt + 2 * kSchedWords) * 4));     // + the two launches' dynamic-schedule words
  B_TRY(hipMemsetAsync(b->d_ticket, 0, (nt + 2 * kSchedWords) * 4, s));
  B_TRY(hipMalloc(&b->d_items, 4));
  b->items_cap = 1;
  lap("ref, workspace, outputs");

  // ---- the direct path's index pass, once: validates every read on the device, sizes the general reads' descriptors,
  // and tells how well the reads are ordered (the path is chosen from that) -------------------------------------------
  B_ST(direct_prepare(b));
  lap("direct index (sizing run)");
  if (b->path == MIDAS_SNPS_PATH_PACKED) B_ST(ensure_packed(b));
#undef B_TRY
#undef B_ST
  lap("packed layout");
  *out_batch = b;
  return MIDAS_SNPS_OK;
}
}  // namespace

extern "C" {
int32_t midas_snps_batch_create(midas_snps_ctx* ctx, const midas_snps_contigs* contigs,
                                const midas_snps_reads* reads, midas_snps_batch** out_batch) {
  if (!reads) return MIDAS_SNPS_ERR_INVALID_ARG;
  return batch_create_impl(ctx, contigs, reads, nullptr, 0, out_batch);
}

int32_t midas_snps_batch_create_resident(midas_snps_ctx* ctx, const midas_snps_contigs* contigs, const midas_bam* bam, int64_t first_read,
                                         midas_snps_batch** out_batch) {
  if (!bam) return MIDAS_SNPS_ERR_INVALID_ARG;
  return batch_create_impl(ctx, contigs, nullptr, bam, first_read, out_batch);
}

int32_t midas_snps_batch_select_path(midas_snps_batch* b, int32_t path) {
  if (!b || (path != MIDAS_SNPS_PATH_AUTO && path != MIDAS_SNPS_PATH_DIRECT && path != MIDAS_SNPS_PATH_PACKED && path != MIDAS_SNPS_PATH_LONG))
    return MIDAS_SNPS_ERR_INVALID_ARG;
  midas_snps_ctx* ctx = b->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (b->n_long > 0 && (path == MIDAS_SNPS_PATH_DIRECT || path == MIDAS_SNPS_PATH_PACKED)) {
    char buf[200];
    snprintf(buf, sizeof buf, "%lld read(s) with l_seq > %d or n_cigar / NM > %d: the batch has the long path only", (long long)b->n_long, kMaxLSeq, kMaxField16);
    return fail(ctx, MIDAS_SNPS_ERR_UNSUPPORTED, buf);
  }
  b->path = path == MIDAS_SNPS_PATH_AUTO ? b->path_auto : path;
  if (b->path == MIDAS_SNPS_PATH_PACKED) return ensure_packed(b);
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_batch_pack(midas_snps_batch* b) {
  if (!b) return MIDAS_SNPS_ERR_INVALID_ARG;
  midas_snps_ctx* ctx = b->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!b->packed_built) return ensure_packed(b);      // (building the layout IS a pack)
  hipEvent_t* ev = b->timing_slots > 0 ? &b->pev[(size_t)(b->timed_packs % b->timing_slots) * 3] : nullptr;
  int32_t st = run_pack(b, ev);
  if (st != MIDAS_SNPS_OK) return st;
  if (ev) b->timed_packs += 1;
  // the hot-spot plan needs the per-tile read counts on the host: one wait per pack (the scatter kernel is part of it)
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return plan_work_items(b, b->h_tile_reads);
}

int32_t midas_snps_batch_fetch_packed(midas_snps_batch* b, void* rec16, void* blob, uint32_t* orig_index, uint32_t* key,
                                      int64_t* out_n_records, int64_t* out_blob_bytes) {
  if (!b) return MIDAS_SNPS_ERR_INVALID_ARG;
  midas_snps_ctx* ctx = b->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    const int32_t pst = ensure_packed(b);
    if (pst != MIDAS_SNPS_OK) return pst;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (out_n_records) *out_n_records = b->n_records;
  if (out_blob_bytes) *out_blob_bytes = b->blob_bytes;
  if (rec16) HIP_TRY(ctx, hipMemcpy(rec16, b->d_rec, (size_t)(b->n_records + 1) * sizeof(ReadRec), hipMemcpyDeviceToHost));
  if (blob && b->blob_bytes > 0) HIP_TRY(ctx, hipMemcpy(blob, b->d_blob, (size_t)b->blob_bytes, hipMemcpyDeviceToHost));
  if (orig_index && b->n_records > 0) HIP_TRY(ctx, hipMemcpy(orig_index, b->d_orig, (size_t)b->n_records * 4, hipMemcpyDeviceToHost));
  if (key && b->n_records > 0) HIP_TRY(ctx, hipMemcpy(key, b->d_key, (size_t)b->n_records * 4, hipMemcpyDeviceToHost));
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_batch_enable_timing(midas_snps_batch* b, int32_t n_slots) {
  if (!b || n_slots < 0 || n_slots > 65536) return MIDAS_SNPS_ERR_INVALID_ARG;
  midas_snps_ctx* ctx = b->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (auto& e : b->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : b->pev)
    if (e) (void)hipEventDestroy(e);
  b->ev.assign((size_t)n_slots * 3, nullptr);
  b->pev.assign((size_t)n_slots * 3, nullptr);
  for (auto& e : b->ev) HIP_TRY(ctx, hipEventCreate(&e));
  for (auto& e : b->pev) HIP_TRY(ctx, hipEventCreate(&e));
  b->timing_slots = n_slots;
  b->timed_runs = 0;
  b->timed_packs = 0;
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_batch_time_pileup_only(midas_snps_batch* b, int32_t on) {
  if (!b) return MIDAS_SNPS_ERR_INVALID_ARG;
  b->timing_pileup_only = on != 0;
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_batch_run(midas_snps_batch* b, const midas_snps_thresholds* thr) {
  if (!b || !thr) return MIDAS_SNPS_ERR_INVALID_ARG;
  midas_snps_ctx* ctx = b->ctx;
  if (thr->reserved != 0) return fail(ctx, MIDAS_SNPS_ERR_INVALID_ARG, "thresholds.reserved must be 0");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // The packed payload keeps a quality in six bits (exact for every threshold <= 62).  A baseq above that on reads that hold
  // qualities above it (BAM allows 93; Illumina tops out in the forties) is served by the direct kernel, which compares the
  // BAM's own bytes and is exact on any read order -- this one run only, the batch's path stays what it is.
  int run_path = b->path;
  if (run_path == MIDAS_SNPS_PATH_PACKED && thr->baseq > kMaxPackedQual) {
    const int32_t pst = ensure_packed(b);
    if (pst != MIDAS_SNPS_OK) return pst;
    if (b->has_high_qual) run_path = MIDAS_SNPS_PATH_DIRECT;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  hipEvent_t* ev = b->timing_slots > 0 ? &b->ev[(size_t)(b->timed_runs % b->timing_slots) * 3] : nullptr;
  // an event record costs ~4 us of stream time (measured): a caller that only wants the pileup kernel's time leaves
  // the one in front of the index kernel out
  if (ev && !b->timing_pileup_only) HIP_TRY(ctx, hipEventRecord(ev[0], s));
  // rbinv / rend / stats zeroed, error word = "no error" (all ones)
  // keep_read's fp64 ratio tests as exact integer thresholds (rebuilt only when mapid / aln_cov change)
  if (!b->filt_valid || memcmp(&b->filt_mapid, &thr->mapid, 8) != 0 || memcmp(&b->filt_aln_cov, &thr->aln_cov, 8) != 0) {
    HIP_TRY(ctx, hipStreamSynchronize(s));  // a previous run may still be reading the tables
    build_filter_tables(thr->mapid, thr->aln_cov, b->max_l_seq, &b->h_filt);
    HIP_TRY(ctx, hipMemcpy(b->d_filt, &b->h_filt, sizeof(FilterTables), hipMemcpyHostToDevice));
    b->filt_mapid = thr->mapid;
    b->filt_aln_cov = thr->aln_cov;
    b->filt_valid = true;
    // (the long-overhang instantiation of the direct kernel keeps these tables in 16 bits: an identity threshold so negative that a
    // table entry lies below -32768 -- mapid < -11 000 -- sends such a batch through the common instantiation, tile by tile)
    b->filt_fits16 = true;
    for (int32_t a = 0; a <= b->max_l_seq && a <= kMaxLSeq; ++a) b->filt_fits16 = b->filt_fits16 && b->h_filt.min_match[a] >= -32768 && b->h_filt.min_align[a] <= 32767;
  }

  if (run_path == MIDAS_SNPS_PATH_LONG) {
    // ---- long path: one thread per read over the caller's arrays (reads beyond the fast paths' limits; any batch may ask for it) ----
    {
      const int32_t rst = ensure_raw_payload(b);
      if (rst != MIDAS_SNPS_OK) return rst;
    }
    if (ev) HIP_TRY(ctx, hipEventRecord(ev[1], s));
    LongParams lp;
    lp.pos = b->d_pos; lp.mapq = b->d_mapq; lp.nm = b->d_nm; lp.l_seq = b->d_lseq;
    lp.seq_off = b->d_seq_off; lp.qual_off = b->d_qual_off; lp.cigar_off = b->d_cigar_off;
    lp.seq4 = b->d_seq4; lp.qual = b->d_qual; lp.cigar = b->d_cigar;
    lp.n_reads = b->n_reads; lp.n_sites = b->n_sites;
    lp.contig_read_begin = b->d_contig_read_begin; lp.contig_tile_base = b->d_contig_tile_base;
    lp.n_contigs = b->n_contigs; lp.n_tiles = (int32_t)b->n_tiles;
    lp.tiles = b->d_tiles; lp.ref = b->d_ref;
    lp.out_counts = b->d_counts; lp.out_allele = b->d_allele;
    lp.stats = work_stats(b); lp.err = work_err(b);
    lp.n_stat_words = b->n_species * MIDAS_STATS;
    lp.baseq = thr->baseq; lp.mapq_min = thr->mapq; lp.readq = thr->readq; lp.pad_advances = b->pad_advances ? 1 : 0;
    lp.mapid = thr->mapid; lp.aln_cov = thr->aln_cov;
    HIP_TRY(ctx, launch_pileup_long(lp, s));
    if (ev) {
      HIP_TRY(ctx, hipEventRecord(ev[2], s));
      b->timed_runs += 1;
    }
    b->ran = true;
    b->run_count += 1;
    return MIDAS_SNPS_OK;
  }
  if (run_path == MIDAS_SNPS_PATH_DIRECT) {
    // ---- direct path: the tile ranges from the positions, then the pileup kernel over the raw arrays (one visit per read) ----
    DirectIndexParams dip;
    fill_direct_index(b, &dip);
    HIP_TRY(ctx, launch_direct_ranges(dip, s));
    if (b->direct_reach_ranges != b->direct_reach) HIP_TRY(ctx, launch_direct_outliers(b->d_outliers, b->n_outliers_listed, dip.tbegin, s));
    if (ev) HIP_TRY(ctx, hipEventRecord(ev[1], s));
    DirectParams dp;
    dp.rec = b->d_drec; dp.payload = b->d_dpay;
    dp.pos = b->d_pos; dp.mapq = b->d_mapq; dp.nm = b->d_nm; dp.l_seq = b->d_lseq;
    dp.seq_off = b->d_seq_off; dp.qual_off = b->d_qual_off; dp.cigar_off = b->d_cigar_off;
    dp.seq4 = b->d_seq4; dp.qual = b->d_qual; dp.cigar = b->d_cigar;
    dp.tbegin = dip.tbegin; dp.tend = dip.tend;
    dp.probe = b->d_probe;
    dp.ref = b->d_ref;
    dp.tiles = b->d_tiles;
    dp.filt = b->d_filt;
    dp.out_counts = b->d_counts; dp.out_allele = b->d_allele;
    dp.stats = work_stats(b); dp.err = work_err(b);
    dp.sched = b->d_ticket + b->n_tiles;
    dp.n_tiles = (int32_t)b->n_tiles; dp.n_reads = (int32_t)b->n_reads;
    const bool chunks = b->direct_chunks && (b->direct_overhang == kDirectOverhang || b->filt_fits16);
    dp.overhang = chunks ? b->direct_overhang : kDirectOverhang;      // (no chunks: nothing is carried, the common instantiation)
    dp.grid_blocks = ctx->prop.multiProcessorCount * kWorkgroupsPerCU;      // (the long overhang's instantiation keeps its tables in 16 bits: four workgroups a CU as well)
#ifdef MIDAS_SNPS_GRID_BLOCKS
    dp.grid_blocks = MIDAS_SNPS_GRID_BLOCKS;
#endif
    dp.lanes_per_read = b->direct_lanes_per_read;
    dp.reads_per_wave = 64 / b->direct_lanes_per_read;
    dp.table_len = b->max_l_seq + 1;
    dp.baseq = thr->baseq; dp.mapq_min = thr->mapq; dp.readq = thr->readq;
    dp.pad_advances = b->pad_advances ? 1 : 0;
    // chunks of consecutive tiles with a carried overhang (a read visited once): position-sorted reads none of which spans more
    // than the overhang; the last eighth of the tiles one by one, so that the workgroups finish together
    dp.chunk_tiles = 1; dp.n_chunked_tiles = 0;
    dp.chunk_ok = b->d_chunk_ok;
    if (chunks) {
      dp.chunk_tiles = kDirectChunkTiles;
      dp.n_chunked_tiles = (int32_t)((b->n_tiles - b->n_tiles / kDirectTailDiv) / kDirectChunkTiles * kDirectChunkTiles);
    }
    HIP_TRY(ctx, launch_pileup_direct(dp, b->direct_lane_bases, s));
    if (ev) {
      HIP_TRY(ctx, hipEventRecord(ev[2], s));
      b->timed_runs += 1;
    }
    b->ran = true;
    b->run_count += 1;
    b->direct_run_count += 1;
    return MIDAS_SNPS_OK;
  }
  {
    const int32_t pst = ensure_packed(b);
    if (pst != MIDAS_SNPS_OK) return pst;
  }

  IndexParams ip;
  ip.rec = b->d_rec;
  ip.blob = b->d_blob;
  ip.key = b->d_key;
  ip.tiles = b->d_tiles;
  const int par = (int)(b->run_count & 1);
  ip.rbinv = work_rbinv(b, par);
  ip.rend = work_rend(b, par);
  ip.rbinv_next = work_rbinv(b, par ^ 1);
  ip.rend_next = work_rend(b, par ^ 1);
  ip.n_tiles = (int32_t)b->n_tiles;
  ip.stats = work_stats(b);
  ip.err = work_err(b);
  ip.n_reads = (int32_t)b->n_records;
  ip.n_stat_words = b->n_species * MIDAS_STATS;
  ip.tile_len = b->tile_len;
  ip.lane_bases = b->lane_bases;
  HIP_TRY(ctx, launch_index_reads(ip, s));
  if (ev) HIP_TRY(ctx, hipEventRecord(ev[1], s));

  PileupParams pp;
  pp.rec = b->d_rec;
  pp.blob = b->d_blob;
  pp.ref = b->d_ref;
  pp.tiles = b->d_tiles;
  pp.rbinv = work_rbinv(b, par);
  pp.rend = work_rend(b, par);
  pp.out_counts = b->d_counts;
  pp.out_allele = b->d_allele;
  pp.stats = work_stats(b);
  pp.err = work_err(b);
  pp.n_tiles = (int32_t)b->n_tiles;
  pp.items = b->d_items;
  pp.split_ticket = b->d_ticket;
  pp.n_items = (int32_t)b->n_items;
  pp.n_whole_items = (int32_t)b->n_whole_items;
  pp.n_reads = (int32_t)b->n_records;
  pp.grid_blocks = ctx->prop.multiProcessorCount * kWorkgroupsPerCU;
#ifdef MIDAS_SNPS_GRID_BLOCKS   // developer variants only (tools/build_variant.sh)
  pp.grid_blocks = MIDAS_SNPS_GRID_BLOCKS;
#endif
  pp.lanes_per_read = b->lanes_per_read;
  pp.lane_bases = b->lane_bases;
  pp.reads_per_wave = 64 / b->lanes_per_read;
  pp.baseq = thr->baseq;
  pp.mapq = thr->mapq;
  pp.readq = thr->readq;
  pp.pad_advances = b->pad_advances ? 1 : 0;
  pp.filt = b->d_filt;
  pp.orig = b->d_orig;
  pp.table_len = b->max_l_seq + 1;
  for (const auto& zr : b->zero_ranges)   // split tiles accumulate with atomics: their counts start from zero
    HIP_TRY(ctx, hipMemsetAsync(b->d_counts + 4 * zr.first, 0, zr.second * 16, s));
  pp.wg_begin = b->d_wg_begin;
  pp.tile_split = b->any_split ? b->d_tile_split : nullptr;
  pp.n_stream_wgs = (int32_t)b->h_wg_begin.size() - 1;
#ifdef MIDAS_SNPS_STREAM_KERNEL   // developer variants only: whole tiles through the barrier-free streaming kernel
  // (pileup_stream.hip: correct, bit-exact, and 8-15 % slower than the barrier-phased kernel -- DESIGN.md section 3.3)
  HIP_TRY(ctx, launch_pileup_stream(pp, s));
  HIP_TRY(ctx, launch_pileup_tiles(pp, s, false, true));
#else
  HIP_TRY(ctx, launch_pileup_tiles(pp, s, true, true));
#endif
  if (ev) {
    HIP_TRY(ctx, hipEventRecord(ev[2], s));
    b->timed_runs += 1;
  }
  b->ran = true;
  b->run_count += 1;
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_batch_sync(midas_snps_batch* b) {
  if (!b) return MIDAS_SNPS_ERR_INVALID_ARG;
  midas_snps_ctx* ctx = b->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (!b->ran) return MIDAS_SNPS_OK;
  unsigned long long e = kNoError;
  HIP_TRY(ctx, hipMemcpy(&e, work_err(b), 8, hipMemcpyDeviceToHost));
  if (e != kNoError) {
    const int32_t kind = (int32_t)(e & 0xFF);
    ctx->err_read = (int64_t)(e >> 8);
    char buf[256];
    snprintf(buf, sizeof buf, "read %lld: %s", (long long)ctx->err_read, read_err_name(kind));
    return fail(ctx, kind, buf);
  }
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_batch_fetch(midas_snps_batch* b, uint32_t* out_counts, uint8_t* out_allele, int64_t* out_stats) {
  int32_t st = midas_snps_batch_sync(b);
  if (st != MIDAS_SNPS_OK) return st;
  midas_snps_ctx* ctx = b->ctx;
  if (!b->ran) return fail(ctx, MIDAS_SNPS_ERR_INVALID_ARG, "batch_fetch before batch_run");
  if (out_counts && b->n_sites > 0) {
    st = copy_to_host(ctx, out_counts, b->d_counts, (size_t)b->n_sites * 16);
    if (st != MIDAS_SNPS_OK) return st;
  }
  if (out_allele && b->n_sites > 0) {
    st = copy_to_host(ctx, out_allele, b->d_allele, (size_t)b->n_sites);
    if (st != MIDAS_SNPS_OK) return st;
  }
  if (out_stats && b->n_species > 0)
    HIP_TRY(ctx, hipMemcpy(out_stats, work_stats(b), (size_t)b->n_species * MIDAS_STATS * 8, hipMemcpyDeviceToHost));
  return MIDAS_SNPS_OK;
}

// Rows of a table straight from the batch's device results: the formatter's threads work on one slab of sites in the
// context's pinned ring while the next one crosses the link, so neither a host copy of the whole result nor the time of
// its transfer is ever paid on its own.
namespace {
// The ring: the context's two 32 MiB staging buffers cut into 16 slots of 4 MiB (15 gzip members each).  Many small
// slabs, not two large ones: a slot is reused only when every member of its previous slab is done, so with two slabs of
// ~125 members each and ~128 formatter threads the whole pool moved in lock step at the pace of its slowest thread
// (measured on a shared host: three rounds of ~86 ms).  With 240 members in flight a straggler holds up nobody.
constexpr int kFeedSlotsPerStage = 8;
constexpr size_t kFeedSlotBytes = midas_snps_ctx::kStageBytes / kFeedSlotsPerStage;
struct BatchFeed {
  midas_snps_batch* b;
  int64_t slab_sites;
  hipError_t err = hipSuccess;
};
bool batch_feed_fetch(void* user, int slot, int64_t src_lo, int64_t n, const uint8_t** allele, const uint32_t** counts) {
  BatchFeed* f = static_cast<BatchFeed*>(user);
  midas_snps_batch* b = f->b;
  midas_snps_ctx* ctx = b->ctx;
  hipStream_t s = ctx->stream;
  hipError_t e = hipSetDevice(ctx->device);     // (called from one of the library's worker threads)
  uint8_t* base = static_cast<uint8_t*>(ctx->stage[slot / kFeedSlotsPerStage]) + (size_t)(slot % kFeedSlotsPerStage) * kFeedSlotBytes;
  uint8_t* al = base + (size_t)f->slab_sites * 16;
  if (e == hipSuccess && n > 0) {
    void* mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, base, 0) == hipSuccess && mapped) {
      hipLaunchKernelGGL(copy_out_kernel, dim3(1024), dim3(256), 0, s, static_cast<copy_u32x4*>(mapped),
                         reinterpret_cast<const copy_u32x4*>(b->d_counts + 4 * src_lo), (size_t)n);
      e = hipGetLastError();
    } else {
      (void)hipGetLastError();
      e = hipMemcpyAsync(base, b->d_counts + 4 * src_lo, (size_t)n * 16, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(al, b->d_allele + src_lo, (size_t)n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  if (e != hipSuccess) { f->err = e; return false; }
  *allele = al;
  *counts = reinterpret_cast<const uint32_t*>(base);
  return true;
}
}  // namespace

namespace {
struct DeviceBuf {      // freed on every way out
  void* p = nullptr;
  ~DeviceBuf() { if (p) (void)hipFree(p); }
};

// The rows of the given contigs formatted and deflated on the device (rows_deflate.hip), framed and written by the host.
// Returns MIDAS_SNPS_OK with *done = false when the device coder declines a member (a contig id beyond its limit, an arena
// that turned out too small): the caller then takes the host's formatter for the whole part.
int32_t write_part_on_device(midas_snps_batch* b, const char* path, bool with_header, int32_t n_contigs, const int64_t* src,
                             const int64_t* n_sites, const int64_t* first, const char* const* ref_ids, int32_t gz_level,
                             int32_t threads, bool* done) {
  midas_snps_ctx* ctx = b->ctx;
  *done = false;
  // MIDAS_SNPS_TRACE=1: where the call spends its time, on stderr
  const bool trace = getenv("MIDAS_SNPS_TRACE") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t_last = now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto t = now();
    fprintf(stderr, "[rows on device] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  std::vector<RowsMember> members;
  std::vector<uint8_t> ids;
  for (int32_t k = 0; k < n_contigs; ++k) {
    const size_t id_off = ids.size(), id_len = strlen(ref_ids[k]);
    if (id_len > 192) return MIDAS_SNPS_OK;
    ids.insert(ids.end(), ref_ids[k], ref_ids[k] + id_len);
    for (int64_t lo = 0; lo < n_sites[k]; lo += kRowsPerMember) {
      RowsMember m;
      m.site0 = src[k] + lo;
      m.pos0 = first[k] + lo + 1;
      m.n_rows = (int32_t)std::min<int64_t>(kRowsPerMember, n_sites[k] - lo);
      m.id_off = (int32_t)id_off; m.id_len = (int32_t)id_len; m.pad = 0;
      members.push_back(m);
    }
  }
  const int64_t n_members = (int64_t)members.size();
  char err[256] = {0};
  if (n_members == 0) {
    const int32_t st = write_coded_members(path, with_header, gz_level, 0, nullptr, threads, err);
    if (st != MIDAS_SNPS_OK) return fail(ctx, st, err);
    *done = true;
    return MIDAS_SNPS_OK;
  }
  if (n_members > 0x7FFFFFFFll || ids.size() > 0x7FFFFFFFull) return MIDAS_SNPS_OK;
  int64_t rows = 0;
  for (const RowsMember& m : members) rows += m.n_rows;
  // the tables of a 20x genome take ~4 bytes a row; ten a row and a header's worth per member is room for any coverage seen
  // so far, and a member that does not fit sends the part to the host's formatter
  const unsigned long long arena_bytes = (unsigned long long)rows * 10ull + (unsigned long long)n_members * 1024ull + 4096ull;
  std::vector<RowsResult> results((size_t)n_members);
  struct HostBuf { uint8_t* p = nullptr; ~HostBuf() { free(p); } } host;     // (malloc: no zero fill)
  // Several host threads write one table each.  What they share is taken in turn, and as briefly as it can be: the context's
  // stream for the row kernel (device_mutex), then the pinned ring + the copy stream for the streams' way down (copy_mutex) --
  // table k's bytes cross the link while table k + 1 is formatted and deflated.  The allocations are nobody's turn.
  HIP_TRY(ctx, hipSetDevice(ctx->device));       // (the calling thread may never have talked to the device)
  struct PooledBuf {      // one buffer (the context keeps a few between tables): | arena | members | results | cursor | ids |
    midas_snps_ctx* ctx; void* p = nullptr; size_t bytes = 0;
    ~PooledBuf() { if (p) ctx->row_buffers.give(p, bytes); }
  } d_arena{ctx};
  auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t at_members = up256((size_t)arena_bytes), at_results = at_members + up256((size_t)n_members * sizeof(RowsMember)),
               at_cursor = at_results + up256((size_t)n_members * sizeof(RowsResult)), at_ids = at_cursor + 256;
  d_arena.p = ctx->row_buffers.take(at_ids + ids.size() + 16, &d_arena.bytes);
  if (!d_arena.p) return fail(ctx, MIDAS_SNPS_ERR_OUT_OF_MEMORY, "batch_write_part: out of device memory for a table's coded rows");
  uint8_t* const d_base = static_cast<uint8_t*>(d_arena.p);
  struct Part { void* p; } d_members{d_base + at_members}, d_results{d_base + at_results}, d_cursor{d_base + at_cursor}, d_ids{d_base + at_ids};
  lap("members + hipMalloc");
  unsigned long long used = 0;
  {
  std::lock_guard<std::mutex> device_part(ctx->device_mutex);
  hipStream_t s = ctx->stream;
  HIP_TRY(ctx, hipMemcpyAsync(d_members.p, members.data(), (size_t)n_members * sizeof(RowsMember), hipMemcpyHostToDevice, s));
  if (!ids.empty()) HIP_TRY(ctx, hipMemcpyAsync(d_ids.p, ids.data(), ids.size(), hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemsetAsync(d_arena.p, 0, (size_t)arena_bytes, s));
  HIP_TRY(ctx, hipMemsetAsync(d_cursor.p, 0, 8, s));
  RowsParams rp;
  rp.counts = b->d_counts; rp.allele = b->d_allele;
  rp.ids = static_cast<const uint8_t*>(d_ids.p);
  rp.members = static_cast<const RowsMember*>(d_members.p);
  rp.n_members = (int32_t)n_members;
  rp.arena = static_cast<uint8_t*>(d_arena.p); rp.arena_bytes = arena_bytes;
  rp.cursor = static_cast<unsigned long long*>(d_cursor.p);
  rp.results = static_cast<RowsResult*>(d_results.p);
  HIP_TRY(ctx, launch_rows_deflate(rp, ctx->prop.multiProcessorCount, s));
  HIP_TRY(ctx, hipMemcpyAsync(results.data(), d_results.p, (size_t)n_members * sizeof(RowsResult), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(&used, d_cursor.p, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  lap("its turn, memset + kernel");
  }
  for (const RowsResult& r : results)
    if (r.status != 0u) return MIDAS_SNPS_OK;
  if (used > arena_bytes) return MIDAS_SNPS_OK;
  host.p = static_cast<uint8_t*>(malloc((size_t)used + 16));
  if (!host.p) return fail(ctx, MIDAS_SNPS_ERR_OUT_OF_MEMORY, "batch_write_part: out of host memory");
  {
    hipStream_t cs = nullptr;
    {
      std::lock_guard<std::mutex> g(ctx->copy_mutex);
      if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->copy_stream = nullptr; }
      cs = ctx->copy_stream;
    }
    // (the row kernel is over -- the stream was waited for above: the copy depends on nothing that is still running)
    const int32_t cst = copy_to_host(ctx, host.p, d_arena.p, (size_t)used, cs);
    if (cst != MIDAS_SNPS_OK) return cst;
  }
  lap("streams to host");
  ctx->row_buffers.give(d_arena.p, d_arena.bytes);       // (before the file is written: the next table takes it)
  d_arena.p = nullptr;
  std::vector<CodedMember> coded((size_t)n_members);
  for (int64_t k = 0; k < n_members; ++k) {
    const RowsResult& r = results[(size_t)k];
    coded[(size_t)k] = CodedMember{host.p + r.off, r.n_bytes, r.crc, r.text_len, (uint32_t)members[(size_t)k].n_rows};
  }
  const int32_t st = write_coded_members(path, with_header, gz_level, n_members, coded.data(), threads, err);
  if (st != MIDAS_SNPS_OK) {
    std::lock_guard<std::mutex> g(ctx->device_mutex);
    return fail(ctx, st, err);
  }
  lap("frame + write");
  *done = true;
  return MIDAS_SNPS_OK;
}
}  // namespace

int32_t midas_snps_batch_write_part(midas_snps_batch* b, const char* path, int32_t with_header, int32_t n_contigs,
                                    const int32_t* contig_index, const char* const* ref_ids, int32_t gz_level, int32_t threads) {
  if (!b || !path || n_contigs < 0 || (n_contigs > 0 && (!contig_index || !ref_ids))) return MIDAS_SNPS_ERR_INVALID_ARG;
  midas_snps_ctx* ctx = b->ctx;
  int32_t st;
  {   // (several host threads may be writing one table each: see ctx_internal.h, device_mutex)
    std::lock_guard<std::mutex> g(ctx->device_mutex);
    st = midas_snps_batch_sync(b);
    if (st != MIDAS_SNPS_OK) return st;
    if (!b->ran) return fail(ctx, MIDAS_SNPS_ERR_INVALID_ARG, "batch_write_part before batch_run");
  }
  std::vector<int64_t> n_sites((size_t)n_contigs), src((size_t)n_contigs), first((size_t)n_contigs, 0);
  for (int32_t k = 0; k < n_contigs; ++k) {
    const int32_t c = contig_index[k];
    if (c < 0 || c >= b->n_contigs || !ref_ids[k]) return fail(ctx, MIDAS_SNPS_ERR_INVALID_ARG, "batch_write_part: contig index out of range");
    src[(size_t)k] = b->h_contig_site[(size_t)c];
    n_sites[(size_t)k] = b->h_contig_site[(size_t)c + 1] - b->h_contig_site[(size_t)c];
    if (!b->h_origin.empty()) first[(size_t)k] = b->h_origin[(size_t)c];     // a piece's rows carry the contig's positions
  }
  // gz levels 1-5 are the row coder's: it runs on the device unless the context was told otherwise
  if (gz_level >= 1 && gz_level <= 5 && ctx->row_coder == MIDAS_SNPS_ROWS_DEVICE) {
    bool done = false;
    st = write_part_on_device(b, path, with_header != 0, n_contigs, src.data(), n_sites.data(), first.data(), ref_ids, gz_level,
                              threads, &done);
    if (st != MIDAS_SNPS_OK || done) return st;
  }
  std::lock_guard<std::mutex> host_path(ctx->device_mutex);      // (the host's formatter owns the staging ring for the whole call)
  std::lock_guard<std::mutex> host_ring(ctx->copy_mutex);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  constexpr size_t kChunk = midas_snps_ctx::kStageBytes;
  ctx->stage_join();
  for (int k = 0; k < midas_snps_ctx::kStageSlots; ++k)
    if (!ctx->stage[k]) HIP_TRY(ctx, hipHostMalloc(&ctx->stage[k], kChunk, kHostAllocFlags));
  BatchFeed user{b, (int64_t)(kFeedSlotBytes / 17 / (size_t)kRowsPerMember) * kRowsPerMember};
  RowFeed feed{src.data(), user.slab_sites, midas_snps_ctx::kStageSlots * kFeedSlotsPerStage, &user, batch_feed_fetch};
  char err[256] = {0};
  st = write_rows_fed(path, with_header != 0, n_contigs, ref_ids, n_sites.data(), gz_level, threads, feed, err, first.data());
  if (user.err != hipSuccess) return hip_fail(ctx, user.err, "batch_write_part: results to host");
  if (st != MIDAS_SNPS_OK) return fail(ctx, st, err);
  return MIDAS_SNPS_OK;
}

#if MIDAS_SNPS_DEBUG_BITS & 256
// developer builds only: the direct pileup kernel's per-wave cycle counts of the last run (8 words per wave)
int32_t midas_snps_debug_probe(midas_snps_batch* b, unsigned long long* out, int64_t n_words) {
  if (!b || !out || !b->d_probe) return MIDAS_SNPS_ERR_INVALID_ARG;
  const int64_t have = (int64_t)b->ctx->prop.multiProcessorCount * kWorkgroupsPerCU * (kPileupBlock / 64) * 8;
  HIP_TRY(b->ctx, hipDeviceSynchronize());
  HIP_TRY(b->ctx, hipMemcpy(out, b->d_probe, (size_t)std::min(have, n_words) * 8, hipMemcpyDeviceToHost));
  return MIDAS_SNPS_OK;
}
#endif

int32_t midas_snps_batch_get_info(const midas_snps_batch* b, midas_snps_batch_info* out) {
  if (!b || !out) return MIDAS_SNPS_ERR_INVALID_ARG;
  out->n_reads = b->n_reads;
  out->n_sites = b->n_sites;
  out->n_tiles = b->n_tiles;
  out->packed_bytes = b->packed_built ? b->blob_bytes + b->n_records * (int64_t)sizeof(ReadRec) : 0;
  out->algorithmic_bytes = b->alg_bytes;
  out->tile_sites = b->tile_len;
  out->lanes_per_read = b->path == MIDAS_SNPS_PATH_DIRECT ? b->direct_lanes_per_read : b->lanes_per_read;
  out->n_work_items = b->path == MIDAS_SNPS_PATH_DIRECT ? b->n_tiles : b->n_items;
  out->path = b->path;
  out->path_auto = b->path_auto;
  out->lane_bases = b->path == MIDAS_SNPS_PATH_DIRECT ? b->direct_lane_bases : b->lane_bases;
  out->layout_build_us = b->layout_build_us;
  out->direct_general_reads = b->direct_general;
  out->direct_reach = b->direct_reach;
  out->direct_stream_reads = b->direct_stream_reads;
  out->direct_max_tile_reads = b->direct_max_tile_reads;
  out->direct_chunk_tiles = b->direct_chunks ? kDirectChunkTiles : 1;
  out->direct_overhang = b->direct_chunks ? b->direct_overhang : kDirectOverhang;
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_batch_timing(midas_snps_batch* b, int32_t slot, float out_ms[3]) {
  if (!b || !out_ms) return MIDAS_SNPS_ERR_INVALID_ARG;
  midas_snps_ctx* ctx = b->ctx;
  if (slot < 0 || slot >= b->timing_slots || slot >= b->timed_runs)
    return fail(ctx, MIDAS_SNPS_ERR_INVALID_ARG, "no timed run recorded in that slot");
  hipEvent_t* ev = &b->ev[(size_t)slot * 3];
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipEventSynchronize(ev[2]));
  HIP_TRY(ctx, hipEventElapsedTime(&out_ms[1], ev[1], ev[2]));
  if (b->timing_pileup_only) {
    out_ms[0] = 0.f;
    out_ms[2] = out_ms[1];
  } else {
    HIP_TRY(ctx, hipEventElapsedTime(&out_ms[0], ev[0], ev[1]));
    HIP_TRY(ctx, hipEventElapsedTime(&out_ms[2], ev[0], ev[2]));
  }
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_batch_pack_timing(midas_snps_batch* b, int32_t slot, float out_ms[2]) {
  if (!b || !out_ms) return MIDAS_SNPS_ERR_INVALID_ARG;
  midas_snps_ctx* ctx = b->ctx;
  if (slot < 0 || slot >= b->timing_slots || slot >= b->timed_packs)
    return fail(ctx, MIDAS_SNPS_ERR_INVALID_ARG, "no timed pack recorded in that slot");
  hipEvent_t* ev = &b->pev[(size_t)slot * 3];
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipEventSynchronize(ev[2]));
  HIP_TRY(ctx, hipEventElapsedTime(&out_ms[0], ev[0], ev[2]));
  HIP_TRY(ctx, hipEventElapsedTime(&out_ms[1], ev[1], ev[2]));
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_batch_stats_to_device(midas_snps_batch* b, void* dst) {
  if (!b || !dst) return MIDAS_SNPS_ERR_INVALID_ARG;
  midas_snps_ctx* ctx = b->ctx;
  if (!b->ran) return fail(ctx, MIDAS_SNPS_ERR_INVALID_ARG, "stats_to_device before batch_run");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (b->n_species > 0)
    HIP_TRY(ctx, hipMemcpyAsync(dst, work_stats(b), (size_t)b->n_species * MIDAS_STATS * 8, hipMemcpyDeviceToDevice,
                                ctx->stream));
  return MIDAS_SNPS_OK;
}

int32_t midas_snps_pileup(midas_snps_ctx* ctx, const midas_snps_thresholds* thr, const midas_snps_contigs* contigs,
                          const midas_snps_reads* reads, uint32_t* out_counts, uint8_t* out_allele,
                          int64_t* out_stats) {
  if (!ctx || !thr) return MIDAS_SNPS_ERR_INVALID_ARG;
  midas_snps_batch* b = nullptr;
  int32_t st = midas_snps_batch_create(ctx, contigs, reads, &b);
  if (st != MIDAS_SNPS_OK) return st;
  st = midas_snps_batch_run(b, thr);
  if (st == MIDAS_SNPS_OK) st = midas_snps_batch_fetch(b, out_counts, out_allele, out_stats);
  midas_snps_batch_destroy(b);
  return st;
}

}  // extern "C"
