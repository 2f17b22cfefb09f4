// gfx950 record walk of a BAM whose inflated stream lies in HBM (bgzf_inflate.hip put it there): what iterating
// pysam.AlignmentFile does behind midas/run/snps.py:186 -- find every alignment record, decode its fixed fields and its NM tag
// -- without the stream ever coming down to the host.
//
// The records form a chain: a record's block_size leads to the next one (SAM spec 4.2).  One thread following it from the
// first record would make ten million dependent trips to memory.  Here the stream -- the whole file's, a rank's slice of it, or
// the runs of blocks that hold a rank's contigs (segments) -- is cut into chunks of at most 32 KiB and
//   bam_walk_kernel     one thread per chunk: GUESSES a record boundary inside its chunk (the first offset from which four
//                       plausible records follow one another -- every fixed field in range, a printable NUL-terminated
//                       name, CIGAR op codes <= 8, the variable parts inside block_size) and walks the chain from there to
//                       the end of the chunk: where it started, where it ended, how many records with refID >= 0 it met.
//                       The host then stitches the chunks IN ORDER: a chunk's walk counts only if the chain that started at
//                       the true first record ended exactly on its guess -- then the guess was a true boundary and the walk
//                       is the one a single thread would have made.  A chunk whose guess the chain does not hit is walked
//                       again from where the chain stands (the same kernel, a list of chunks with forced starts): nothing is
//                       ever taken on plausibility alone.
//   bam_offsets_kernel  one thread per chunk, from the chunk's confirmed start: the offset of every kept record.
//   bam_columns_kernel  one thread per record: refID, pos, mapq, flag, l_seq, the lengths of its CIGAR / SEQ / QUAL (for the
//                       CSR offsets), NM from the aux fields (any integer width, as pysam's tags do), and the one malformation
//                       a record can have on its own: variable parts that overrun block_size (the lowest such record wins).
//   bam_scan_*          inclusive scans of the three length arrays, in place: the CSR offsets.
// The payload columns are then cut by bam_payload_kernel (bgzf_inflate.hip) as before.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace midas {

namespace {

typedef uint32_t u32_a1 __attribute__((aligned(1)));
typedef uint16_t u16_a1 __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t rd32(const uint8_t* p) { return *reinterpret_cast<const u32_a1*>(p); }
__device__ __forceinline__ uint32_t rd16(const uint8_t* p) { return *reinterpret_cast<const u16_a1*>(p); }

constexpr unsigned long long kNone = ~0ull;
constexpr int kChain = 4;

// Could an alignment record start at d + u?  (The host's plausible_bytes, hostio.cpp.)  *bs = its block_size.
__device__ bool plausible(const BamWalkParams& p, unsigned long long u, unsigned long long total, uint32_t* bs_out) {
  if (u + 36 > total) return false;
  const uint8_t* r = p.d + u;
  const uint32_t bs = rd32(r);
  if (bs < 32u || bs > (1u << 26)) return false;
  const int32_t refid = (int32_t)rd32(r + 4), pos = (int32_t)rd32(r + 8);
  const uint32_t lrn = r[12], n_cig = rd16(r + 16), l = rd32(r + 20);
  const int32_t nref = (int32_t)rd32(r + 24), npos = (int32_t)rd32(r + 28);
  if (refid < -1 || refid >= p.n_ref || nref < -1 || nref >= p.n_ref || pos < -1 || npos < -1) return false;
  if (refid >= 0 && (long long)pos > p.ref_lens[refid]) return false;
  if (lrn < 1u || l > (1u << 26)) return false;
  if (32ull + lrn + 4ull * n_cig + (l + 1u) / 2u + l > bs) return false;
  if (u + 4ull + 32ull + lrn + 4ull * n_cig > total) return false;
  const uint8_t* name = r + 36;
  if (name[lrn - 1u] != 0) return false;
  for (uint32_t k = 0; k + 1u < lrn; ++k)
    if (name[k] < 33 || name[k] > 126) return false;
  const uint8_t* cg = name + lrn;
  for (uint32_t k = 0; k < n_cig && k < 64u; ++k)
    if ((rd32(cg + 4u * k) & 15u) > 8u) return false;
  *bs_out = bs;
  return true;
}

__global__ __launch_bounds__(64) void bam_walk_kernel(BamWalkParams p, const long long* list, long long n_list) {
  const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
  long long c;
  if (list) {
    if (i >= n_list) return;
    c = list[i];
  } else {
    if (i >= p.n_chunks) return;
    c = i;
  }
  const unsigned long long lo = p.lo[c], hi = p.hi[c], total = p.limit[c], stop = p.stop[c] < hi ? p.stop[c] : hi;
  unsigned long long start = kNone;
  if (list || p.forced[c]) {
    start = p.start[c];                                 // given: a known record start, or where the chain stands
  } else {
    for (unsigned long long u = lo; u < hi && u + 36 <= total && start == kNone; ++u) {
      unsigned long long v = u;
      int ok = 0;
      while (ok < kChain) {
        if (v == total) break;                          // the segment ends on a record boundary: as good as a full chain
        uint32_t bs = 0;
        if (!plausible(p, v, total, &bs)) { ok = -1; break; }
        v += 4ull + bs;
        if (v > total) { ok = -1; break; }
        ++ok;
      }
      if (ok >= 0) start = u;
    }
    p.start[c] = start;
  }
  uint32_t kept = 0, unmapped = 0, bad = 0;
  unsigned long long first_unmapped = kNone;
  unsigned long long q = start;
  if (start != kNone) {
    while (q + 4 <= total && q < stop) {
      const uint32_t bs = rd32(p.d + q);
      if (bs < 32u || q + 4ull + bs > total) { bad = 1u; break; }
      if ((int32_t)rd32(p.d + q + 4) >= 0) {
        ++kept;
      } else {
        if (!unmapped) first_unmapped = q;
        ++unmapped;
      }
      q += 4ull + bs;
    }
  }
  p.end[c] = q;
  p.kept[c] = kept;
  p.unmapped[c] = unmapped;
  p.first_unmapped[c] = first_unmapped;
  p.bad[c] = bad;
}

__global__ __launch_bounds__(64) void bam_offsets_kernel(BamWalkParams p, const unsigned long long* base, unsigned long long* rec_off) {
  const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
  if (c >= p.n_chunks || p.kept[c] == 0u) return;
  const unsigned long long hi = p.hi[c], total = p.limit[c], stop = p.stop[c] < hi ? p.stop[c] : hi;
  unsigned long long q = p.start[c], j = base[c];
  while (q + 4 <= total && q < stop) {
    const uint32_t bs = rd32(p.d + q);
    if (bs < 32u || q + 4ull + bs > total) break;
    if ((int32_t)rd32(p.d + q + 4) >= 0) rec_off[j++] = q;
    q += 4ull + bs;
  }
}

// NM:i (any integer width) from the aux block, or -1 (hostio.cpp find_nm)
__device__ int32_t find_nm(const uint8_t* a, const uint8_t* end) {
  while (a + 3 <= end) {
    const char t0 = (char)a[0], t1 = (char)a[1], ty = (char)a[2];
    a += 3;
    unsigned long long sz = 0;
    switch (ty) {
      case 'A': case 'c': case 'C': sz = 1; break;
      case 's': case 'S': sz = 2; break;
      case 'i': case 'I': case 'f': sz = 4; break;
      case 'Z': case 'H': {
        const uint8_t* z = a;
        while (z < end && *z) ++z;
        if (z >= end) return -1;
        sz = (unsigned long long)(z - a) + 1ull;
        break;
      }
      case 'B': {
        if (a + 5 > end) return -1;
        const char st = (char)a[0];
        const unsigned long long cnt = rd32(a + 1);
        const unsigned long long es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
        sz = 5ull + cnt * es;
        break;
      }
      default: return -1;
    }
    if (sz > (unsigned long long)(end - a)) return -1;
    if (t0 == 'N' && t1 == 'M') {
      switch (ty) {
        case 'c': return (int8_t)a[0];
        case 'C': return a[0];
        case 's': return (int16_t)rd16(a);
        case 'S': return (int32_t)rd16(a);
        case 'i': return (int32_t)rd32(a);
        case 'I': { const uint32_t v = rd32(a); return v > 0x7FFFFFFFu ? 0x7FFFFFFF : (int32_t)v; }
        default: return -1;
      }
    }
    a += sz;
  }
  return -1;
}

__global__ __launch_bounds__(256) void bam_columns_kernel(BamColumnsParams p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) { p.seq_off[0] = p.base[0]; p.qual_off[0] = p.base[1]; p.cigar_off[0] = p.base[2]; if (p.unit_off) p.unit_off[0] = p.base[3]; }
  if (i >= p.n) return;
  const uint8_t* r = p.d + p.rec_off[i];
  const uint32_t bs = rd32(r);
  const uint32_t w3 = rd32(r + 12), w4 = rd32(r + 16), l = rd32(r + 20);
  const uint32_t lrn = w3 & 0xFFu, n_cig = w4 & 0xFFFFu;
  const unsigned long long body = 32ull + lrn + 4ull * n_cig + (l + 1u) / 2u + l;
  const int32_t refid = (int32_t)rd32(r + 4);
  p.refid[i] = refid;
  p.pos[i] = (int32_t)rd32(r + 8);
  p.mapq[i] = (uint8_t)(w3 >> 8);
  p.flag[i] = (uint16_t)(w4 >> 16);
  p.l_seq[i] = (int32_t)l;
  p.cigar_off[i + 1] = n_cig;
  p.seq_off[i + 1] = (l + 1u) / 2u;
  p.qual_off[i + 1] = l;
  if (p.unit_off) p.unit_off[i + 1] = (4ll * n_cig + (l + 1u) / 2u + (long long)l + 7ll) >> 3;     // (layout.h direct_payload_units, any l)
  int32_t nm = -1;
  // (a chained walk checks block_size only: a record whose refID names no reference is caught here, as the host walk's
  // plausible_record() catches it -- the host folds per-reference tables by it)
  if (body > bs || refid < 0 || refid >= p.n_ref) atomicMin(p.bad_record, (unsigned long long)i);
  else nm = find_nm(r + 4 + body, r + 4 + bs);
  p.nm[i] = nm;
  if (p.span) {       // reference span: the lengths of the ops that consume reference (M, D, N, =, X)
    long long span = 0;
    if (body <= bs) {
      const uint8_t* cg = r + 36 + lrn;
      for (uint32_t k = 0; k < n_cig; ++k) {
        const uint32_t v = rd32(cg + 4u * k), op = v & 15u;
        if (op == 0u || op == 2u || op == 3u || op == 7u || op == 8u) span += v >> 4;
      }
    }
    p.span[i] = (int32_t)(span > 0x7FFFFFFFll ? 0x7FFFFFFFll : span);
  }
}

// ---- inclusive scan of an int64 array, in place: tiles of 4096, three launches ---------------------------------------------
constexpr int kScanBlock = 256, kScanItems = 16, kScanTile = kScanBlock * kScanItems;

__device__ __forceinline__ long long block_inclusive(long long v, long long* lds, long long* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long x = v;
  for (int d = 1; d < 64; d <<= 1) {
    const long long o = __shfl_up(x, d);
    if (lane >= d) x += o;
  }
  if (lane == 63) lds[wave] = x;
  __syncthreads();
  long long base = 0;
  for (int w = 0; w < wave; ++w) base += lds[w];
  *total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return x + base;
}

__global__ __launch_bounds__(kScanBlock) void bam_scan_sums_kernel(const long long* a0, const long long* a1, const long long* a2, const long long* a3,
                                                                   long long n, long long* sums) {
  __shared__ long long lds[4];
  const long long* a = blockIdx.y == 0 ? a0 : (blockIdx.y == 1 ? a1 : (blockIdx.y == 2 ? a2 : a3));
  const long long lo = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanItems;
  long long s = 0;
  for (int k = 0; k < kScanItems; ++k) s += lo + k < n ? a[lo + k] : 0;
  long long total;
  (void)block_inclusive(s, lds, &total);
  if (threadIdx.x == 0) sums[(long long)blockIdx.y * gridDim.x + blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanBlock) void bam_scan_tiles_kernel(long long* sums, long long n_tiles) {     // one block per array: exclusive scan of its tile sums
  __shared__ long long lds[4];
  long long* s = sums + (long long)blockIdx.x * n_tiles;
  long long carry = 0;
  for (long long lo = 0; lo < n_tiles; lo += kScanBlock) {
    const long long i = lo + threadIdx.x;
    const long long v = i < n_tiles ? s[i] : 0;
    long long total;
    const long long inc = block_inclusive(v, lds, &total);
    if (i < n_tiles) s[i] = carry + inc - v;
    carry += total;
  }
}
__global__ __launch_bounds__(kScanBlock) void bam_scan_apply_kernel(long long* a0, long long* a1, long long* a2, long long* a3, long long n, const long long* sums) {
  __shared__ long long lds[4];
  long long* a = blockIdx.y == 0 ? a0 : (blockIdx.y == 1 ? a1 : (blockIdx.y == 2 ? a2 : a3));
  const long long lo = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanItems;
  long long v[kScanItems];
  long long s = 0;
  for (int k = 0; k < kScanItems; ++k) { v[k] = lo + k < n ? a[lo + k] : 0; s += v[k]; }
  long long total;
  const long long inc = block_inclusive(s, lds, &total);
  long long run = sums[(long long)blockIdx.y * gridDim.x + blockIdx.x] + inc - s;
  for (int k = 0; k < kScanItems; ++k) {
    run += v[k];
    if (lo + k < n) a[lo + k] = run;
  }
}

// ---- the records in the pileup kernel's own layout (layout.h DirectRec + payload), straight from the inflated stream ----------
// A record's CIGAR ops, 4-bit SEQ and QUAL are ONE run of its bytes (SAM spec 4.2: ... read_name, cigar, seq, qual, aux) -- and
// that run, in that order, is the direct layout's payload of a read.  HALF a wavefront per record copies it ONCE to the 8-byte
// unit the scan gave it (eight bytes a lane and load, any alignment; the tail to the next unit zeroed) and writes the read's
// 16-byte record.  What pysam.AlignmentFile + the iteration of midas/run/snps.py:186-199 hand to count_coverage, as the kernel
// that replaces count_coverage reads it: nothing is cut into columns and gathered back.
typedef unsigned long long u64_a1 __attribute__((aligned(1)));
__global__ __launch_bounds__(256) void bam_direct_kernel(BamDirectParams p) {
  const uint32_t lane = threadIdx.x & 63u, sub = lane >> 5, sl = lane & 31u;
  const long long slot = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + sub, n_slots = (long long)gridDim.x * 8;
  for (long long i = slot; i < p.n_records; i += n_slots) {
    const uint8_t* r = p.stream + p.rec_off[i];
    const uint32_t l_name = r[12];
    const uint32_t n_cig = rd16(r + 16);
    const uint32_t l = rd32(r + 20);
    const uint8_t* src = r + 36 + l_name;
    const unsigned long long used = 4ull * n_cig + (l + 1u) / 2u + (unsigned long long)l;
    const unsigned long long u0 = (unsigned long long)p.unit_off[i], room = ((unsigned long long)p.unit_off[i + 1] - u0) << 3;
    uint8_t* dst = p.payload + (u0 << 3);
    const unsigned long long whole = used & ~7ull;
    for (unsigned long long k = (unsigned long long)sl * 8ull; k < whole; k += 256ull) *reinterpret_cast<u64_a1*>(dst + k) = *reinterpret_cast<const u64_a1*>(src + k);
    if (sl < 8u && whole + sl < room) dst[whole + sl] = whole + sl < used ? src[whole + sl] : (uint8_t)0;      // the last unit: bytes, then zeros
    if (sl == 0u) {
      const int32_t nm = p.nm[i];
      DirectRec rec;
      rec.pos = p.pos[i];
      rec.l_nc = (l > 0xFFFFu ? 0xFFFFu : l) | ((n_cig > 0xFFFFu ? 0xFFFFu : n_cig) << 16);     // (beyond the fast paths' limits: the batch takes the long path)
      rec.nmq = (nm < 0 ? (uint32_t)kNmAbsent : (nm > kMaxField16 ? (uint32_t)kMaxField16 : (uint32_t)nm)) | ((uint32_t)r[13] << 16);
      rec.off8 = (uint32_t)u0;
      p.rec[i] = rec;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {       // the sentinel: where the payload ends
    DirectRec rec;
    rec.pos = 0; rec.l_nc = 0u; rec.nmq = 0u; rec.off8 = (uint32_t)p.unit_off[p.n_records];
    p.rec[p.n_records] = rec;
  }
}

}  // namespace

hipError_t launch_bam_direct(const BamDirectParams& p, int grid_blocks, hipStream_t s) {
  long long g = (p.n_records + 3) / 4;
  const long long cap = (long long)grid_blocks * 16;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(bam_direct_kernel, dim3((unsigned)g), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_bam_walk(const BamWalkParams& p, const long long* list, long long n_list, hipStream_t s) {
  const long long n = list ? n_list : p.n_chunks;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(bam_walk_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, p, list, n_list);
  return hipGetLastError();
}

hipError_t launch_bam_offsets(const BamWalkParams& p, const unsigned long long* base, unsigned long long* rec_off, hipStream_t s) {
  if (p.n_chunks <= 0) return hipSuccess;
  hipLaunchKernelGGL(bam_offsets_kernel, dim3((unsigned)((p.n_chunks + 63) / 64)), dim3(64), 0, s, p, base, rec_off);
  return hipGetLastError();
}

size_t bam_scan_scratch_bytes(long long n_records) {
  const long long tiles = (n_records + 1 + kScanTile - 1) / kScanTile;
  return (size_t)(4 * (tiles > 0 ? tiles : 1)) * sizeof(long long);
}

// the columns of n records and, by three scans, their CSR offsets (n + 1 entries each)
hipError_t launch_bam_columns(const BamColumnsParams& p, long long* scan_scratch, hipStream_t s) {
  hipLaunchKernelGGL(bam_columns_kernel, dim3((unsigned)((p.n + 256) / 256)), dim3(256), 0, s, p);
  const long long n1 = p.n + 1;
  const long long tiles = (n1 + kScanTile - 1) / kScanTile;
  const unsigned na = p.unit_off ? 4u : 3u;       // (the payload units of the direct layout are a fourth array)
  hipLaunchKernelGGL(bam_scan_sums_kernel, dim3((unsigned)tiles, na), dim3(kScanBlock), 0, s, p.seq_off, p.qual_off, p.cigar_off, p.unit_off, n1, scan_scratch);
  hipLaunchKernelGGL(bam_scan_tiles_kernel, dim3(na), dim3(kScanBlock), 0, s, scan_scratch, tiles);
  hipLaunchKernelGGL(bam_scan_apply_kernel, dim3((unsigned)tiles, na), dim3(kScanBlock), 0, s, p.seq_off, p.qual_off, p.cigar_off, p.unit_off, n1, scan_scratch);
  return hipGetLastError();
}

}  // namespace midas
