// BGZF blocks inflated on the device: one THREAD per block decodes, one WAVEFRONT per block copies the matches.
//
// A BAM is a chain of independent <= 64 KiB DEFLATE streams (BGZF, SAM spec 4.1); htslib inflates them one after the other,
// this library's host decoder on all cores -- and with the pileup at a millisecond and the rows coded on the device, that
// host inflate is two thirds of the stage.  DEFLATE decoding is a serial dependency chain inside a stream, but
// a 1.3 GB BAM holds 45 000 streams: each lane of a wavefront takes one and runs an ordinary table-driven inflater on it
// (RFC 1951: stored, fixed and dynamic blocks).  What makes that workable on a GPU is where the tables live: every lane's
// 9-bit literal/length and 7-bit distance look-up tables sit in LDS, interleaved [entry][lane], so the one dependent memory
// access per symbol is an LDS read, not a trip to HBM.  Codes longer than the look-up width (rare symbols) fall back to the
// canonical bit-by-bit walk over the per-length counts (the classic `puff` decoder), also from LDS.
// Two kernels.  The decoder writes the literals where they belong and only NOTES the matches (position, length, distance):
// Huffman decoding never looks at the output, and a lane that stopped to copy a match -- a round trip to memory for bytes it
// wrote a moment ago -- would hold up the other 63 lanes of its wavefront on almost every symbol (measured: 2.3 us per symbol
// that way).  The resolver then takes one wavefront per block through the block's matches in order, 64 bytes of a match per
// step, with a fence only in front of a match whose source was written since the last one.
// Replaces the inflate inside `pysam.AlignmentFile(...)` of midas/run/snps.py:186 (htslib's bgzf.c); bounds-checked against
// both buffers at every step: corrupt input yields a status, never a fault.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace midas {
namespace {

constexpr int kLanes = 64;
constexpr int kLlBits = 9, kDBits = 7;
// u16 arrays, per lane, interleaved [index][lane]
constexpr int kLutLl = 0;                       // 512: (symbol << 4) | length, 0 = not in the table
constexpr int kLutD = kLutLl + (1 << kLlBits);  // 128 (also the code-length code's table while a dynamic header is read)
constexpr int kSymLl = kLutD + (1 << kDBits);   // 288: symbols sorted by (length, symbol)
constexpr int kSymD = kSymLl + 288;             // 32
constexpr int kCntLl = kSymD + 32;              // 16: codes per length
constexpr int kCntD = kCntLl + 16;              // 16
constexpr int kU16 = kCntD + 16;                // 992
constexpr int kLens = 320;                      // u8: code lengths while a table is built

enum : uint32_t { kOk = 0, kBadBlockType = 1, kBadStored = 2, kBadCodeLengths = 3, kBadSymbol = 4, kBadDistance = 5,
                  kOutputOverrun = 6, kInputOverrun = 7, kShortOutput = 8 };

// (pointers that SAY they point into LDS: through a generic pointer every table look-up would be a FLAT access, which waits
// for the thread's outstanding global stores -- one store acknowledgement per symbol)
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
struct Lds {
  lds_u16* s16;
  lds_u8* s8;
  int lane;
  __device__ __forceinline__ lds_u16& h(int i) const { return s16[i * kLanes + lane]; }
  __device__ __forceinline__ lds_u8& b(int i) const { return s8[i * kLanes + lane]; }
};

// The compressed stream, least significant bit first (RFC 1951 3.1.1), read in aligned 32-bit words.
struct BitIn {
  const uint32_t* w;       // next aligned word to load
  const uint8_t* end;      // first byte behind the stream
  unsigned long long buf;
  uint32_t ahead;          // the word after the ones in buf, loaded one refill early: nobody waits for a load just issued
  int n;                   // valid bits in buf
  long long budget;        // bits of the stream not yet moved into buf (negative: the stream has been overrun)
  __device__ __forceinline__ void open(const uint8_t* p, size_t len) {
    end = p + len;
    buf = 0; n = 0;
    budget = (long long)len * 8;
    while ((reinterpret_cast<uintptr_t>(p) & 3u) && n < 32) {      // bytes up to the first aligned word
      buf |= (unsigned long long)(*p++) << n;
      n += 8;
    }
    budget -= n;
    w = reinterpret_cast<const uint32_t*>(p);
    ahead = *w++;
  }
  // at least 32 valid bits behind this (the buffers have 8 bytes of slack behind the last stream)
  __device__ __forceinline__ void refill() {
    if (n <= 32) {
      buf |= (unsigned long long)ahead << n;
      n += 32;
      budget -= 32;
      ahead = budget > -64 ? *w : 0u;      // (never more than a few words behind the stream's end)
      ++w;
    }
  }
  __device__ __forceinline__ uint32_t peek(int k) const { return (uint32_t)buf & ((1u << k) - 1u); }
  __device__ __forceinline__ void skip(int k) { buf >>= k; n -= k; }
  __device__ __forceinline__ uint32_t take(int k) { const uint32_t v = peek(k); skip(k); return v; }
  // bits consumed beyond the stream's end?
  __device__ __forceinline__ bool overrun() const { return budget + n < 0; }
};

// Canonical Huffman tables of one alphabet from the code lengths in L.b(0..n): counts per length, symbols sorted by
// (length, symbol), and the look-up table over the first `bits` bits.  False: over-subscribed or incomplete (an incomplete
// code is allowed only as the single-code case, as zlib allows it).
__device__ bool build_tables(const Lds& L, int n, int cnt_at, int sym_at, int lut_at, int bits) {
  for (int l = 0; l < 16; ++l) L.h(cnt_at + l) = 0;
  for (int s = 0; s < n; ++s) L.h(cnt_at + L.b(s)) += 1;
  for (int i = 0; i < (1 << bits); ++i) L.h(lut_at + i) = 0;
  if (L.h(cnt_at) == n) return true;              // no codes at all: legal for the distance alphabet of a literal-only block
  int left = 1;
  for (int l = 1; l < 16; ++l) {
    left <<= 1;
    left -= (int)L.h(cnt_at + l);
    if (left < 0) return false;                   // over-subscribed
  }
  if (left > 0 && !(n - (int)L.h(cnt_at) == 1 && L.h(cnt_at + 1) == 1)) return false;     // incomplete
  // offsets of every length in the sorted symbol list (kept in registers: 15 small numbers would not pay for LDS round trips)
  uint32_t offs[16];
  offs[1] = 0;
#pragma unroll
  for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + L.h(cnt_at + l);
  // first code of every length (RFC 1951 3.2.2)
  uint32_t next[16];
  uint32_t code = 0;
  next[0] = 0;
#pragma unroll
  for (int l = 1; l < 16; ++l) { code = (code + (l > 1 ? (uint32_t)L.h(cnt_at + l - 1) : 0u)) << 1; next[l] = code; }
  for (int s = 0; s < n; ++s) {
    const int l = L.b(s);
    if (!l) continue;
    uint32_t o = 0, c = 0;
#pragma unroll
    for (int k = 1; k < 16; ++k) { if (k == l) { o = offs[k]; offs[k] = o + 1; c = next[k]; next[k] = c + 1; } }
    L.h(sym_at + (int)o) = (uint16_t)s;
    if (l <= bits) {
      const uint32_t r = __brev(c) >> (32 - l);
      for (uint32_t k = r; k < (1u << bits); k += 1u << l) L.h(lut_at + (int)k) = (uint16_t)((s << 4) | l);
    }
  }
  return true;
}

// One symbol: the look-up table, else the canonical walk length by length.  Returns -1 on a code no symbol has.
__device__ __forceinline__ int decode(const Lds& L, BitIn& in, int cnt_at, int sym_at, int lut_at, int bits) {
  const uint32_t e = L.h(lut_at + (int)in.peek(bits));
  if (e) { in.skip((int)(e & 15u)); return (int)(e >> 4); }
  int code = 0, first = 0, index = 0;
  unsigned long long b = in.buf;
  for (int l = 1; l < 16; ++l) {
    code |= (int)(b & 1ull);
    b >>= 1;
    const int count = (int)L.h(cnt_at + l);
    if (code - count < first) { in.skip(l); return (int)L.h(sym_at + index + (code - first)); }
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

__constant__ uint16_t c_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// The literal / length and distance tables of a dynamic block (RFC 1951 3.2.7) into LDS.
__device__ uint32_t read_dynamic_header(const Lds& L, BitIn& in) {
  in.refill();
  const int hlit = (int)in.take(5) + 257, hdist = (int)in.take(5) + 1, hclen = (int)in.take(4) + 4;
  if (hlit > 286 || hdist > 30) return kBadCodeLengths;
  for (int i = 0; i < 19; ++i) L.b(i) = 0;
  for (int i = 0; i < hclen; ++i) {
    in.refill();
    L.b(c_cl_order[i]) = (uint8_t)in.take(3);
  }
  // the code-length code borrows the distance alphabet's arrays (they are built last)
  if (!build_tables(L, 19, kCntD, kSymD, kLutD, kDBits)) return kBadCodeLengths;
  // (the lengths are decoded into the top of the byte array first: the code-length code's own lengths sit at 0..18 until here)
  int i = 0, prev = 0;
  const int total = hlit + hdist;
  while (i < total) {
    in.refill();
    const int s = decode(L, in, kCntD, kSymD, kLutD, kDBits);
    if (s < 0) return kBadCodeLengths;
    int rep = 1, val = s;
    if (s == 16) { if (i == 0) return kBadCodeLengths; val = prev; rep = 3 + (int)in.take(2); }
    else if (s == 17) { val = 0; rep = 3 + (int)in.take(3); }
    else if (s == 18) { val = 0; rep = 11 + (int)in.take(7); }
    if (i + rep > total) return kBadCodeLengths;
    // both alphabets' lengths land in one run; they are moved apart below.  The array is read by build_tables from index 0,
    // and the code-length code's own lengths (0..18) are dead by now -- decode() reads counts and symbols, not lengths.
    for (int k = 0; k < rep; ++k) L.b(i + k) = (uint8_t)val;
    i += rep;
    prev = val;
    if (in.overrun()) return kInputOverrun;
  }
  if (L.b(256) == 0) return kBadCodeLengths;                  // no end-of-block code
  // distance lengths first (they sit behind the literal / length ones and are copied down to a scratch stretch of the
  // symbol array while the literal / length tables are built from 0..hlit)
  for (int k = 0; k < hdist; ++k) L.h(kSymD + k) = L.b(hlit + k);
  if (!build_tables(L, hlit, kCntLl, kSymLl, kLutLl, kLlBits)) return kBadCodeLengths;
  for (int k = 0; k < hdist; ++k) L.b(k) = (uint8_t)L.h(kSymD + k);
  if (!build_tables(L, hdist, kCntD, kSymD, kLutD, kDBits)) return kBadCodeLengths;
  return kOk;
}

__device__ uint32_t fixed_tables(const Lds& L) {
  for (int s = 0; s < 288; ++s) L.b(s) = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
  if (!build_tables(L, 288, kCntLl, kSymLl, kLutLl, kLlBits)) return kBadCodeLengths;
  for (int s = 0; s < 30; ++s) L.b(s) = 5;
  // (30 codes of 5 bits leave the code incomplete, as the format defines it: build by hand what build_tables would refuse)
  for (int l = 0; l < 16; ++l) L.h(kCntD + l) = 0;
  L.h(kCntD + 5) = 30;
  for (int i = 0; i < (1 << kDBits); ++i) L.h(kLutD + i) = 0;
  for (int s = 0; s < 30; ++s) {
    L.h(kSymD + s) = (uint16_t)s;
    const uint32_t r = __brev((uint32_t)s) >> 27;
    for (uint32_t k = r; k < (1u << kDBits); k += 32u) L.h(kLutD + (int)k) = (uint16_t)((s << 4) | 5);
  }
  return kOk;
}

// The inflated bytes: literals are gathered four at a time into aligned words (a byte store per literal is four times the
// stores and four times the acknowledgements the next load of input waits behind); a match first puts the gathered bytes out.
struct ByteOut {
  uint8_t* dst;
  uint32_t o;          // bytes produced, the gathered ones included
  uint32_t acc;
  int na;              // gathered bytes (they belong at o - na ..)
  uint32_t head;       // bytes in front of the first aligned word
  __device__ __forceinline__ void open(uint8_t* d) { dst = d; o = 0; acc = 0; na = 0; head = (uint32_t)((0u - reinterpret_cast<uintptr_t>(d)) & 3u); }
  __device__ __forceinline__ void literal(uint32_t b) {
    if (o < head) { dst[o++] = (uint8_t)b; return; }
    acc |= b << (8 * na);
    ++o;
    if (++na == 4) { *reinterpret_cast<uint32_t*>(dst + o - 4) = acc; acc = 0; na = 0; }
  }
  __device__ __forceinline__ void flush() {
    for (int k = 0; k < na; ++k) dst[o - na + k] = (uint8_t)(acc >> (8 * k));
    acc = 0; na = 0;
  }
};

// a noted match: position in the block's output | length << 32 | distance << 41
__device__ __forceinline__ unsigned long long match_pack(uint32_t o, uint32_t len, uint32_t dist) {
  return (unsigned long long)o | ((unsigned long long)len << 32) | ((unsigned long long)dist << 41);
}

__device__ uint32_t inflate_one(const Lds& L, const uint8_t* src, size_t clen, uint8_t* dst_, uint32_t ulen,
                                unsigned long long* mlist, uint32_t* n_matches) {
  uint32_t m = 0;
  BitIn in;
  in.open(src, clen);
  ByteOut out;
  out.open(dst_);
  uint8_t* const dst = dst_;
  for (;;) {
    in.refill();
    const uint32_t last = in.take(1), type = in.take(2);
    if (type == 0u) {                                        // stored: to the next byte, LEN, ~LEN, the bytes
      in.skip(in.n & 7);
      in.refill();
      const uint32_t len = in.take(16);
      in.refill();
      const uint32_t nlen = in.take(16);
      if ((len ^ 0xFFFFu) != nlen) return kBadStored;
      if (out.o + len > ulen) return kOutputOverrun;
      for (uint32_t k = 0; k < len; ++k) {
        in.refill();
        out.literal(in.take(8));
        if (in.overrun()) return kInputOverrun;
      }
    } else if (type == 3u) {
      return kBadBlockType;
    } else {
      const uint32_t st = type == 1u ? fixed_tables(L) : read_dynamic_header(L, in);
      if (st != kOk) return st;
      for (;;) {
        in.refill();
        int s = decode(L, in, kCntLl, kSymLl, kLutLl, kLlBits);
        if (s < 0) return kBadSymbol;
        if (s < 256) {
          if (out.o >= ulen) return kOutputOverrun;
          out.literal((uint32_t)s);
          if (in.overrun()) return kInputOverrun;
          continue;
        }
        if (s == 256) break;
        s -= 257;
        if (s >= 29) return kBadSymbol;
        const uint32_t len = c_len_base[s] + in.take(c_len_extra[s]);      // (<= 5 extra bits: still >= 12 valid bits left)
        in.refill();
        const int d = decode(L, in, kCntD, kSymD, kLutD, kDBits);
        if (d < 0 || d >= 30) return kBadDistance;
        in.refill();
        const uint32_t dist = c_dist_base[d] + in.take(c_dist_extra[d]);
        if (dist > out.o) return kBadDistance;
        if (out.o + len > ulen) return kOutputOverrun;
        // noted, not copied (at most ulen / 3 of them: the list's room)
        out.flush();
        mlist[m++] = match_pack(out.o, len, dist);
        out.o += len;
        if (in.overrun()) return kInputOverrun;
      }
      if (in.overrun()) return kInputOverrun;
    }
    if (last) break;
  }
  out.flush();
  *n_matches = m;
  return out.o == ulen ? kOk : kShortOutput;
}

__global__ __launch_bounds__(kLanes) void bgzf_inflate_kernel(InflateParams p) {
  __shared__ uint16_t s16[kU16 * kLanes];
  __shared__ uint8_t s8[kLens * kLanes];
  const long long k = (long long)blockIdx.x * kLanes + threadIdx.x;
  if (k >= p.n_blocks) return;
  Lds L{(lds_u16*)s16, (lds_u8*)s8, (int)threadIdx.x};
  const InflateBlock b = p.blocks[k];
  uint32_t st = kOk, nm = 0;
  if (b.ulen) st = inflate_one(L, p.comp + b.cpos, (size_t)b.clen, p.out + b.upos, b.ulen, p.matches + b.mbase, &nm);
  p.status[k] = st;
  p.n_matches[k] = st == kOk ? nm : 0u;
}

// The matches of one block, in order, by one wavefront: byte i of a match is the byte `distance` in front of it -- for a match
// longer than its distance the first `distance` bytes repeat, so every byte's source lies in front of the match and the 64
// lanes copy without looking at each other.  A match whose source reaches into what this wavefront has written since its last
// fence waits for those stores (and drops its cache lines) first.
constexpr int kResolveWaves = 4;
__global__ __launch_bounds__(kLanes * kResolveWaves) void bgzf_resolve_kernel(InflateParams p) {
  const long long k = (long long)blockIdx.x * kResolveWaves + (threadIdx.x >> 6);
  if (k >= p.n_blocks) return;
  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t n = p.n_matches[k];
  if (n == 0u) return;
  const InflateBlock b = p.blocks[k];
  uint8_t* out = p.out + b.upos;
  const unsigned long long* list = p.matches + b.mbase;
  uint32_t dirty = 0xFFFFFFFFu;           // the lowest position written since the last fence
  unsigned long long rec = list[0];
  for (uint32_t m = 0; m < n; ++m) {
    const unsigned long long next = m + 1 < n ? list[m + 1] : 0ull;
    const uint32_t o = (uint32_t)rec, len = (uint32_t)(rec >> 32) & 511u, dist = (uint32_t)(rec >> 41);
    const uint32_t span = len < dist ? len : dist;
    if (o - dist + span > dirty) {          // (wave-uniform)
      __threadfence();
      dirty = 0xFFFFFFFFu;
    }
    for (uint32_t i = (uint32_t)lane; i < len; i += 64u) {
      const uint32_t from = o - dist + (i < dist ? i : i % dist);
      out[o + i] = out[from];
    }
    dirty = dirty < o ? dirty : o;
    rec = next;
  }
}

}  // namespace

hipError_t launch_bgzf_inflate(const InflateParams& p, hipStream_t s) {
  if (p.n_blocks <= 0) return hipSuccess;
  const long long g = (p.n_blocks + kLanes - 1) / kLanes;
  hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((unsigned)g), dim3(kLanes), 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const long long g2 = (p.n_blocks + kResolveWaves - 1) / kResolveWaves;
  hipLaunchKernelGGL(bgzf_resolve_kernel, dim3((unsigned)g2), dim3(kLanes * kResolveWaves), 0, s, p);
  return hipGetLastError();
}

}  // namespace midas
