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
#include "crc32.h"

namespace midas {
namespace {

constexpr int kLanes = 64;
// Streams per workgroup of the decoder.  A stream's tables take 2.5 KiB of LDS, so a CU holds 64 streams whatever the shape;
// what the shape decides is how many WAVEFRONTS those 64 streams are.  The decoder is a chain of dependent LDS reads (table
// look-up, input ring) and the lanes of a wavefront diverge at every step (literal / match / long code / refill: a step costs
// the sum of the paths any lane takes).  Measured on the 46 500 blocks of configs[2]'s BAM (profiles/r05_inflate_shapes.txt):
// 64 streams per wavefront 78.7 ms, 16: 76.3, 8: 68.3, 4: 83.4 (the CU no longer holds 64 streams), 2: 122.  With the literal
// stores compiled out: 73 of 76 ms -- the stores are not what it waits for.
#ifndef MIDAS_INFLATE_LANES
#define MIDAS_INFLATE_LANES 8
#endif
constexpr int kDecLanes = MIDAS_INFLATE_LANES;
static_assert(kDecLanes >= 2 && kDecLanes <= 64 && (kDecLanes & (kDecLanes - 1)) == 0, "a power of two up to a wavefront");
constexpr int kLlBits = 9, kDBits = 7;
// u16 arrays, per lane, interleaved [index][lane]
constexpr int kLutLl = 0;                       // 512: (symbol << 4) | length, 0 = not in the table
constexpr int kLutD = kLutLl + (1 << kLlBits);  // 128 (also the code-length code's table while a dynamic header is read)
constexpr int kSymLl = kLutD + (1 << kDBits);   // 288: symbols sorted by (length, symbol)
constexpr int kSymD = kSymLl + 288;             // 32
constexpr int kCntLl = kSymD + 32;              // 16: codes per length
constexpr int kCntD = kCntLl + 16;              // 16
// per length, while a table is built: where the next symbol of that length goes in the sorted list, and the next code of that
// length; afterwards, therefore: the END of the length's symbols and the END of its codes -- what the long-code path needs
constexpr int kOffsLl = kCntD + 16;             // 16
constexpr int kNextLl = kOffsLl + 16;           // 16
constexpr int kOffsD = kNextLl + 16;            // 16
constexpr int kNextD = kOffsD + 16;             // 16
constexpr int kU16 = kNextD + 16;               // 1056
constexpr int kRing = 16;                       // u32, per lane: the next 64 bytes of the lane's stream
constexpr int kLens = 320;                      // u8: code lengths while a table is built

enum : uint32_t { kOk = 0, kBadBlockType = 1, kBadStored = 2, kBadCodeLengths = 3, kBadSymbol = 4, kBadDistance = 5,
                  kOutputOverrun = 6, kInputOverrun = 7, kShortOutput = 8, kMatchRoom = kInflateMatchRoom };

// (pointers that SAY they point into LDS: through a generic pointer every table look-up would be a FLAT access, which waits
// for the thread's outstanding global stores -- one store acknowledgement per symbol)
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
struct Lds {
  lds_u16* s16;
  lds_u8* s8;
  lds_u32* s32;
  int lane;
  __device__ __forceinline__ lds_u32& r(uint32_t i) const { return s32[(i & (kRing - 1)) * kDecLanes + lane]; }
  __device__ __forceinline__ lds_u16& h(int i) const { return s16[i * kDecLanes + lane]; }
  __device__ __forceinline__ lds_u8& b(int i) const { return s8[i * kDecLanes + lane]; }
};

// The compressed stream, least significant bit first (RFC 1951 3.1.1).  Between memory and the bit buffer sits a ring of 32
// words per lane in LDS, topped up 32 bytes at a time by loads that are issued EIGHT steps of the symbol loop before their
// words are needed.  (A lane that loaded its next word only when it ran dry would not stall just itself: a wavefront waits
// for its outstanding loads as one, some lane of 64 runs dry on nearly every step, and every step would cost a trip to
// memory -- 2.3 us per symbol, measured.)
struct BitIn {
  const uint32_t* w;       // next aligned words to fetch
  unsigned long long buf;
  int n;                   // valid bits in buf
  long long budget;        // bits of the stream not yet moved into buf (negative: the stream has been overrun)
  uint32_t rd, wr;         // ring positions (words, counted up for ever)
  uint32_t pre[8];         // the words on their way
  bool on_the_way;
  __device__ __forceinline__ void fetch() {
#pragma unroll
    for (int k = 0; k < 8; ++k) pre[k] = w[k];
    w += 8;
    on_the_way = true;
  }
  __device__ __forceinline__ void land(const Lds& L) {
#pragma unroll
    for (int k = 0; k < 8; ++k) L.r(wr + (uint32_t)k) = pre[k];
    wr += 8u;
    on_the_way = false;
  }
  __device__ __forceinline__ void open(const Lds& L, const uint8_t* p, size_t len) {
    buf = 0; n = 0;
    budget = (long long)len * 8;
    while ((reinterpret_cast<uintptr_t>(p) & 3u) && n < 32) {      // bytes up to the first aligned word
      buf |= (unsigned long long)(*p++) << n;
      n += 8;
    }
    budget -= n;
    w = reinterpret_cast<const uint32_t*>(p);
    rd = wr = 0u;
    fetch();
  }
  // every eighth step of the symbol loop, all lanes of the step together: what was fetched last time goes into the ring,
  // the next 32 bytes are asked for (if the ring has room for them behind those)
  __device__ __forceinline__ void top_up(const Lds& L) {
    if (on_the_way && wr - rd <= (uint32_t)kRing - 8u) land(L);
    if (!on_the_way && wr - rd <= (uint32_t)kRing - 8u && budget - 32ll * (long long)(wr - rd) > -1024) fetch();
  }
  // at least 32 valid bits behind this (the stream buffer has 256 bytes of slack behind the last stream)
  __device__ __forceinline__ void refill(const Lds& L) {
    if (n <= 32) {
      if (rd == wr) {                 // the ring ran dry (a table header, a stretch of long matches): wait for the words
        if (!on_the_way) fetch();
        land(L);
      }
      buf |= (unsigned long long)L.r(rd) << n;
      ++rd;
      n += 32;
      budget -= 32;
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
__device__ bool build_tables(const Lds& L, int n, int cnt_at, int sym_at, int lut_at, int bits, int kOffs, int kNext) {
  for (int l = 0; l < 16; ++l) { L.h(cnt_at + l) = 0; L.h(kOffs + l) = 0; L.h(kNext + l) = 0; }
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
  // per length: where its next symbol goes in the sorted list, and its next code (RFC 1951 3.2.2) -- in LDS, indexed by the
  // symbol's length (a lane's own index: registers cannot be indexed per lane)
  {
    uint32_t off = 0, code = 0;
    L.h(kOffs) = 0; L.h(kNext) = 0;
    for (int l = 1; l < 16; ++l) {
      const uint32_t c = L.h(cnt_at + l);
      code = (code + (l > 1 ? (uint32_t)L.h(cnt_at + l - 1) : 0u)) << 1;
      L.h(kOffs + l) = (uint16_t)off;
      L.h(kNext + l) = (uint16_t)code;
      off += c;
    }
  }
  for (int s = 0; s < n; ++s) {
    const int l = L.b(s);
    if (!l) continue;
    const uint32_t o = L.h(kOffs + l), c = L.h(kNext + l);
    L.h(kOffs + l) = (uint16_t)(o + 1u);
    L.h(kNext + l) = (uint16_t)(c + 1u);
    L.h(sym_at + (int)o) = (uint16_t)s;
    if (l <= bits) {
      const uint32_t r = __brev(c) >> (32 - l);
      for (uint32_t k = r; k < (1u << bits); k += 1u << l) L.h(lut_at + (int)k) = (uint16_t)((s << 4) | l);
    }
  }
  return true;
}

// One symbol: the look-up table; a code longer than the table's width (a rare symbol -- but with 64 streams in a wavefront
// some lane meets one on most steps, so this path has to be short too) is found by its length: canonical codes of length l
// are the numbers below end[l] that no shorter code is a prefix of, and their symbols end at sorted position offs_end[l].
// Returns -1 on a code no symbol has.
__device__ __forceinline__ int decode(const Lds& L, BitIn& in, int cnt_at, int sym_at, int lut_at, int bits, int offs_at, int next_at) {
  const uint32_t e = L.h(lut_at + (int)in.peek(bits));
  if (e) { in.skip((int)(e & 15u)); return (int)(e >> 4); }
  const uint32_t v15 = __brev((uint32_t)in.buf) >> 17;          // the next 15 bits as a code reads them, first bit on top
  for (int l = bits + 1; l < 16; ++l) {
    const uint32_t code = v15 >> (15 - l);
    const uint32_t end = L.h(next_at + l);
    if (code < end) {
      const uint32_t count = L.h(cnt_at + l);
      if (code + count < end) return -1;                         // (below the length's first code: an incomplete code's gap)
      in.skip(l);
      return (int)L.h(sym_at + (int)(L.h(offs_at + l) - (end - code)));
    }
  }
  return -1;
}

// RFC 1951 3.2.5 in closed form (a table in constant memory would be a trip to memory, waited for, on every match -- and
// with 64 streams in a wavefront some lane has a match on almost every step):
//   length code 257 + s -> (first length, extra bits);  distance code d -> (first distance, extra bits)
__device__ __forceinline__ void length_of(int s, uint32_t* base, int* extra) {
  if (s < 8) { *base = 3u + (uint32_t)s; *extra = 0; return; }
  if (s == 28) { *base = 258u; *extra = 0; return; }
  *extra = (s >> 2) - 1;
  *base = 3u + ((4u + (uint32_t)(s & 3)) << *extra);
}
__device__ __forceinline__ void distance_of(int d, uint32_t* base, int* extra) {
  if (d < 4) { *base = 1u + (uint32_t)d; *extra = 0; return; }
  *extra = (d >> 1) - 1;
  *base = 1u + ((2u + (uint32_t)(d & 1)) << *extra);
}
__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// The literal / length and distance tables of a dynamic block (RFC 1951 3.2.7) into LDS.
__device__ uint32_t read_dynamic_header(const Lds& L, BitIn& in) {
  in.refill(L);
  const int hlit = (int)in.take(5) + 257, hdist = (int)in.take(5) + 1, hclen = (int)in.take(4) + 4;
  if (hlit > 286 || hdist > 30) return kBadCodeLengths;
  for (int i = 0; i < 19; ++i) L.b(i) = 0;
  for (int i = 0; i < hclen; ++i) {
    in.refill(L);
    L.b(c_cl_order[i]) = (uint8_t)in.take(3);
  }
  // the code-length code borrows the distance alphabet's arrays (they are built last)
  if (!build_tables(L, 19, kCntD, kSymD, kLutD, kDBits, kOffsD, kNextD)) return kBadCodeLengths;
  // (the lengths are decoded into the top of the byte array first: the code-length code's own lengths sit at 0..18 until here)
  int i = 0, prev = 0;
  const int total = hlit + hdist;
  while (i < total) {
    in.refill(L);
    const int s = decode(L, in, kCntD, kSymD, kLutD, kDBits, kOffsD, kNextD);
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
  if (!build_tables(L, hlit, kCntLl, kSymLl, kLutLl, kLlBits, kOffsLl, kNextLl)) return kBadCodeLengths;
  for (int k = 0; k < hdist; ++k) L.b(k) = (uint8_t)L.h(kSymD + k);
  if (!build_tables(L, hdist, kCntD, kSymD, kLutD, kDBits, kOffsD, kNextD)) return kBadCodeLengths;
  return kOk;
}

__device__ uint32_t fixed_tables(const Lds& L) {
  for (int s = 0; s < 288; ++s) L.b(s) = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
  if (!build_tables(L, 288, kCntLl, kSymLl, kLutLl, kLlBits, kOffsLl, kNextLl)) return kBadCodeLengths;
  for (int s = 0; s < 30; ++s) L.b(s) = 5;
  // (30 codes of 5 bits leave the code incomplete, as the format defines it: build by hand what build_tables would refuse)
  for (int l = 0; l < 16; ++l) L.h(kCntD + l) = 0;
  L.h(kCntD + 5) = 30;
  for (int l = 0; l < 16; ++l) { L.h(kOffsD + l) = (uint16_t)(l >= 5 ? 30 : 0); L.h(kNextD + l) = 0; }     // (every code is in the table)
  for (int i = 0; i < (1 << kDBits); ++i) L.h(kLutD + i) = 0;
  for (int s = 0; s < 30; ++s) {
    L.h(kSymD + s) = (uint16_t)s;
    const uint32_t r = __brev((uint32_t)s) >> 27;
    for (uint32_t k = r; k < (1u << kDBits); k += 32u) L.h(kLutD + (int)k) = (uint16_t)((s << 4) | 5);
  }
  return kOk;
}

// The inflated bytes: literals are gathered four at a time into ALIGNED words (a byte store per literal is four times the
// stores and four times the acknowledgements the next load of input waits behind). A match first puts the gathered bytes
// out; the literals behind it go out byte by byte until the output position is a multiple of four again (`lead`), so that
// every word store is aligned wherever the matches fall.
struct ByteOut {
  uint8_t* dst;
  uint32_t o;          // bytes produced, the gathered ones included
  uint32_t acc;
  int na;              // gathered bytes (they belong at o - na ..)
  uint32_t lead;       // literals to store singly before the next aligned word starts
  uint32_t skew;       // (0 - dst) & 3: dst + o is a multiple of four exactly where o - skew is
  __device__ __forceinline__ void open(uint8_t* d) {
    dst = d; o = 0; acc = 0; na = 0;
    skew = (uint32_t)((0u - reinterpret_cast<uintptr_t>(d)) & 3u);
    lead = skew;
  }
  __device__ __forceinline__ void literal(uint32_t b) {
    if (lead) { dst[o++] = (uint8_t)b; --lead; return; }
    acc |= b << (8 * na);
    ++o;
#ifdef MIDAS_INFLATE_NO_STORE      // (developer timing variant: the literals are not stored -- WRONG output)
    if (++na == 4) { acc = 0; na = 0; }
#else
    if (++na == 4) { *reinterpret_cast<uint32_t*>(dst + o - 4) = acc; acc = 0; na = 0; }
#endif
  }
  __device__ __forceinline__ void flush() {
    for (int k = 0; k < na; ++k) dst[o - na + k] = (uint8_t)(acc >> (8 * k));
    acc = 0; na = 0;
  }
  // `len` bytes are left for someone else to fill (a noted match); the gathered literals must be out already
  __device__ __forceinline__ void leave(uint32_t len) { o += len; lead = (skew - o) & 3u; }
};

// a noted match: position in the block's output | length << 32 | distance << 41
__device__ __forceinline__ unsigned long long match_pack(uint32_t o, uint32_t len, uint32_t dist) {
  return (unsigned long long)o | ((unsigned long long)len << 32) | ((unsigned long long)dist << 41);
}

__device__ uint32_t inflate_one(const Lds& L, const uint8_t* src, size_t clen, uint8_t* dst_, uint32_t ulen,
                                unsigned long long* mlist, uint32_t mcap, uint32_t* n_matches) {
  uint32_t m = 0, tick = 0;
  BitIn in;
  in.open(L, src, clen);
  ByteOut out;
  out.open(dst_);
  uint8_t* const dst = dst_;
  for (;;) {
    in.refill(L);
    const uint32_t last = in.take(1), type = in.take(2);
    if (type == 0u) {                                        // stored: to the next byte, LEN, ~LEN, the bytes
      in.skip(in.n & 7);
      in.refill(L);
      const uint32_t len = in.take(16);
      in.refill(L);
      const uint32_t nlen = in.take(16);
      if ((len ^ 0xFFFFu) != nlen) return kBadStored;
      if (out.o + len > ulen) return kOutputOverrun;
      for (uint32_t k = 0; k < len; ++k) {
        in.refill(L);
        out.literal(in.take(8));
        if (in.overrun()) return kInputOverrun;
      }
    } else if (type == 3u) {
      return kBadBlockType;
    } else {
      const uint32_t st = type == 1u ? fixed_tables(L) : read_dynamic_header(L, in);
      if (st != kOk) return st;
      in.refill(L);
      for (;;) {
        // (one count for all the lanes that take this step together: they top their rings up at the same steps)
        tick = (uint32_t)__builtin_amdgcn_readfirstlane((int)tick) + 1u;
        if ((tick & 7u) == 0u) in.top_up(L);
        // at least 15 valid bits here (32 after a literal, 19 after a match): the symbol is decoded first and the buffer filled
        // up behind it -- one round trip to the ring fewer on the way to the next table look-up
        int s = decode(L, in, kCntLl, kSymLl, kLutLl, kLlBits, kOffsLl, kNextLl);
        in.refill(L);
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
        uint32_t len;
        int extra;
        length_of(s, &len, &extra);
        len += in.take(extra);                                               // (<= 5 extra bits of >= 32: 27 left for the distance code)
        const int d = decode(L, in, kCntD, kSymD, kLutD, kDBits, kOffsD, kNextD);
        if (d < 0 || d >= 30) return kBadDistance;
        in.refill(L);
        uint32_t dist;
        distance_of(d, &dist, &extra);
        dist += in.take(extra);
        if (dist > out.o) return kBadDistance;
        if (out.o + len > ulen) return kOutputOverrun;
        // noted, not copied
        out.flush();
        if (m == mcap) return kMatchRoom;
        mlist[m++] = match_pack(out.o, len, dist);
        out.leave(len);
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

__global__ __launch_bounds__(kDecLanes) void bgzf_inflate_kernel(InflateParams p) {
  __shared__ uint16_t s16[kU16 * kDecLanes];
  __shared__ uint8_t s8[kLens * kDecLanes];
  __shared__ uint32_t s32[kRing * kDecLanes];
  const long long k = (long long)blockIdx.x * kDecLanes + threadIdx.x;
  if (k >= p.n_blocks) return;
  Lds L{(lds_u16*)s16, (lds_u8*)s8, (lds_u32*)s32, (int)threadIdx.x};
  const InflateBlock b = p.blocks[k];
  uint32_t st = kOk, nm = 0;
  if (b.ulen) st = inflate_one(L, p.comp + b.cpos, (size_t)b.clen, p.out + b.upos, b.ulen, p.matches + b.mbase, b.mcap, &nm);
  p.status[k] = st;
  p.n_matches[k] = st == kOk ? nm : 0u;
}

// The matches of one block, in order, by one wavefront -- 64 at a time where they allow it.  A match only reads what lies in
// front of it, so a run of consecutive matches whose sources all lie in front of the run's FIRST destination can be copied at
// the same moment, one lane per match (most matches of a BAM are a few bytes long); the run ends in front of the first match
// that reads what the run writes.  One fence per run makes the run's bytes visible to the next one's loads.  Inside a lane the
// copy goes in pieces no longer than the distance and no longer than 16 bytes, loads first, then stores (a match longer than
// its distance repeats its first `distance` bytes: the pieces read what the lane itself stored a step earlier).
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
  uint32_t m = 0;
  while (m < n) {
    const bool have = m + (uint32_t)lane < n;
    const unsigned long long rec = have ? list[m + (uint32_t)lane] : 0ull;
    const uint32_t o = (uint32_t)rec, len = (uint32_t)(rec >> 32) & 511u, dist = (uint32_t)(rec >> 41);
    const uint32_t first = (uint32_t)__shfl((int)o, 0);                  // the run's first destination
    const uint32_t span = len < dist ? len : dist;
    const bool free_of_run = have && (lane == 0 || o - dist + span <= first);
    const unsigned long long ok = __ballot(free_of_run);
    const int run = ok == ~0ull ? 64 : (int)__ffsll((long long)~ok) - 1;  // leading lanes that are free
    if (lane < run) {
      uint32_t at = o, left = len;
      while (left) {
        uint32_t piece = left < dist ? left : dist;
        piece = piece < 16u ? piece : 16u;
        uint8_t t[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = (uint32_t)q < piece ? out[at - dist + q] : (uint8_t)0;
#pragma unroll
        for (int q = 0; q < 16; ++q) if ((uint32_t)q < piece) out[at + q] = t[q];
        at += piece;
        left -= piece;
      }
    }
    // (the run's stores in front of the next run's loads.  The same wavefront, the same CU's cache: a workgroup-scope fence.
    // An agent-scope __threadfence() writes the XCD's whole L2 back -- 150 us a time here.)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    m += (uint32_t)run;
  }
}

// The payload columns cut out of the inflated stream where it lies, in HBM: one wavefront per record copies the record's
// CIGAR ops, 4-bit SEQ and QUAL (BAM record layout: SAM spec 4.2) to the offsets the host's size pass computed.  What the host
// decoder does with three memcpy per record -- and then sends up the link again.
__global__ __launch_bounds__(256) void bam_payload_kernel(PayloadParams p) {
  const int lane = (int)(threadIdx.x & 63u);
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * 4;
  for (long long i = wave; i < p.n_records; i += n_waves) {
    const uint8_t* r = p.stream + p.rec_off[i];
    const uint32_t l_name = r[12];
    const uint32_t n_cig = (uint32_t)r[16] | ((uint32_t)r[17] << 8);
    const uint32_t l = (uint32_t)r[20] | ((uint32_t)r[21] << 8) | ((uint32_t)r[22] << 16) | ((uint32_t)r[23] << 24);
    const uint8_t* q = r + 36 + l_name;
    uint32_t* cg = p.cigar + p.cigar_off[i];
    for (uint32_t k = (uint32_t)lane; k < n_cig; k += 64u) {
      const uint8_t* s = q + 4u * k;
      cg[k] = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
    }
    q += 4ull * n_cig;
    uint8_t* sq = p.seq4 + p.seq_off[i];
    const uint32_t ns = (l + 1u) / 2u;
    for (uint32_t k = (uint32_t)lane; k < ns; k += 64u) sq[k] = q[k];
    q += ns;
    uint8_t* ql = p.qual + p.qual_off[i];
    for (uint32_t k = (uint32_t)lane; k < l; k += 64u) ql[k] = q[k];
  }
}

}  // namespace

// The CRC-32 of every inflated block against the one its BGZF footer holds (htslib checks it behind pysam.AlignmentFile,
// midas/run/snps.py:186; a flipped literal bit inflates to the right SIZE).  One wavefront per block, behind the resolver: a
// lane runs the table-driven CRC over its 1/64 of the block from a zero register, the registers are moved to the end of the
// block (crc32.h) and folded.  A block that inflated cleanly but sums wrongly gets the status kInflateCrc.
constexpr int kCrcWaves = 4;
__global__ __launch_bounds__(kLanes * kCrcWaves) void bgzf_crc_kernel(InflateParams p) {
  __shared__ crc::Tables T;
  crc::build_tables(T);
  const int lane = (int)(threadIdx.x & 63u);
  const long long n_waves = (long long)gridDim.x * kCrcWaves;
  for (long long k = (long long)blockIdx.x * kCrcWaves + (threadIdx.x >> 6); k < p.n_blocks; k += n_waves) {
    if (p.status[k] != 0u) continue;
    const InflateBlock b = p.blocks[k];
    const uint32_t chunk = ((b.ulen + 63u) / 64u + 3u) & ~3u;
    const uint32_t lo = (uint32_t)lane * chunk;
    uint32_t part = 0u;
    if (lo < b.ulen) {
      const uint32_t n = b.ulen - lo < chunk ? b.ulen - lo : chunk;
      const uint32_t c = crc::update(T, 0u, p.out + b.upos + lo, n);
      part = crc::gf_mul(c, crc::gf_xpow8((unsigned long long)(b.ulen - lo - n), T.x2n));
    }
    if (lane == 0) part ^= crc::gf_mul(0xFFFFFFFFu, crc::gf_xpow8((unsigned long long)b.ulen, T.x2n));      // the initial register
    for (int d = 32; d >= 1; d >>= 1) part ^= (uint32_t)__shfl_xor((int)part, d);
    if (lane == 0 && (part ^ 0xFFFFFFFFu) != p.want_crc[k]) p.status[k] = kInflateCrc;
  }
}

hipError_t launch_bam_payload(const PayloadParams& p, int grid_blocks, hipStream_t s) {
  if (p.n_records <= 0) return hipSuccess;
  long long g = (p.n_records + 3) / 4;
  const long long cap = (long long)grid_blocks * 16;
  if (g > cap) g = cap;
  hipLaunchKernelGGL(bam_payload_kernel, dim3((unsigned)g), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_bgzf_inflate(const InflateParams& p, hipStream_t s, int phases) {
  if (p.n_blocks <= 0) return hipSuccess;
  if (phases & 1) {
    // the block inflater keeps its streams' tables and input rings in LDS (2.5 KiB a stream). A device that cannot give a
    // workgroup that much cannot run it; say so instead of leaving it to the launch (the callers fall back to the host's inflater).
    static const bool fits = [] {
      hipFuncAttributes fa{};
      int dev = 0, lds = 0;
      if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(bgzf_inflate_kernel)) != hipSuccess) return true;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return true;
      return fa.sharedSizeBytes <= (size_t)lds;
    }();
    if (!fits) return hipErrorLaunchOutOfResources;
    const long long g = (p.n_blocks + kDecLanes - 1) / kDecLanes;
    hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((unsigned)g), dim3(kDecLanes), 0, s, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (phases & 4) {                        // the CRC check alone
    if (!p.want_crc) return hipSuccess;
    long long g3 = (p.n_blocks + kCrcWaves - 1) / kCrcWaves;
    g3 = g3 > 2048 ? 2048 : g3;
    hipLaunchKernelGGL(bgzf_crc_kernel, dim3((unsigned)g3), dim3(kLanes * kCrcWaves), 0, s, p);
    return hipGetLastError();
  }
  if (!(phases & 2)) return hipSuccess;
  const long long g2 = (p.n_blocks + kResolveWaves - 1) / kResolveWaves;
  hipLaunchKernelGGL(bgzf_resolve_kernel, dim3((unsigned)g2), dim3(kLanes * kResolveWaves), 0, s, p);
  if (p.want_crc && !(phases & 8)) {       // (8: a caller that times the phases launches the check by itself, phases = 4)
    long long g3 = (p.n_blocks + kCrcWaves - 1) / kCrcWaves;
    g3 = g3 > 2048 ? 2048 : g3;
    hipLaunchKernelGGL(bgzf_crc_kernel, dim3((unsigned)g3), dim3(kLanes * kCrcWaves), 0, s, p);
  }
  return hipGetLastError();
}

}  // namespace midas
