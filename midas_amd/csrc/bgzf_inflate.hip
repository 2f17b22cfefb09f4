// BGZF blocks inflated on the device: one LANE per block decodes its Huffman codes into a token stream, one WAVEFRONT per
// block then lays the block's bytes out from the tokens.
//
// A BAM is a chain of independent <= 64 KiB DEFLATE streams (BGZF, SAM spec 4.1); htslib inflates them one after the other,
// this library's host decoder on all cores -- and with the pileup at a millisecond and the rows coded on the device, that
// host inflate is two thirds of the stage.  DEFLATE decoding is a serial dependency chain inside a stream, but
// a 1.3 GB BAM holds 45 000 streams: each lane of a wavefront takes one (RFC 1951: stored, fixed and dynamic blocks).
//
// Two kernels, and what passes between them (a block of a BAM is ~13 000 literals and ~8 000 matches of 6.5 bytes):
//   decode   (namespace w64 below) Huffman decoding never looks at the output, so the decoder writes NO output byte.  A lane
//            turns its stream into (a) TOKENS, one dword per match: literals in front of it (9 bits) | length - 3 (8) |
//            distance - 1 (15) -- a run of 511 literals or more, and the literals behind the last match, as an escape token
//            that only carries a count -- and (b) the stream's LITERALS, back to back.  Both are gathered in small per-lane
//            rings in LDS and leave for HBM in 16-byte stores at the one point of the symbol loop that all lanes of a
//            wavefront pass together every fourth step, where the input ring is topped up as well: no memory instruction sits
//            in a divergent branch of the loop.  The tokens grow from the front of the block's room in `matches`, the
//            literals (bytes in reverse order) from its end: one room, no second buffer.
//   place    one wavefront per block, 64 tokens at a time: two prefix sums give every token its place in the output and in
//            the literal stream; the literals of the NEXT 64 tokens are copied while this window's matches are resolved --
//            a match only reads what lies in front of it, so every unfinished match whose source ends in front of the first
//            unfinished token's bytes is copied at the same moment, one lane per match, 8 bytes per load (most matches of a
//            BAM are a few bytes long); one workgroup-scope fence per such step.  Tokens are fetched a window ahead.
// History of the decoder (git log): round 3 -- one lane per stream with 9-bit / 7-bit look-up tables in LDS
// (2.5 KiB a stream: 64 streams a CU), literals stored where they belong, matches noted for a resolver that copied them with
// byte loads and a fence per run: 68-77 ms + 20 ms on configs[2]'s 46 500 blocks.  Round 5 -- tokens instead of output bytes
// (no faster by itself: a step of the look-up decoder is three hundred dependent instructions of one wavefront, 3 000
// cycles, whatever the stores do), then the decoder below: 26 ms + 10 ms.
// Replaces the inflate inside `pysam.AlignmentFile(...)` of midas/run/snps.py:186 (htslib's bgzf.c); bounds-checked against
// both buffers at every step: corrupt input yields a status, never a fault.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include "crc32.h"

namespace midas {
namespace {

constexpr int kLanes = 64;

enum : uint32_t { kOk = 0, kBadBlockType = 1, kBadStored = 2, kBadCodeLengths = 3, kBadSymbol = 4, kBadDistance = 5,
                  kOutputOverrun = 6, kInputOverrun = 7, kShortOutput = 8, kMatchRoom = kInflateMatchRoom };

constexpr uint32_t kEscape = 0xFF800000u;       // token: 511 in the literal field, the low 23 bits a count of literals, no match

// (pointers that SAY they point into LDS: through a generic pointer every table look-up would be a FLAT access, which waits
// for the thread's outstanding global stores -- one store acknowledgement per symbol)
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32_a1 __attribute__((aligned(1)));
typedef uint16_t u16_a1 __attribute__((aligned(1)));
typedef unsigned long long u64_a1 __attribute__((aligned(1)));

// RFC 1951 3.2.5 in closed form (a table in constant memory would be a trip to memory, waited for, on every match -- and
// with several streams in a wavefront some lane has a match on almost every step):
//   length code 257 + s -> (first length, extra bits);  distance code d -> (first distance, extra bits)
__device__ __forceinline__ void length_of(int s, uint32_t* base, int* extra) {
  const int e = s < 8 ? 0 : (s >> 2) - 1;
  uint32_t b = s < 8 ? 3u + (uint32_t)s : 3u + ((4u + (uint32_t)(s & 3)) << e);
  *extra = s == 28 ? 0 : e;
  *base = s == 28 ? 258u : b;
}
__device__ __forceinline__ void distance_of(int d, uint32_t* base, int* extra) {
  const int e = d < 4 ? 0 : (d >> 1) - 1;
  *extra = e;
  *base = d < 4 ? 1u + (uint32_t)d : 1u + ((2u + (uint32_t)(d & 1)) << e);
}
__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// What a stream's decoder leaves behind: tokens and literals, in the stream's ROOM of `room` dwords (kernels.h InflateBlock:
// mbase, mcap).  Dwords 0 and 1 stay free (the placer's 8-byte loads of literals may reach that far down), token i is dword
// 2 + i, literal j the byte 4 * room - 1 - j: the literals run DOWN from the room's end, four to a dword.
template <class LV>
struct TokenOut {
  static constexpr uint32_t kLitW = LV::kLitW, kTokW = LV::kTokW;
  uint32_t* area;
  uint32_t room;
  uint32_t o;          // bytes of output accounted for
  uint32_t run;        // literals since the last token
  uint32_t acc;        // literals gathered for the next dword (first literal in the top byte)
  int na;
  uint32_t lw, lr;     // literal dwords gathered / stored
  uint32_t tw, tr;     // tokens gathered / stored
  bool full;           // the room is too small (the stream is decoded again with more)
  __device__ __forceinline__ void open(uint32_t* a, uint32_t room_dwords) {
    area = a; room = room_dwords; o = 0; run = 0; acc = 0; na = 0; lw = lr = tw = tr = 0; full = false;
  }
  // (one dword kept free for the literals' last, partly filled dword)
  __device__ __forceinline__ bool fits() const { return 2u + tw + lw + 2u <= room; }
  __device__ __forceinline__ void store_lits(const LV& L) {           // four dwords, the earliest literals at the highest address
    u32x4_a4 v;
    v.w = L.lit(lr); v.z = L.lit(lr + 1u); v.y = L.lit(lr + 2u); v.x = L.lit(lr + 3u);
    *reinterpret_cast<u32x4_a4*>(area + (room - 4u - lr)) = v;
    lr += 4u;
  }
  __device__ __forceinline__ void store_toks(const LV& L) {
    u32x4_a4 v;
    v.x = L.tok(tr); v.y = L.tok(tr + 1u); v.z = L.tok(tr + 2u); v.w = L.tok(tr + 3u);
    *reinterpret_cast<u32x4_a4*>(area + (2u + tr)) = v;
    tr += 4u;
  }
  __device__ __forceinline__ void literal(const LV& L, uint32_t b) {
    acc |= b << (24 - 8 * na);
    ++o; ++run;
    if (++na == 4) {
      if (!fits()) { full = true; acc = 0; na = 0; return; }
      if (lw - lr == kLitW) store_lits(L);               // (only when the lanes' common step was long in coming)
      L.lit(lw) = acc;
      ++lw;
      acc = 0; na = 0;
    }
  }
  __device__ __forceinline__ void push(const LV& L, uint32_t t) {
    if (!fits()) { full = true; return; }
    if (tw - tr == kTokW) store_toks(L);
    L.tok(tw) = t;
    ++tw;
  }
  // one step of the symbol loop: a literal, a match or neither (predicated: every lane runs it)
  __device__ __forceinline__ void step(const LV& L, bool lit, uint32_t b, bool mat, uint32_t len, uint32_t dist) {
    acc |= lit ? b << (24 - 8 * na) : 0u;
    na += lit ? 1 : 0;
    if (na == 4) {
      if (!fits()) { full = true; }
      else {
        if (lw - lr == kLitW) store_lits(L);
        L.lit(lw) = acc;
        ++lw;
      }
      acc = 0; na = 0;
    }
    if (mat && run >= 511u) { push(L, kEscape | run); run = 0; }
    if (mat) push(L, (run << 23) | ((len - 3u) << 15) | (dist - 1u));
    run = mat ? 0u : run + (lit ? 1u : 0u);
    o += lit ? 1u : (mat ? len : 0u);
  }
  __device__ __forceinline__ void match(const LV& L, uint32_t len, uint32_t dist) {
    if (run >= 511u) { push(L, kEscape | run); run = 0; }
    push(L, (run << 23) | ((len - 3u) << 15) | (dist - 1u));
    run = 0;
    o += len;
  }
  // the lanes' common step: whole 16-byte stores of what has gathered
  __device__ __forceinline__ void service(const LV& L) {
    if (lw - lr >= 4u) store_lits(L);
    while (tw - tr >= 4u) store_toks(L);
  }
  __device__ __forceinline__ void finish(const LV& L) {
    if (run) { push(L, kEscape | run); run = 0; }
    if (na) {
      if (!fits()) full = true;
      else { if (lw - lr == kLitW) store_lits(L); L.lit(lw) = acc; ++lw; }
      acc = 0; na = 0;
    }
    if (full) return;
    for (; tr < tw; ++tr) area[2u + tr] = L.tok(tr);
    for (; lr < lw; ++lr) area[room - 1u - lr] = L.lit(lr);
  }
};

// ---- the decoder ----------------------------------------------------------------------------------------------------------
// What a CU must hold is STREAMS: a step of the symbol loop is a few hundred instructions of one wavefront, most of them
// depending on the one before, and only other wavefronts can fill the gaps.  This decoder needs 512 bytes of LDS a stream, so
// a CU holds five full wavefronts of 64 streams (a look-up-table decoder: 2.3 KiB, 64 streams):
//   * no look-up table.  A canonical Huffman code is decoded by COMPARISON: with the next 15 bits read as a code reads them
//     (first bit on top), the codes of length l are exactly the values in [limit[l-1], limit[l]) -- limit[l] = (first code of
//     length l + their number) << (15 - l), increasing in l -- so the length is one more than the number of limits the value
//     has reached: fifteen compares against REGISTERS (limit - 1 as 16-bit halves, two to a register, one packed subtraction
//     per pair and no condition code: a lane's two alphabets are 16 VGPRs), no branch, no rare long-code path for the lanes of
//     a wavefront to diverge into.  The symbol then is sorted_symbols[(value >> (15 - l)) + base[l]]: two dependent LDS reads
//     (16 + 288 entries; the symbols' ninth bit in a bitmap).
//   * the lanes' LDS is interleaved by dword ([word][lane]): any access pattern is conflict-free.
//   * a table header is read in TWO PASSES over its bits instead of through a buffer of code lengths (320 bytes a lane that
//     LDS does not have): pass one counts the codes per length, the limits and bases follow from the counts, pass two reads
//     the same bits again and puts every symbol at its place in the sorted list.  The code-length code itself (19 symbols of at
//     most 7 bits) lives entirely in registers (symbols 5 bits each in two words, bases a byte each in one).
//   * one loop for all lanes with a state per lane (header wanted / symbols / stored bytes / done): headers are read at the
//     lanes' common step, by all lanes that want one together -- zlib ends a block after 16 383 symbols, so the streams of a
//     BAM reach their second header at the same step and no lane waits for another's header.
// Measured (profiles/r05_inflate_w64.txt): configs[2]'s 46 500 blocks in 26 ms -- 727 wavefronts, all resident at once (2.8 a
// CU), each taking ~3 000 cycles a step for its 270 vector + 120 scalar instructions: the kernel's time is one wavefront's
// latency, and what the 512 bytes buy shows on BAMs of more blocks than a chip holds wavefronts.
namespace w64 {

constexpr int kW = 64;
// byte offsets in a lane's own LDS space: 512 bytes a stream, 32 KiB a workgroup -- five workgroups fill a CU's 160 KiB
constexpr int aSymLl = 0;       // u8 x 288: literal/length symbols sorted by (length, symbol), their low eight bits ...
constexpr int aSymHi = 288;     // ... and bit 8 (a length code or the end of the block), one bit a symbol: 9 dwords
constexpr int aSymD = 324;      // u8 x 32: distance symbols
constexpr int aBaseLl = 356;    // u16 x 16: per length, sorted position of its first symbol - its first code
constexpr int aBaseD = 388;     // u16 x 16
constexpr int aRing = 420;      // u32 x 8: the next 32 bytes of the stream
constexpr int aLit = 452;       // u32 x 4
constexpr int aTok = 468;       // u32 x 8
constexpr int kLaneBytes = 512;
constexpr int aCntLl = aTok;    // u16 x 16, while a table header is read (the token ring is empty then)
constexpr int aCntD = aLit;     // u8 x 16 (at most 30 distance codes; the literal ring is empty then)
constexpr uint32_t kRingW = 8, kFetchW = 4;
constexpr uint32_t kCommonStep = 4;      // the lanes send what they gathered and top their rings up every fourth step

struct Lane {
  static constexpr uint32_t kLitW = 4, kTokW = 8;
  lds_u8* base;      // the lane's byte 0
  __device__ __forceinline__ lds_u8* at(int a) const { return base + ((a >> 2) << 8) + (a & 3); }     // [word][lane]: 256 bytes a word row
  __device__ __forceinline__ lds_u8& b(int a) const { return *at(a); }
  __device__ __forceinline__ lds_u16& h(int a) const { return *reinterpret_cast<lds_u16*>(at(a)); }     // a even
  __device__ __forceinline__ lds_u32& w(int a) const { return *reinterpret_cast<lds_u32*>(at(a)); }     // a a multiple of four
  __device__ __forceinline__ lds_u32& r(uint32_t i) const { return w(aRing + 4 * (int)(i & (kRingW - 1u))); }
  __device__ __forceinline__ lds_u32& lit(uint32_t i) const { return w(aLit + 4 * (int)(i & (kLitW - 1u))); }
  __device__ __forceinline__ lds_u32& tok(uint32_t i) const { return w(aTok + 4 * (int)(i & (kTokW - 1u))); }
  // the literal/length symbol at sorted position pos (< 288)
  __device__ __forceinline__ uint32_t sym_ll(uint32_t pos) const {
    const uint32_t lo = b(aSymLl + (int)pos), hi = w(aSymHi + 4 * (int)(pos >> 5));
    return lo | (((hi >> (pos & 31u)) & 1u) << 8);
  }
  __device__ __forceinline__ void put_sym_ll(uint32_t pos, uint32_t sym) const {
    b(aSymLl + (int)pos) = (uint8_t)sym;
    if (sym & 256u) w(aSymHi + 4 * (int)(pos >> 5)) |= 1u << (pos & 31u);
  }
  __device__ __forceinline__ void clear_sym_hi() const {
    for (int k = 0; k < 9; ++k) w(aSymHi + 4 * k) = 0u;
  }
};

// (BitIn of the decoder above, over this layout, that can also go back to a bit it has passed)
struct Bits {
  const uint8_t* src;
  size_t clen;
  const uint32_t* wp;
  const uint8_t* fetch_end;   // no word is fetched from here on (the streams' buffer has 256 bytes of slack behind its last stream)
  unsigned long long buf;
  int n;
  uint32_t rd, wr, lead;
  uint32_t pre[kFetchW];
  bool on_the_way;
  __device__ __forceinline__ void fetch() {
#pragma unroll
    for (int k = 0; k < (int)kFetchW; ++k) pre[k] = wp[k];
    wp += kFetchW;
    on_the_way = true;
  }
  __device__ __forceinline__ void land(const Lane& L) {
#pragma unroll
    for (int k = 0; k < (int)kFetchW; ++k) L.r(wr + (uint32_t)k) = pre[k];
    wr += kFetchW;
    on_the_way = false;
  }
  __device__ __forceinline__ void open_at(const Lane& L, size_t byte) {
    const uint8_t* p = src + byte;
    buf = 0; n = 0;
    fetch_end = src + clen + 128;
    while ((reinterpret_cast<uintptr_t>(p) & 3u) && n < 32) {
      buf |= (unsigned long long)(*p++) << n;
      n += 8;
    }
    lead = (uint32_t)n + 8u * (uint32_t)byte;       // bits in front of the first aligned word, from the stream's first bit
    wp = reinterpret_cast<const uint32_t*>(p);
    rd = wr = 0u;
    fetch();
  }
  __device__ __forceinline__ void open(const Lane& L, const uint8_t* s, size_t len) { src = s; clen = len; open_at(L, 0); }
  // bits of the stream consumed so far, and the way back to such a position
  __device__ __forceinline__ unsigned long long consumed() const { return (unsigned long long)lead + 32ull * rd - (unsigned long long)n; }
  __device__ __forceinline__ void seek(const Lane& L, unsigned long long bit) {
    open_at(L, (size_t)(bit >> 3));
    refill(L);
    skip((int)(bit & 7ull));
  }
  __device__ __forceinline__ void top_up(const Lane& L) {
    if (on_the_way && wr - rd <= kRingW - kFetchW) land(L);
    if (!on_the_way && wr - rd <= kRingW - kFetchW && reinterpret_cast<const uint8_t*>(wp) < fetch_end) fetch();
  }
  // at least 32 valid bits behind this; wherever the lanes are NOT at their common step (table headers, stored bytes)
  __device__ __forceinline__ void refill(const Lane& L) {
    if (n <= 32) {
      if (rd == wr) {
        if (!on_the_way) fetch();
        land(L);
      }
      buf |= (unsigned long long)L.r(rd) << n;
      ++rd;
      n += 32;
    }
  }
  // ... and in the symbol loop: no branch.  The ring holds the word (the loop looks after that once a step: short())
  __device__ __forceinline__ void refill_fast(const Lane& L) {
    const bool need = n <= 32;
    const unsigned long long w = (unsigned long long)L.r(rd) << (need ? n : 0);
    buf |= need ? w : 0ull;
    rd += need ? 1u : 0u;
    n += need ? 32 : 0;
  }
  __device__ __forceinline__ bool short_of_words() const { return wr - rd < 2u; }      // (a step refills twice at most)
  __device__ __forceinline__ void emergency(const Lane& L) {
    if (!on_the_way && reinterpret_cast<const uint8_t*>(wp) < fetch_end) fetch();
    if (on_the_way && wr - rd <= kRingW - kFetchW) land(L);
  }
  __device__ __forceinline__ uint32_t peek(int k) const { return (uint32_t)buf & ((1u << k) - 1u); }
  __device__ __forceinline__ void skip(int k) { buf >>= k; n -= k; }
  __device__ __forceinline__ uint32_t take(int k) { const uint32_t v = peek(k); skip(k); return v; }
  __device__ __forceinline__ bool overrun() const { return consumed() > 8ull * (unsigned long long)clen; }
};

// The limits of an alphabet as the symbol loop compares them: limit - 1 as 16-bit halves, two to a register (lengths 1|2, 3|4
// ... 15|none), so that one packed subtraction answers two compares and no condition code is involved.
typedef short pk16 __attribute__((ext_vector_type(2)));
struct Limits {
  pk16 p[8];
  __device__ __forceinline__ void pack(const uint32_t (&lim)[16]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t lo = (lim[2 * j + 1] - 1u) & 0xFFFFu;
      const uint32_t hi = 2 * j + 2 < 16 ? (lim[2 * j + 2] - 1u) & 0xFFFFu : 0x7FFFu;      // (a limit no value reaches)
      p[j].x = (short)lo; p[j].y = (short)hi;
    }
  }
  // the length of the code the 15 bits v begin with: 1 + the number of limits v has reached (16: no code)
  __device__ __forceinline__ uint32_t length(uint32_t v) const {
    pk16 vv; vv.x = (short)v; vv.y = (short)v;
    pk16 acc = (p[0] - vv) >> 15;            // -1 where limit - 1 - v < 0, i.e. v >= limit
#pragma unroll
    for (int j = 1; j < 8; ++j) acc += (p[j] - vv) >> 15;
    return (uint32_t)(1 - (int)acc.x - (int)acc.y);
  }
};
// From the counts per length (u16 x 16 at cnt_at): the limits (registers), the bases (LDS), and in the counts' place the
// sorted position of every length's first symbol.  False: over-subscribed, or incomplete (allowed as the one-code case zlib
// allows, as an alphabet without any code -- the distances of a literal-only block -- and where the format itself is: fixed).
template <bool CNT8>
__device__ __forceinline__ bool limits_from_counts(const Lane& L, int cnt_at, int base_at, uint32_t (&lim)[16], bool fixed) {
  int left = 1;
  uint32_t code = 0, off = 0;
  bool ok = true;
  lim[0] = 0;
#pragma unroll
  for (int l = 1; l < 16; ++l) {
    const uint32_t c = CNT8 ? (uint32_t)L.b(cnt_at + l) : (uint32_t)L.h(cnt_at + 2 * l);
    left = (left << 1) - (int)c;
    ok = ok && left >= 0;
    lim[l] = (code + c) << (15 - l);
    L.h(base_at + 2 * l) = (uint16_t)(off - code);
    if (CNT8) L.b(cnt_at + l) = (uint8_t)off; else L.h(cnt_at + 2 * l) = (uint16_t)off;
    off += c;
    code = (code + c) << 1;
  }
  const bool one = off == 1u && lim[1] == (1u << 14);        // a single code of one bit
  return ok && (left == 0 || one || off == 0u || fixed);
}
// One symbol of an alphabet: its code's length, 0 when the next bits are no code.  (sym8: the distance alphabet's bytes)
template <bool SYM8>
__device__ __forceinline__ int decode(const Lane& L, const Bits& in, const Limits& lim, int base_at, int sym_at, uint32_t* sym) {
  const uint32_t v = __brev((uint32_t)in.buf) >> 17;
  uint32_t l = lim.length(v);
  const bool bad = l > 15u;
  l = bad ? 15u : l;
  const uint32_t idx = ((v >> (15u - l)) + (uint32_t)L.h(base_at + 2 * (int)l)) & 0xFFFFu;
  // (a position beyond the table can only come from a value that is no code)
  *sym = SYM8 ? (uint32_t)L.b(sym_at + (int)(idx & 31u)) : L.sym_ll(idx < 288u ? idx : 0u);
  return bad ? 0 : (int)l;
}

// The code-length code of a dynamic header, in registers.
struct ClCode {
  uint32_t lim[8];                  // limits over the next SEVEN bits
  unsigned long long base;          // per length, a byte: sorted position of its first symbol - its first code
  unsigned long long sym_a, sym_b;  // sorted symbols, five bits each: positions 0..11, 12..18
  __device__ __forceinline__ bool build(unsigned long long lens3) {      // lens3: 3 bits per symbol 0..18
    uint32_t cnt[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) cnt[l] = 0;
    for (int s = 0; s < 19; ++s) {
      const uint32_t l = (uint32_t)(lens3 >> (3 * s)) & 7u;
#pragma unroll
      for (int q = 1; q < 8; ++q) cnt[q] += l == (uint32_t)q ? 1u : 0u;
    }
    int left = 1;
    uint32_t code = 0, off = 0;
    unsigned long long offs = 0;      // per length, a byte: where its next symbol goes
    bool ok = true;
    base = 0;
    lim[0] = 0;
#pragma unroll
    for (int l = 1; l < 8; ++l) {
      const uint32_t c = cnt[l];
      left = (left << 1) - (int)c;
      ok = ok && left >= 0;
      lim[l] = (code + c) << (7 - l);
      base |= (unsigned long long)((off - code) & 255u) << (8 * l);
      offs |= (unsigned long long)off << (8 * l);
      off += c;
      code = (code + c) << 1;
    }
    const bool one = off == 1u && cnt[1] == 1u;
    if (!(ok && (left == 0 || one))) return false;
    sym_a = 0; sym_b = 0;
    for (int s = 0; s < 19; ++s) {
      const uint32_t l = (uint32_t)(lens3 >> (3 * s)) & 7u;
      if (!l) continue;
      const uint32_t pos = (uint32_t)(offs >> (8 * l)) & 255u;
      offs += 1ull << (8 * l);
      if (pos < 12u) sym_a |= (unsigned long long)s << (5 * pos); else sym_b |= (unsigned long long)s << (5 * (pos - 12u));
    }
    return true;
  }
  // a code-length symbol (0..18), -1: no code; the bits are taken
  __device__ __forceinline__ int decode(Bits& in) const {
    const uint32_t v = __brev((uint32_t)in.buf) >> 25;
    uint32_t l = 1;
#pragma unroll
    for (int k = 1; k < 8; ++k) l += v >= lim[k] ? 1u : 0u;
    if (l > 7u) return -1;
    const uint32_t pos = ((v >> (7u - l)) + ((uint32_t)(base >> (8 * l)) & 255u)) & 255u;
    if (pos >= 19u) return -1;
    in.skip((int)l);
    return (int)((pos < 12u ? (sym_a >> (5 * pos)) : (sym_b >> (5 * (pos - 12u)))) & 31ull);
  }
};

struct Tables {
  Limits ll, d;                 // of the block in work
};

// One pass over the code lengths of a dynamic header (hlit + hdist of them, run-length coded): COUNT them per alphabet and
// length, or PLACE every symbol in its alphabet's sorted list.
template <bool PLACE>
__device__ __forceinline__ uint32_t lengths_pass(const Lane& L, Bits& in, const ClCode& cl, int hlit, int hdist, bool* has_eob) {
  int i = 0, prev = 0;
  const int total = hlit + hdist;
  while (i < total) {
    in.refill(L);
    const int s = cl.decode(in);
    if (s < 0) return kBadCodeLengths;
    int rep = 1, val = s;
    if (s == 16) { if (i == 0) return kBadCodeLengths; val = prev; rep = 3 + (int)in.take(2); }
    else if (s == 17) { val = 0; rep = 3 + (int)in.take(3); }
    else if (s == 18) { val = 0; rep = 11 + (int)in.take(7); }
    if (i + rep > total) return kBadCodeLengths;
    if (val) {
      for (int k = 0; k < rep; ++k) {
        const int sy = i + k;
        const bool isd = sy >= hlit;
        uint32_t c;
        if (isd) { c = L.b(aCntD + val); L.b(aCntD + val) = (uint8_t)(c + 1u); }
        else { c = L.h(aCntLl + 2 * val); L.h(aCntLl + 2 * val) = (uint16_t)(c + 1u); }
        if (PLACE) {
          if (isd) L.b(aSymD + (int)(c & 31u)) = (uint8_t)(sy - hlit);
          else L.put_sym_ll(c < 288u ? c : 0u, (uint32_t)sy);
        } else if (sy == 256) {
          *has_eob = true;
        }
      }
    }
    i += rep;
    prev = val;
    if (in.overrun()) return kInputOverrun;
  }
  return kOk;
}

__device__ __forceinline__ uint32_t dynamic_tables(const Lane& L, Bits& in, Tables& T) {
  in.refill(L);
  const int hlit = (int)in.take(5) + 257, hdist = (int)in.take(5) + 1, hclen = (int)in.take(4) + 4;
  if (hlit > 286 || hdist > 30) return kBadCodeLengths;
  unsigned long long lens3 = 0;
  for (int i = 0; i < hclen; ++i) {
    in.refill(L);
    lens3 |= (unsigned long long)in.take(3) << (3 * c_cl_order[i]);
  }
  ClCode cl;
  if (!cl.build(lens3)) return kBadCodeLengths;
  const unsigned long long mark = in.consumed();
  for (int l = 0; l < 16; ++l) { L.h(aCntLl + 2 * l) = 0; L.b(aCntD + l) = 0; }
  L.clear_sym_hi();
  bool has_eob = false;
  uint32_t st = lengths_pass<false>(L, in, cl, hlit, hdist, &has_eob);
  if (st != kOk) return st;
  if (!has_eob) return kBadCodeLengths;                  // no end-of-block code
  uint32_t lim[16];
  if (!limits_from_counts<false>(L, aCntLl, aBaseLl, lim, false)) return kBadCodeLengths;
  T.ll.pack(lim);
  if (!limits_from_counts<true>(L, aCntD, aBaseD, lim, false)) return kBadCodeLengths;
  T.d.pack(lim);
  const unsigned long long end = in.consumed();
  in.seek(L, mark);                                      // the same bits again: every symbol to its place
  st = lengths_pass<true>(L, in, cl, hlit, hdist, &has_eob);
  if (st != kOk) return st;
  return in.consumed() == end ? kOk : kBadCodeLengths;
}

__device__ __forceinline__ uint32_t fixed_tables(const Lane& L, Tables& T) {
  for (int l = 0; l < 16; ++l) { L.h(aCntLl + 2 * l) = 0; L.b(aCntD + l) = 0; }
  L.clear_sym_hi();
  L.h(aCntLl + 2 * 7) = 24; L.h(aCntLl + 2 * 8) = 152; L.h(aCntLl + 2 * 9) = 112;
  L.b(aCntD + 5) = 30;                               // (30 codes of 5 bits: incomplete, as the format defines it)
  uint32_t lim[16];
  if (!limits_from_counts<false>(L, aCntLl, aBaseLl, lim, true)) return kBadCodeLengths;
  T.ll.pack(lim);
  if (!limits_from_counts<true>(L, aCntD, aBaseD, lim, true)) return kBadCodeLengths;
  T.d.pack(lim);
  for (int s = 0; s < 288; ++s) {
    const int l = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
    const uint32_t c = L.h(aCntLl + 2 * l);
    L.h(aCntLl + 2 * l) = (uint16_t)(c + 1u);
    L.put_sym_ll(c, (uint32_t)s);
  }
  for (int s = 0; s < 30; ++s) L.b(aSymD + s) = (uint8_t)s;
  return kOk;
}

enum : uint32_t { kWantHeader = 0, kSymbols = 1, kStored = 2, kDone = 3 };

__global__ __launch_bounds__(kW) void bgzf_decode_kernel(InflateParams p) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[kLaneBytes / 4 * kW];
  const long long k = (long long)blockIdx.x * kW + threadIdx.x;
  const bool have = k < p.n_blocks;
  Lane L{reinterpret_cast<lds_u8*>((lds_u32*)lds) + 4 * (int)threadIdx.x};
  InflateBlock b{};
  if (have) b = p.blocks[k];
  Bits in;
  TokenOut<Lane> out;
  Tables T;
  {
    uint32_t none[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) none[l] = 0;
    T.ll.pack(none); T.d.pack(none);
  }
  uint32_t state = kDone, status = kOk, last = 0, stored_left = 0;
  const uint32_t ulen = b.ulen;
  out.open(have ? reinterpret_cast<uint32_t*>(p.matches + b.mbase) : nullptr, b.mcap * 2u);
  if (have && ulen) {
    if (out.room < 8u) { status = kMatchRoom; }
    else { in.open(L, p.comp + b.cpos, (size_t)b.clen); state = kWantHeader; }
  } else {
    in.src = nullptr; in.clen = 0; in.wp = nullptr; in.fetch_end = nullptr; in.buf = 0; in.n = 0; in.rd = in.wr = in.lead = 0; in.on_the_way = false;
  }
  auto fail = [&](uint32_t code) { status = code; state = kDone; };
  uint32_t iter = 0;
  while (__ballot(state != kDone) != 0ull) {
    if ((iter & (kCommonStep - 1u)) == 0u) {
      // (the words fetched eight steps ago go into the ring BEFORE this step's stores are issued: the wait in front of them
      // would otherwise wait for those stores as well)
      if (state != kDone) { in.top_up(L); out.service(L); }
      if (state == kWantHeader) {                          // all lanes that want one, together
        // (the tokens' and the literals' rings lend their places to the counts: what they hold goes out first)
        while (out.tr < out.tw) { out.area[2u + out.tr] = L.tok(out.tr); ++out.tr; }
        while (out.lr < out.lw) { out.area[out.room - 1u - out.lr] = L.lit(out.lr); ++out.lr; }
        in.refill(L);
        last = in.take(1);
        const uint32_t type = in.take(2);
        if (type == 3u) {
          fail(kBadBlockType);
        } else if (type == 0u) {                           // stored: to the next byte, LEN, ~LEN, the bytes
          in.skip(in.n & 7);
          in.refill(L);
          const uint32_t len = in.take(16);
          in.refill(L);
          const uint32_t nlen = in.take(16);
          if ((len ^ 0xFFFFu) != nlen) fail(kBadStored);
          else if (out.o + len > ulen) fail(kOutputOverrun);
          else { stored_left = len; state = len ? kStored : (last ? kDone : kWantHeader); }
        } else {
          const uint32_t st = type == 1u ? fixed_tables(L, T) : dynamic_tables(L, in, T);
          if (st != kOk) fail(st); else { in.refill(L); state = kSymbols; }
        }
        if (state != kDone && in.overrun()) fail(kInputOverrun);
      }
    }
    ++iter;
    // (a lane whose ring is about to run dry -- a stretch of long matches with many extra bits -- gets its words here, in a
    // branch the wavefront rarely takes; the refills below then never wait)
    if (__ballot(state == kSymbols && in.short_of_words()) != 0ull) {
      if (state == kSymbols && in.short_of_words()) in.emergency(L);
    }
    if (state == kSymbols) {
      // One symbol, and where it is a length code the distance behind it -- written so that a wavefront whose lanes meet
      // literals AND matches at every step (64 streams: always) runs ONE instruction sequence: both halves are computed by every
      // lane, a lane's kind decides what is committed.  (As nested branches the same step spent a third of its time on the
      // scalar unit moving execution masks.)
      uint32_t s;
      const int l = decode<false>(L, in, T.ll, aBaseLl, aSymLl, &s);
      in.skip(l);
      in.refill_fast(L);
      const bool good = l != 0;
      const bool is_lit = good && s < 256u, is_eob = good && s == 256u, is_len = good && s > 256u;
      const int sl = (int)s - 257;
      const bool sl_ok = sl < 29;
      uint32_t len, dist, d;
      int extra, dextra;
      length_of(is_len && sl_ok ? sl : 0, &len, &extra);
      len += in.take(is_len ? extra : 0);                  // (<= 5 extra bits of >= 32: 27 left for the distance code)
      const int dl0 = decode<true>(L, in, T.d, aBaseD, aSymD, &d);
      in.skip(is_len ? dl0 : 0);
      in.refill_fast(L);
      distance_of(d < 30u ? (int)d : 0, &dist, &dextra);
      dist += in.take(is_len ? dextra : 0);
      uint32_t err = good ? 0u : (uint32_t)kBadSymbol;
      err = is_len && !sl_ok ? (uint32_t)kBadSymbol : err;
      err = !err && is_len && (dl0 == 0 || d >= 30u || dist > out.o) ? (uint32_t)kBadDistance : err;
      err = !err && ((is_lit && out.o >= ulen) || (is_len && out.o + len > ulen)) ? (uint32_t)kOutputOverrun : err;
      out.step(L, is_lit && !err, s, is_len && !err, len, dist);
      state = is_eob ? (last ? (uint32_t)kDone : (uint32_t)kWantHeader) : state;
      if (!err && in.overrun()) err = kInputOverrun;
      if (!err && out.full) err = kMatchRoom;
      if (err) fail(err);
    } else if (state == kStored) {
      for (int q = 0; q < 4 && stored_left; ++q) {
        in.refill(L);
        out.literal(L, in.take(8));
        --stored_left;
      }
      if (in.overrun()) fail(kInputOverrun);
      else if (out.full) fail(kMatchRoom);
      else if (!stored_left) state = last ? kDone : kWantHeader;
    }
  }
  if (!have) return;
  uint32_t nt = 0;
  if (status == kOk && ulen) {
    out.finish(L);
    if (out.full) status = kMatchRoom;
    else if (out.o != ulen) status = kShortOutput;
    nt = out.tw;
  }
  p.status[k] = status;
  p.n_matches[k] = status == kOk ? nt : 0u;
}

}  // namespace w64

// ---- the placer ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)v, d);
    if (lane >= d) v += o;
  }
  return v;
}
// n (1..8) bytes of v to an address of any alignment: two overlapping dwords from four bytes on
__device__ __forceinline__ void store_upto8(uint8_t* dst, unsigned long long v, uint32_t n) {
  if (n >= 8u) {
    *reinterpret_cast<u64_a1*>(dst) = v;
  } else if (n >= 4u) {
    *reinterpret_cast<u32_a1*>(dst) = (uint32_t)v;
    *reinterpret_cast<u32_a1*>(dst + (n - 4u)) = (uint32_t)(v >> (8u * (n - 4u)));
  } else {
    if (n >= 2u) *reinterpret_cast<u16_a1*>(dst) = (uint16_t)v;
    if (n & 1u) dst[n - 1u] = (uint8_t)(v >> (8u * (n - 1u)));
  }
}
__device__ __forceinline__ unsigned long long bswap64(unsigned long long v) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  return ((unsigned long long)__builtin_bswap32(lo) << 32) | (unsigned long long)__builtin_bswap32(hi);
}

// eight bytes of the sequence with period `dist` (1..7 bytes, in the low bytes of pat), from its byte `phase` on
__device__ __forceinline__ unsigned long long periodic8(unsigned long long pat, uint32_t dist, uint32_t phase) {
  const uint32_t bits = 8u * dist;
  unsigned long long w = phase ? ((pat >> (8u * phase)) | (pat << (bits - 8u * phase))) & ((1ull << bits) - 1ull) : pat;
  w |= w << bits;                          // 2 * dist bytes (dist <= 7: bits <= 56)
  if (2u * bits < 64u) w |= w << (2u * bits);
  if (4u * bits < 64u) w |= w << (4u * bits);
  return w;
}

// 64 tokens of a block, one a lane: where each one's bytes go
struct Window {
  uint32_t lits, len, dist;   // len 0: no match (an escape token, or a lane behind the block's last token)
  uint32_t start;             // the token's first byte in the block's output (its literals), the match follows at start + lits
  uint32_t lsrc;              // its first literal's index in the block's literal stream
  uint32_t end_o, end_l;      // (all lanes) what the window's tokens add up to, the tokens in front of it included
};
__device__ __forceinline__ Window make_window(uint32_t t, uint32_t carry_o, uint32_t carry_l, int lane) {
  Window w;
  const bool esc = (t >> 23) == 511u;
  w.lits = esc ? (t & 0x7FFFFFu) : (t >> 23);
  w.len = esc ? 0u : ((t >> 15) & 255u) + 3u;
  w.dist = (t & 0x7FFFu) + 1u;
  const uint32_t io = wave_scan_incl(w.lits + w.len, lane), il = wave_scan_incl(w.lits, lane);
  w.start = carry_o + io - (w.lits + w.len);
  w.lsrc = carry_l + il - w.lits;
  w.end_o = carry_o + (uint32_t)__shfl((int)io, 63);
  w.end_l = carry_l + (uint32_t)__shfl((int)il, 63);
  return w;
}
// the window's literals from the block's literal stream (bytes in reverse order below lit_end) to their places
__device__ __forceinline__ void place_literals(const Window& w, uint8_t* out, const uint8_t* lit_end, int lane) {
  const uint32_t n = w.lits;
  uint8_t* dst = out + w.start;
  if (n > 0u && n <= 64u) {
    uint32_t k = 0;
    for (; k + 8u <= n; k += 8u)
      *reinterpret_cast<u64_a1*>(dst + k) = bswap64(*reinterpret_cast<const u64_a1*>(lit_end - (w.lsrc + k) - 8u));
    if (k < n) {
      if (n >= 8u) {          // the last eight once more (the same bytes)
        *reinterpret_cast<u64_a1*>(dst + (n - 8u)) = bswap64(*reinterpret_cast<const u64_a1*>(lit_end - (w.lsrc + n)));
      } else {
        store_upto8(dst, bswap64(*reinterpret_cast<const u64_a1*>(lit_end - w.lsrc - 8u)), n);
      }
    }
  }
  // a long run of literals (a stored block, incompressible bytes): the whole wavefront copies it, a byte a lane
  unsigned long long big = __ballot(n > 64u);
  while (big) {
    const int src = (int)__ffsll((long long)big) - 1;
    big &= big - 1ull;
    const uint32_t bn = (uint32_t)__shfl((int)n, src), bs = (uint32_t)__shfl((int)w.start, src), bl = (uint32_t)__shfl((int)w.lsrc, src);
    for (uint32_t t = (uint32_t)lane; t < bn; t += 64u) out[bs + t] = *(lit_end - 1 - (bl + t));
  }
}

constexpr int kResolveWaves = 4;
__global__ __launch_bounds__(kLanes * kResolveWaves) void bgzf_place_kernel(InflateParams p) {
  const long long k = (long long)blockIdx.x * kResolveWaves + (threadIdx.x >> 6);
  if (k >= p.n_blocks) return;
  if (p.status[k] != kOk) return;
  const int lane = (int)(threadIdx.x & 63u);
  const uint32_t n = p.n_matches[k];
  if (n == 0u) return;
  const InflateBlock b = p.blocks[k];
  uint8_t* out = p.out + b.upos;
  const uint32_t* area = reinterpret_cast<const uint32_t*>(p.matches + b.mbase);
  const uint32_t* toks = area + 2;
  const uint8_t* lit_end = reinterpret_cast<const uint8_t*>(area + 2u * (size_t)b.mcap);
  auto fetch = [&](uint32_t base) -> uint32_t { const uint32_t i = base + (uint32_t)lane; return i < n ? toks[i] : kEscape; };
  uint32_t t_next = fetch(64u);
  Window cur = make_window(fetch(0u), 0u, 0u, lane);
  place_literals(cur, out, lit_end, lane);
  for (uint32_t base = 0; base < n; base += 64u) {
    // the next window: its tokens arrived while the last one was resolved; its literals go out now, in front of this window's
    // matches (every step below starts with a fence: they are in place, and visible, before this wavefront reads them)
    const Window nxt = make_window(t_next, cur.end_o, cur.end_l, lane);
    t_next = fetch(base + 128u);
    if (base + 64u < n) place_literals(nxt, out, lit_end, lane);
    unsigned long long todo = __ballot(cur.len > 0u);
    const uint32_t o = cur.start + cur.lits;
    const uint32_t span = cur.len < cur.dist ? cur.len : cur.dist;
    // Which of this window's tokens a match waits for: those whose bytes its source [a, bnd) touches.  The tokens' bytes lie
    // back to back in token order, so that is a run of lanes, found by two binary searches over the lanes' `start` (shuffles);
    // a source that ends in front of the window waits for nothing.  A match is copied as soon as none of the tokens it waits
    // for still has its own match to copy (their literals are in place already): the steps of a window are the longest chain
    // of matches that feed one another, not the number of runs that happen to sit in file order.
    unsigned long long dep = 0;
    {
      const uint32_t a = o - cur.dist, bnd = a + span;
      const uint32_t wstart = (uint32_t)__shfl((int)cur.start, 0);
      uint32_t jlo = 0, jhi = 0;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t c1 = jlo + (uint32_t)d, c2 = jhi + (uint32_t)d;
        const uint32_t s1 = (uint32_t)__shfl((int)cur.start, (int)(c1 & 63u)), s2 = (uint32_t)__shfl((int)cur.start, (int)(c2 & 63u));
        if (c1 < 64u && s1 <= a) jlo = c1;
        if (c2 < 64u && s2 <= bnd - 1u) jhi = c2;
      }
      if (cur.len > 0u && bnd > wstart) {
        const uint32_t lo = a >= wstart ? jlo : 0u;
        dep = (jhi >= 63u ? ~0ull : ((2ull << jhi) - 1ull)) & ~((1ull << lo) - 1ull) & ~(1ull << lane);
      }
    }
    while (todo) {
      // (this wavefront's stores in front of its loads.  The same wavefront, the same CU's cache: a workgroup-scope fence.  An
      // agent-scope __threadfence() writes the XCD's whole L2 back -- 150 us a time here.)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      const bool mine = ((todo >> lane) & 1ull) != 0ull && (todo & dep) == 0ull;
      if (mine) {
        uint8_t* dst = out + o;
        const uint8_t* s = dst - cur.dist;
        const uint32_t len = cur.len;
        if (cur.dist >= 8u || cur.dist >= len) {
          // eight bytes a time.  Source and destination apart (nearly every match), or at least eight bytes apart: a load
          // then reads what this lane itself stored an iteration earlier at the latest (a lane's accesses keep their order)
          if (len <= 8u) {
            store_upto8(dst, *reinterpret_cast<const u64_a1*>(s), len);
          } else {
            uint32_t q = 0;
            for (; q + 8u <= len; q += 8u) *reinterpret_cast<u64_a1*>(dst + q) = *reinterpret_cast<const u64_a1*>(s + q);
            if (q < len) *reinterpret_cast<u64_a1*>(dst + (len - 8u)) = *reinterpret_cast<const u64_a1*>(s + (len - 8u));
          }
        } else {
          // a match longer than its distance of fewer than eight bytes repeats its first `distance` bytes (a run of one
          // quality, of zeros): the period is built in registers, nothing is read back
          const uint32_t dist = cur.dist;
          const unsigned long long pat = *reinterpret_cast<const u64_a1*>(s) & ((1ull << (8u * dist)) - 1ull);
          uint32_t q = 0, phase = 0;
          for (; q < len; q += 8u) {
            const unsigned long long w = periodic8(pat, dist, phase);
            const uint32_t left = len - q;
            if (left >= 8u) *reinterpret_cast<u64_a1*>(dst + q) = w; else store_upto8(dst + q, w, left);
            phase = (phase + 8u) % dist;
          }
        }
      }
      todo &= ~__ballot(mine);
    }
    cur = nxt;
  }
}

// The payload columns cut out of the inflated stream where it lies, in HBM: HALF a wavefront per record copies the record's
// CIGAR ops, 4-bit SEQ and QUAL (BAM record layout: SAM spec 4.2) to the offsets the size pass computed, eight bytes a lane
// and load (any alignment), the last few bytes singly -- a 150 bp record is 30 lanes' worth, two records share a wavefront's
// trip to memory.  What the host decoder does with three memcpy per record -- and then sends up the link again.
__device__ __forceinline__ void copy_run32(uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t sl) {
  const uint32_t whole = n & ~7u;
  for (uint32_t k = sl * 8u; k < whole; k += 256u) *reinterpret_cast<u64_a1*>(dst + k) = *reinterpret_cast<const u64_a1*>(src + k);
  if (sl < (n & 7u)) dst[whole + sl] = src[whole + sl];
}
__global__ __launch_bounds__(256) void bam_payload_kernel(PayloadParams p) {
  const uint32_t lane = threadIdx.x & 63u, sub = lane >> 5, sl = lane & 31u;
  const long long slot = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + sub, n_slots = (long long)gridDim.x * 8;
  for (long long i = slot; i < p.n_records; i += n_slots) {
    uint32_t n_cig, l;
    const uint8_t* q;
    if (p.drec) {        // out of the direct layout: the same run of bytes, found by the read's record
      n_cig = (uint32_t)(p.cigar_off[i + 1] - p.cigar_off[i]);
      l = (uint32_t)(p.qual_off[i + 1] - p.qual_off[i]);
      q = p.stream + ((unsigned long long)p.drec[i].off8 << 3);
    } else {
      const uint8_t* r = p.stream + p.rec_off[i];
      const uint32_t l_name = r[12];
      n_cig = (uint32_t)r[16] | ((uint32_t)r[17] << 8);
      l = (uint32_t)r[20] | ((uint32_t)r[21] << 8) | ((uint32_t)r[22] << 16) | ((uint32_t)r[23] << 24);
      q = r + 36 + l_name;
    }
    copy_run32(reinterpret_cast<uint8_t*>(p.cigar + p.cigar_off[i]), q, 4u * n_cig, sl);
    q += 4ull * n_cig;
    const uint32_t ns = (l + 1u) / 2u;
    copy_run32(p.seq4 + p.seq_off[i], q, ns, sl);
    q += ns;
    copy_run32(p.qual + p.qual_off[i], q, l, sl);
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

// How many BGZF blocks fill the device ONCE with the decoder's workgroups (a lane a block, kW lanes a workgroup, as many
// workgroups a CU as its LDS holds): the decoder is latency-bound, a launch takes about as long for one such wave of blocks as
// for a tenth of it -- what a streamed decode's groups are sized by (snps_abi.hip device_decode_stream).
long long bgzf_inflate_wave_blocks(int n_cu) {
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, w64::bgzf_decode_kernel, w64::kW, 0) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 4; }
  return (long long)occ * (n_cu > 0 ? n_cu : 256) * w64::kW;
}

hipError_t launch_bgzf_inflate(const InflateParams& p, hipStream_t s, int phases) {
  if (p.n_blocks <= 0) return hipSuccess;
  if (phases & 1) {
    // the block inflater keeps its streams' tables and rings in LDS (2.3 KiB a stream). A device that cannot give a
    // workgroup that much cannot run it; say so instead of leaving it to the launch (the callers fall back to the host's inflater).
    static const bool fits = [] {
      hipFuncAttributes fa{};
      int dev = 0, lds = 0;
      if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(w64::bgzf_decode_kernel)) != hipSuccess) return true;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return true;
      return fa.sharedSizeBytes <= (size_t)lds;
    }();
    if (!fits) return hipErrorLaunchOutOfResources;
    const long long g = (p.n_blocks + w64::kW - 1) / w64::kW;
    if (getenv("MIDAS_SNPS_TRACE")) {
      int occ = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, w64::bgzf_decode_kernel, w64::kW, 0) == hipSuccess)
        fprintf(stderr, "[device decode] decoder: %lld workgroups of %d streams, %d resident a CU\n", g, w64::kW, occ);
    }
    hipLaunchKernelGGL(w64::bgzf_decode_kernel, dim3((unsigned)g), dim3(w64::kW), 0, s, p);
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
  hipLaunchKernelGGL(bgzf_place_kernel, dim3((unsigned)g2), dim3(kLanes * kResolveWaves), 0, s, p);
  if (p.want_crc && !(phases & 8)) {       // (8: a caller that times the phases launches the check by itself, phases = 4)
    long long g3 = (p.n_blocks + kCrcWaves - 1) / kCrcWaves;
    g3 = g3 > 2048 ? 2048 : g3;
    hipLaunchKernelGGL(bgzf_crc_kernel, dim3((unsigned)g3), dim3(kLanes * kCrcWaves), 0, s, p);
  }
  return hipGetLastError();
}

}  // namespace midas
