// The rows of <species>.snps.gz, formatted AND deflated on the device: one workgroup per gzip member (16 384 rows).
//
// Replaces, for the device results of a batch, the per-site emit loop of midas/run/snps.py:201-210 and the gzip writer
// behind utility.iopen(..., 'w') (midas/utility.py:194-206).  What leaves the device is the DEFLATE stream (about 4 bytes a
// row) instead of the 17 bytes a site of counts + allele, and the host neither formats nor compresses: it frames the
// members (gzip header, CRC-32, ISIZE) and writes them.
//
// The coder is the row coder of row_deflate.h re-thought for a workgroup.  A row is
//     <ref_id> \t <ref_pos> \t <ref_allele> \t <depth> \t <count_a> \t <count_c> \t <count_g> \t <count_t> \n
// = head (id, tab, position) + tail (from the tab before the allele through the newline).  The tail of a row repeats the
// tail of some recent row with the same (allele, counts); the head repeats the head of the row before but for a digit or
// two.  The host coder finds "the latest earlier row with this tail" with a hash table it updates row by row; a workgroup
// gets the same answer for all rows at once by SORTING (hash of the tail, row number) in LDS -- a row's predecessor in the
// sorted order is the nearest earlier row with its hash, verified against the row's numbers.  Tokens per row:
//     literals for the head's bytes that the match of the row before does not cover,
//     one match = the row's tail + as much of the NEXT row's head as agrees with the head behind the matched tail
//     (no such row within 32 KiB: the tail as literals, then a match of the next head against this row's head).
// Nothing of the text is ever stored: row lengths, token lengths and the literal bytes are recomputed from the row's five
// numbers wherever they are needed (three passes: symbol histogram, bit lengths, emission + CRC).  One dynamic-Huffman block
// per member (RFC 1951 3.2.7; code construction as in row_deflate.cpp, its serial parts on one thread, the rank sort and
// the canonical codes by all threads).  Every thread owns 32 consecutive rows and writes their bits where the prefix sum of
// the bit lengths puts them; the two words a thread may share with its neighbours are OR-ed in atomically.
// The CRC-32 of the member's text: every thread runs the table-driven CRC over its rows' bytes from a zero register, shifts
// it to the end of the text (multiplication by x^(8 * bytes behind it) mod P, square-and-multiply) and the XOR of all of
// them plus the shifted initial register is the CRC.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace midas {
namespace {

constexpr int kT = 512;                 // threads per workgroup
constexpr int kRowsMax = 16384;         // rows per member (hostio.h, kRowsPerMember)
constexpr int kPer = kRowsMax / kT;     // rows (and sorted slots) per thread
constexpr int kMaxId = 192;             // longer contig ids: the member is left to the host coder
constexpr uint32_t kCrcPoly = 0xEDB88320u;

constexpr uint32_t kTokMatch = 1u << 31;
__device__ __forceinline__ uint32_t tok_make(bool has, uint32_t cover, uint32_t dist) {
  return (has ? kTokMatch : 0u) | (cover << 16) | (has ? dist - 1u : 0u);
}
__device__ __forceinline__ uint32_t tok_cover(uint32_t t) { return (t >> 16) & 511u; }
__device__ __forceinline__ uint32_t tok_dist(uint32_t t) { return (t & 32767u) + 1u; }

__device__ __forceinline__ int nd32(uint32_t v) {
  return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6
       : v < 10000000u ? 7 : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}
__device__ __forceinline__ int nd64(unsigned long long v) {       // v < 2^34 (a sum of four 32-bit counts)
  return v < 4294967296ull ? nd32((uint32_t)v) : (v < 10000000000ull ? 10 : 11);
}
// decimal digits, most significant in the lowest nibble
__device__ __forceinline__ unsigned long long rev_digits(unsigned long long v) {
  unsigned long long r = 0;
  do { r = (r << 4) | (v % 10ull); v /= 10ull; } while (v);
  return r;
}
__device__ __forceinline__ unsigned long long rev_digits32(uint32_t v) {
  unsigned long long r = 0;
  do { r = (r << 4) | (unsigned long long)(v % 10u); v /= 10u; } while (v);
  return r;
}
// leading decimal characters two positions have in common
__device__ __forceinline__ int common_digits(uint32_t a, uint32_t b) {
  const int na = nd32(a), nb = nd32(b);
  const unsigned long long x = rev_digits32(a) ^ rev_digits32(b);
  const int same = x ? (__ffsll((long long)x) - 1) >> 2 : 16;
  const int m = na < nb ? na : nb;
  return same < m ? same : m;
}

struct Row {
  uint32_t c0, c1, c2, c3, allele, pos;
  unsigned long long depth;
  int n_pos, n_tail;
};
__device__ __forceinline__ Row load_row(const RowsParams& p, const RowsMember& m, int r) {
  Row x;
  const uint4 c = *reinterpret_cast<const uint4*>(p.counts + 4 * (size_t)(m.site0 + r));
  x.c0 = c.x; x.c1 = c.y; x.c2 = c.z; x.c3 = c.w;
  x.allele = p.allele[(size_t)(m.site0 + r)];
  x.pos = (uint32_t)(m.pos0 + r);
  x.depth = (unsigned long long)c.x + c.y + c.z + c.w;
  x.n_pos = nd32(x.pos);
  x.n_tail = 8 + nd64(x.depth) + nd32(c.x) + nd32(c.y) + nd32(c.z) + nd32(c.w);
  return x;
}
__device__ __forceinline__ bool same_tail(const Row& a, const Row& b) {
  return a.c0 == b.c0 && a.c1 == b.c1 && a.c2 == b.c2 && a.c3 == b.c3 && a.allele == b.allele;
}
__device__ __forceinline__ uint32_t tail_hash(const Row& x) {
  unsigned long long h = 0x9E3779B97F4A7C15ull ^ x.allele;
  h = (h ^ x.c0) * 0xFF51AFD7ED558CCDull;
  h = (h ^ x.c1) * 0xFF51AFD7ED558CCDull;
  h = (h ^ (h >> 29) ^ x.c2) * 0xFF51AFD7ED558CCDull;
  h = (h ^ x.c3) * 0xC4CEB9FE1A85EC53ull;
  return (uint32_t)(h >> 46);        // 18 bits
}

// the tail's bytes, in order
template <class F>
__device__ __forceinline__ void tail_bytes(const Row& x, F f) {
  f((uint32_t)'\t'); f(x.allele);
  auto dec = [&](unsigned long long rv, int n) {
    f((uint32_t)'\t');
    for (int i = 0; i < n; ++i) { f((uint32_t)'0' + (uint32_t)(rv & 15ull)); rv >>= 4; }
  };
  dec(rev_digits(x.depth), nd64(x.depth));
  dec(rev_digits32(x.c0), nd32(x.c0));
  dec(rev_digits32(x.c1), nd32(x.c1));
  dec(rev_digits32(x.c2), nd32(x.c2));
  dec(rev_digits32(x.c3), nd32(x.c3));
  f((uint32_t)'\n');
}

// RFC 1951 3.2.5 in closed form: length 3..258 -> (code 0..28, extra bits, extra value); distance 1..32768 likewise
__device__ __forceinline__ void len_code(uint32_t L, uint32_t* code, uint32_t* eb, uint32_t* ev) {
  const uint32_t l = L - 3u;
  if (L == 258u) { *code = 28u; *eb = 0u; *ev = 0u; return; }
  if (l < 8u) { *code = l; *eb = 0u; *ev = 0u; return; }
  const uint32_t e = 31u - (uint32_t)__clz((int)l);
  *eb = e - 2u;
  *code = 4u * (e - 1u) + ((l >> *eb) & 3u);
  *ev = l & ((1u << *eb) - 1u);
}
__device__ __forceinline__ void dist_code(uint32_t D, uint32_t* code, uint32_t* eb, uint32_t* ev) {
  const uint32_t d = D - 1u;
  if (d < 4u) { *code = d; *eb = 0u; *ev = 0u; return; }
  const uint32_t e = 31u - (uint32_t)__clz((int)d);
  *eb = e - 1u;
  *code = 2u * e + ((d >> *eb) & 1u);
  *ev = d & ((1u << *eb) - 1u);
}

__device__ __forceinline__ uint32_t reverse_bits(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// ---- GF(2) arithmetic of the CRC (reflected: bit 31 is x^0) -----------------------------------------------------------------
__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int i = 0; i < 32; ++i) {
    p ^= (a & 0x80000000u) ? b : 0u;
    a <<= 1;
    b = (b >> 1) ^ ((b & 1u) ? kCrcPoly : 0u);
  }
  return p;
}
// x^(8 n) mod P; x2n[k] = x^(2^k) mod P
__device__ __forceinline__ uint32_t gf_xpow8(unsigned long long n, const uint32_t* x2n) {
  uint32_t p = 0x80000000u;
  int k = 3;
  while (n) {
    if (n & 1ull) p = gf_mul(x2n[k & 31], p);
    n >>= 1;
    ++k;
  }
  return p;
}

struct Shared {
  uint32_t key[kRowsMax];          // sort keys (hash << 14 | row), later the rows' tokens
  uint32_t off[kRowsMax + 1];      // byte offset of every row in the member's text, later ... bit offsets are per thread
  uint32_t hist_ll[288], hist_d[32], hist_cl[20];
  uint32_t code_ll[288], code_d[32], code_cl[20];     // bits | length << 16
  uint8_t len_ll[288], len_d[32], len_cl[20];
  uint32_t w[2 * 288];             // Huffman work: weights, then depths
  int16_t parent[2 * 288];
  int16_t order[288];
  uint32_t first_code[16], bl_count[16];
  uint32_t crc_tab[256], x2n[32];
  uint32_t scan[kT];
  uint32_t hdr[96];                // the block header's bits (<= 17 + 57 + 316 * 7 = 2286 bits)
  uint32_t hdr_bits, n_ll, n_d, n_cl, m_sorted;
  uint32_t crc;
  unsigned long long base;         // where the member's stream goes in the arena (bytes)
  uint32_t ok;
  uint8_t id[kMaxId + 8];
};

// Exclusive prefix sum over the workgroup's threads; returns this thread's base, *total = the sum.
__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t* scratch, uint32_t* total) {
  const int t = (int)threadIdx.x;
  scratch[t] = v;
  __syncthreads();
  for (int d = 1; d < kT; d <<= 1) {
    const uint32_t add = t >= d ? scratch[t - d] : 0u;
    __syncthreads();
    scratch[t] += add;
    __syncthreads();
  }
  const uint32_t incl = scratch[t];
  *total = scratch[kT - 1];
  __syncthreads();
  return incl - v;
}

// Code lengths of a prefix code for the symbols with freq > 0, none longer than max_len, Kraft sum exactly one: the
// construction of RowDeflate::build_lengths (row_deflate.cpp).  Called by every thread; the rank sort is everybody's work,
// the tree and the repair one thread's (thread `boss`).
__device__ void build_lengths(Shared& S, const uint32_t* freq, int n, int max_len, uint8_t* len_out, int boss) {
  const int t = (int)threadIdx.x;
  if (t == boss) S.m_sorted = 0u;
  __syncthreads();
  if (t < n) {
    len_out[t] = 0;
    const uint32_t f = freq[t];
    if (f) {
      int rank = 0;
      for (int s = 0; s < n; ++s) {
        const uint32_t g = freq[s];
        rank += (g != 0u && (g < f || (g == f && s < t))) ? 1 : 0;
      }
      S.order[rank] = (int16_t)t;
      atomicAdd(&S.m_sorted, 1u);
    }
  }
  __syncthreads();
  if (t == boss) {
    const int m = (int)S.m_sorted;
    if (m == 1) {
      len_out[S.order[0]] = 1;
    } else if (m > 1) {
      for (int i = 0; i < m; ++i) { S.w[i] = freq[S.order[i]]; S.parent[i] = -1; }
      int leaf = 0, inner = m, made = m;
      while ((m - leaf) + (made - inner) > 1) {
        int pick[2];
        for (int k = 0; k < 2; ++k) {     // the lighter of the next unused leaf and the next unused internal node
          if (leaf < m && (inner >= made || S.w[leaf] <= S.w[inner])) pick[k] = leaf++; else pick[k] = inner++;
        }
        S.w[made] = S.w[pick[0]] + S.w[pick[1]];
        S.parent[made] = -1;
        S.parent[pick[0]] = S.parent[pick[1]] = (int16_t)made;
        ++made;
      }
      S.w[made - 1] = 0u;
      for (int i = made - 2; i >= 0; --i) S.w[i] = S.w[S.parent[i]] + 1u;      // depths
      long long kraft = 0;
      const long long one = 1ll << max_len;
      for (int i = 0; i < m; ++i) {
        if (S.w[i] > (uint32_t)max_len) S.w[i] = (uint32_t)max_len;
        kraft += one >> S.w[i];
      }
      while (kraft > one) {
        int pick = -1;
        for (int i = 0; i < m; ++i)
          if ((int)S.w[i] < max_len && (pick < 0 || S.w[i] > S.w[pick])) pick = i;
        kraft -= one >> (S.w[pick] + 1u);
        S.w[pick] += 1u;
      }
      while (kraft < one) {
        int pick = -1;
        for (int i = m - 1; i >= 0; --i)
          if (S.w[i] > 1u && kraft + (one >> S.w[i]) <= one && (pick < 0 || S.w[i] > S.w[pick])) pick = i;
        if (pick < 0) break;
        kraft += one >> S.w[pick];
        S.w[pick] -= 1u;
      }
      for (int i = 0; i < m; ++i) len_out[S.order[i]] = (uint8_t)S.w[i];
    }
  }
  __syncthreads();
}

// RFC 1951 3.2.2: canonical codes from the lengths, stored bit-reversed (| length << 16).  Everybody's work.
__device__ void make_codes(Shared& S, const uint8_t* len, int n, uint32_t* codes) {
  const int t = (int)threadIdx.x;
  if (t < 16) S.bl_count[t] = 0u;
  __syncthreads();
  if (t < n && len[t]) atomicAdd(&S.bl_count[len[t]], 1u);
  __syncthreads();
  if (t == 0) {
    uint32_t code = 0;
    S.first_code[0] = 0u;
    for (int b = 1; b <= 15; ++b) {
      code = (code + (b > 1 ? S.bl_count[b - 1] : 0u)) << 1;
      S.first_code[b] = code;
    }
  }
  __syncthreads();
  if (t < n) {
    const int l = len[t];
    uint32_t v = 0;
    if (l) {
      uint32_t before = 0;
      for (int s = 0; s < t; ++s) before += len[s] == l ? 1u : 0u;
      v = reverse_bits(S.first_code[l] + before, l) | ((uint32_t)l << 16);
    }
    codes[t] = v;
  }
  __syncthreads();
}

// A thread's share of the member's bit stream: bits [bit0, ...) of the stream that starts at word `base`.
struct BitOut {
  uint32_t* base;
  unsigned long long acc;
  int fill;
  size_t word;
  bool first;
  __device__ __forceinline__ void init(uint32_t* b, unsigned long long bit0) {
    base = b; word = (size_t)(bit0 >> 5); fill = (int)(bit0 & 31ull); acc = 0ull; first = true;
  }
  __device__ __forceinline__ void put(uint32_t bits, int n) {       // n <= 28
    acc |= (unsigned long long)bits << fill;
    fill += n;
    if (fill >= 32) {
      if (first) { atomicOr(&base[word], (uint32_t)acc); first = false; } else base[word] = (uint32_t)acc;
      ++word;
      acc >>= 32;
      fill -= 32;
    }
  }
  __device__ __forceinline__ void finish() {
    if (fill > 0 && acc != 0ull) atomicOr(&base[word], (uint32_t)acc);
  }
};

// The tokens of row r, given its token word and the cover the row before left on its head.
//   lit(byte) / match(length, distance)
template <class L, class M>
__device__ __forceinline__ void row_tokens(const Shared& S, const RowsMember& m, const Row& x, uint32_t tok, uint32_t cover_prev,
                                           L lit, M match) {
  const int n_head = m.id_len + 1 + x.n_pos;
  if ((int)cover_prev < n_head) {                 // the head's bytes no match covers
    const unsigned long long rv = rev_digits32(x.pos);
    for (int k = (int)cover_prev; k < n_head; ++k) {
      uint32_t b;
      if (k < m.id_len) b = S.id[k];
      else if (k == m.id_len) b = (uint32_t)'\t';
      else b = (uint32_t)'0' + (uint32_t)((rv >> (4 * (k - m.id_len - 1))) & 15ull);
      lit(b);
    }
  }
  const uint32_t cover = tok_cover(tok);
  if (tok & kTokMatch) {
    match((uint32_t)x.n_tail + cover, tok_dist(tok));
  } else {
    tail_bytes(x, lit);
    if (cover) match(cover, (uint32_t)(n_head + x.n_tail));
  }
}

__global__ __launch_bounds__(kT) void rows_deflate_kernel(RowsParams p) {
  __shared__ Shared S;
  const int t = (int)threadIdx.x;
  for (int mi = (int)blockIdx.x; mi < p.n_members; mi += (int)gridDim.x) {
    const RowsMember m = p.members[mi];
    const int n = m.n_rows;
    __syncthreads();
    // ---- set-up --------------------------------------------------------------------------------------------------------
    for (int i = t; i < 288; i += kT) S.hist_ll[i] = 0u;
    if (t < 32) S.hist_d[t] = 0u;
    if (t < 20) S.hist_cl[t] = 0u;
    if (t < 256) {
      uint32_t c = (uint32_t)t;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? kCrcPoly : 0u);
      S.crc_tab[t] = c;
    }
    if (t == 0) {
      uint32_t v = 0x40000000u;      // x^1
      for (int k = 0; k < 32; ++k) { S.x2n[k] = v; v = gf_mul(v, v); }
      S.crc = 0u;
      S.ok = (m.id_len <= kMaxId && n > 0 && n <= kRowsMax) ? 1u : 0u;
    }
    for (int i = t; i < m.id_len && i < kMaxId; i += kT) S.id[i] = p.ids[m.id_off + i];
    __syncthreads();
    if (!S.ok) {
      if (t == 0) { RowsResult r; r.off = 0ull; r.n_bytes = 0u; r.crc = 0u; r.text_len = 0u; r.status = 1u; p.results[mi] = r; }
      continue;
    }
    // the sort works on the next power of two >= n
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    const int r0 = t * kPer;

    // ---- pass 1: row lengths -> offsets; sort keys ----------------------------------------------------------------------
    uint32_t mine = 0;
    for (int k = 0; k < kPer; ++k) {
      const int r = r0 + k;
      if (r < n) {
        const Row x = load_row(p, m, r);
        const uint32_t len = (uint32_t)(m.id_len + 1 + x.n_pos + x.n_tail);
        S.off[r] = mine;                  // relative to the thread's first row for now
        mine += len;
        S.key[r] = (tail_hash(x) << 14) | (uint32_t)r;
      } else if (r < n2) {
        S.key[r] = 0xFFFFFFFFu;
      }
    }
    uint32_t text_len = 0;
    const uint32_t base_bytes = block_scan(mine, S.scan, &text_len);
    for (int k = 0; k < kPer; ++k) {
      const int r = r0 + k;
      if (r < n) S.off[r] += base_bytes;
    }
    if (t == 0) S.off[n] = text_len;
    __syncthreads();

    // ---- bitonic sort of the keys ---------------------------------------------------------------------------------------
    for (int k2 = 2; k2 <= n2; k2 <<= 1) {
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        for (int i = t; i < (n2 >> 1); i += kT) {
          const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1));
          const int hi = lo | j;
          const uint32_t a = S.key[lo], b = S.key[hi];
          const bool up = (lo & k2) == 0;
          if ((a > b) == up) { S.key[lo] = b; S.key[hi] = a; }
        }
        __syncthreads();
      }
    }

    // ---- every row's predecessor: the nearest earlier row with its hash ---------------------------------------------------
    uint32_t pair[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = r0 + k;
      uint32_t v = 0xFFFFFFFFu;
      if (i < n) {
        const uint32_t cur = S.key[i];
        const uint32_t prev = i > 0 ? S.key[i - 1] : 0xFFFFFFFFu;
        const uint32_t j = (i > 0 && (prev >> 14) == (cur >> 14)) ? (prev & 16383u) : 0xFFFFu;
        v = (cur & 16383u) | (j << 16);
      }
      pair[k] = v;
    }
    __syncthreads();
    // tokens (the keys' array is free now)
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const uint32_t v = pair[k];
      if (v == 0xFFFFFFFFu) continue;
      const int r = (int)(v & 0xFFFFu), j = (int)(v >> 16);
      const Row x = load_row(p, m, r);
      bool has = false;
      uint32_t cover = 0, dist = 0;
      if (j != 0xFFFF) {
        const Row y = load_row(p, m, j);
        dist = (S.off[r] + (uint32_t)x.n_pos) - (S.off[j] + (uint32_t)y.n_pos);
        has = same_tail(x, y) && dist <= 32768u;
      }
      if (r + 1 < n) {
        // the next row's head against the head behind the matched tail (row j + 1), else against this row's own head
        const uint32_t other = has ? (uint32_t)(m.pos0 + j + 1) : x.pos;
        uint32_t c = (uint32_t)(m.id_len + 1 + common_digits(other, x.pos + 1u));
        const uint32_t cap = has ? 258u - (uint32_t)x.n_tail : 258u;
        c = c < cap ? c : cap;
        const uint32_t row_len = (uint32_t)(m.id_len + 1 + x.n_pos + x.n_tail);
        if (!has && (c < 3u || row_len > 32768u)) c = 0u;
        cover = c;
      }
      S.key[r] = tok_make(has, cover, dist);
    }
    __syncthreads();

    // ---- pass 2: symbol histogram ------------------------------------------------------------------------------------------
    for (int k = 0; k < kPer; ++k) {
      const int r = r0 + k;
      if (r >= n) break;
      const Row x = load_row(p, m, r);
      const uint32_t cover_prev = r > 0 ? tok_cover(S.key[r - 1]) : 0u;
      row_tokens(S, m, x, S.key[r], cover_prev,
                 [&](uint32_t b) { atomicAdd(&S.hist_ll[b], 1u); },
                 [&](uint32_t L, uint32_t D) {
                   uint32_t c, eb, ev;
                   len_code(L, &c, &eb, &ev);
                   atomicAdd(&S.hist_ll[257u + c], 1u);
                   dist_code(D, &c, &eb, &ev);
                   atomicAdd(&S.hist_d[c], 1u);
                 });
    }
    if (t == 0) atomicAdd(&S.hist_ll[256], 1u);
    __syncthreads();
    if (t == 0) {      // two distance codes at the least, as zlib's deflate keeps it
      int used = 0;
      for (int s = 0; s < 30; ++s) used += S.hist_d[s] != 0u;
      for (int s = 0; s < 2 && used < 2; ++s)
        if (!S.hist_d[s]) { S.hist_d[s] = 1u; ++used; }
    }
    __syncthreads();

    // ---- the three codes and the block header --------------------------------------------------------------------------------
    build_lengths(S, S.hist_ll, 286, 15, S.len_ll, 0);
    build_lengths(S, S.hist_d, 30, 15, S.len_d, 0);
    if (t == 0) {
      int n_ll = 286, n_d = 30;
      while (n_ll > 257 && S.len_ll[n_ll - 1] == 0) --n_ll;
      while (n_d > 1 && S.len_d[n_d - 1] == 0) --n_d;
      S.n_ll = (uint32_t)n_ll; S.n_d = (uint32_t)n_d;
      for (int s = 0; s < n_ll; ++s) S.hist_cl[S.len_ll[s]] += 1u;
      for (int s = 0; s < n_d; ++s) S.hist_cl[S.len_d[s]] += 1u;
      int used = 0;
      for (int s = 0; s < 19; ++s) used += S.hist_cl[s] != 0u;
      for (int s = 0; s < 2 && used < 2; ++s)
        if (!S.hist_cl[s]) { S.hist_cl[s] = 1u; ++used; }
    }
    __syncthreads();
    build_lengths(S, S.hist_cl, 19, 7, S.len_cl, 0);
    make_codes(S, S.len_ll, (int)S.n_ll, S.code_ll);
    make_codes(S, S.len_d, (int)S.n_d, S.code_d);
    make_codes(S, S.len_cl, 19, S.code_cl);
    if (t < 96) S.hdr[t] = 0u;
    __syncthreads();
    if (t == 0) {
      const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      int n_cl = 19;
      while (n_cl > 4 && S.len_cl[order[n_cl - 1]] == 0) --n_cl;
      unsigned long long acc = 0;
      int fill = 0, word = 0;
      auto put = [&](uint32_t bits, int nb) {
        acc |= (unsigned long long)bits << fill;
        fill += nb;
        if (fill >= 32) { S.hdr[word++] = (uint32_t)acc; acc >>= 32; fill -= 32; }
      };
      put(1u, 1);                 // BFINAL
      put(2u, 2);                 // BTYPE = dynamic Huffman
      put(S.n_ll - 257u, 5);
      put(S.n_d - 1u, 5);
      put((uint32_t)(n_cl - 4), 4);
      for (int i = 0; i < n_cl; ++i) put(S.len_cl[order[i]], 3);
      for (int s = 0; s < (int)S.n_ll; ++s) { const uint32_t c = S.code_cl[S.len_ll[s]]; put(c & 0xFFFFu, (int)(c >> 16)); }
      for (int s = 0; s < (int)S.n_d; ++s) { const uint32_t c = S.code_cl[S.len_d[s]]; put(c & 0xFFFFu, (int)(c >> 16)); }
      S.hdr_bits = (uint32_t)(word * 32 + fill);
      if (fill > 0) S.hdr[word] = (uint32_t)acc;
    }
    __syncthreads();

    // ---- pass 3: bit lengths -> where every thread's bits go ----------------------------------------------------------------
    uint32_t bits = 0;
    for (int k = 0; k < kPer; ++k) {
      const int r = r0 + k;
      if (r >= n) break;
      const Row x = load_row(p, m, r);
      const uint32_t cover_prev = r > 0 ? tok_cover(S.key[r - 1]) : 0u;
      row_tokens(S, m, x, S.key[r], cover_prev,
                 [&](uint32_t b) { bits += S.code_ll[b] >> 16; },
                 [&](uint32_t L, uint32_t D) {
                   uint32_t c, eb, ev;
                   len_code(L, &c, &eb, &ev);
                   bits += (S.code_ll[257u + c] >> 16) + eb;
                   dist_code(D, &c, &eb, &ev);
                   bits += (S.code_d[c] >> 16) + eb;
                 });
    }
    const int last_thread = (n - 1) / kPer;
    if (t == 0) bits += S.hdr_bits;
    if (t == last_thread) bits += S.code_ll[256] >> 16;
    uint32_t total_bits = 0;
    const uint32_t bit0 = block_scan(bits, S.scan, &total_bits);
    const uint32_t n_bytes = (total_bits + 7u) >> 3;
    if (t == 0) {
      const unsigned long long need = ((unsigned long long)n_bytes + 3ull) & ~3ull;
      const unsigned long long at = atomicAdd(p.cursor, need);
      S.base = at;
      S.ok = (at + need <= p.arena_bytes) ? 1u : 0u;
    }
    __syncthreads();
    if (!S.ok) {
      if (t == 0) { RowsResult r; r.off = 0ull; r.n_bytes = 0u; r.crc = 0u; r.text_len = text_len; r.status = 2u; p.results[mi] = r; }
      continue;
    }

    // ---- pass 4: the bits, and the CRC of the text ------------------------------------------------------------------------------
    if (t * kPer < n) {
      BitOut out;
      out.init(reinterpret_cast<uint32_t*>(p.arena + S.base), bit0);
      if (t == 0) {
        const int full = (int)(S.hdr_bits >> 5), rest = (int)(S.hdr_bits & 31u);
        for (int w = 0; w < full; ++w) { out.put(S.hdr[w] & 0xFFFFu, 16); out.put(S.hdr[w] >> 16, 16); }
        if (rest > 16) { out.put(S.hdr[full] & 0xFFFFu, 16); out.put((S.hdr[full] >> 16) & ((1u << (rest - 16)) - 1u), rest - 16); }
        else if (rest > 0) out.put(S.hdr[full] & ((1u << rest) - 1u), rest);
      }
      uint32_t crc = 0u;
      uint32_t my_bytes = 0u;
      for (int k = 0; k < kPer; ++k) {
        const int r = r0 + k;
        if (r >= n) break;
        const Row x = load_row(p, m, r);
        const uint32_t cover_prev = r > 0 ? tok_cover(S.key[r - 1]) : 0u;
        row_tokens(S, m, x, S.key[r], cover_prev,
                   [&](uint32_t b) { const uint32_t c = S.code_ll[b]; out.put(c & 0xFFFFu, (int)(c >> 16)); },
                   [&](uint32_t L, uint32_t D) {
                     uint32_t c, eb, ev;
                     len_code(L, &c, &eb, &ev);
                     uint32_t cw = S.code_ll[257u + c];
                     out.put(cw & 0xFFFFu, (int)(cw >> 16));
                     if (eb) out.put(ev, (int)eb);
                     dist_code(D, &c, &eb, &ev);
                     cw = S.code_d[c];
                     out.put(cw & 0xFFFFu, (int)(cw >> 16));
                     if (eb) out.put(ev, (int)eb);
                   });
        // the row's text through the CRC
        auto feed = [&](uint32_t b) { crc = S.crc_tab[(crc ^ b) & 255u] ^ (crc >> 8); };
        for (int i = 0; i < m.id_len; ++i) feed(S.id[i]);
        feed((uint32_t)'\t');
        unsigned long long rv = rev_digits32(x.pos);
        for (int i = 0; i < x.n_pos; ++i) { feed((uint32_t)'0' + (uint32_t)(rv & 15ull)); rv >>= 4; }
        tail_bytes(x, feed);
        my_bytes += (uint32_t)(m.id_len + 1 + x.n_pos + x.n_tail);
      }
      if (t == last_thread) { const uint32_t c = S.code_ll[256]; out.put(c & 0xFFFFu, (int)(c >> 16)); }
      out.finish();
      // shift this thread's CRC register to the end of the text
      const uint32_t behind = text_len - (S.off[r0] + my_bytes);
      uint32_t part = gf_mul(crc, gf_xpow8((unsigned long long)behind, S.x2n));
      if (t == 0) part ^= gf_mul(0xFFFFFFFFu, gf_xpow8((unsigned long long)text_len, S.x2n));    // the initial register
      atomicXor(&S.crc, part);
    }
    __syncthreads();
    if (t == 0) {
      RowsResult r;
      r.off = S.base; r.n_bytes = n_bytes; r.crc = S.crc ^ 0xFFFFFFFFu; r.text_len = text_len; r.status = 0u;
      p.results[mi] = r;
    }
  }
}

}  // namespace

hipError_t launch_rows_deflate(const RowsParams& p, int grid_blocks, hipStream_t s) {
  if (p.n_members <= 0) return hipSuccess;
  const int g = p.n_members < grid_blocks ? p.n_members : grid_blocks;
  hipLaunchKernelGGL(rows_deflate_kernel, dim3((unsigned)g), dim3(kT), 0, s, p);
  return hipGetLastError();
}

}  // namespace midas
