// See row_deflate.h.  RFC 1951 section numbers in the comments.
#include "row_deflate.h"

#include <algorithm>
#include <cstring>

namespace midas {
namespace {

constexpr int kMaxMatch = 258, kMinMatch = 3, kWindow = 32768;
constexpr int kTableBits = 13;

// 3.2.5: length 3..258 -> code 257..285 + extra bits; distance 1..32768 -> code 0..29 + extra bits
constexpr uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
// 3.2.7: the order in which the code lengths of the code length alphabet are sent
constexpr uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Tables {
  uint8_t len_code[kMaxMatch + 1];     // length -> index into kLenBase
  uint8_t dist_code_lo[256];           // distance - 1 < 256
  uint8_t dist_code_hi[256];           // (distance - 1) >> 7 for the rest
  Tables() {
    for (int c = 0; c < 29; ++c) {
      const int hi = c == 28 ? 258 : kLenBase[c] + (1 << kLenExtra[c]) - 1;
      for (int l = kLenBase[c]; l <= hi && l <= kMaxMatch; ++l) len_code[l] = (uint8_t)c;
    }
    len_code[258] = 28;
    for (int c = 0; c < 30; ++c) {
      const int lo = kDistBase[c], hi = lo + (1 << kDistExtra[c]) - 1;
      for (int d = lo; d <= hi; ++d) {
        if (d - 1 < 256) dist_code_lo[d - 1] = (uint8_t)c;
        else dist_code_hi[(d - 1) >> 7] = (uint8_t)c;      // codes 16.. span multiples of 128
      }
    }
  }
  int dist_code(size_t d) const { return d <= 256 ? dist_code_lo[d - 1] : dist_code_hi[(d - 1) >> 7]; }
};
const Tables kT;

inline uint16_t reverse_bits(uint16_t v, int n) {
  uint16_t r = 0;
  for (int i = 0; i < n; ++i) { r = (uint16_t)((r << 1) | (v & 1)); v >>= 1; }
  return r;
}

inline uint64_t load_upto8(const uint8_t* p, size_t n) {
  uint64_t v = 0;
  memcpy(&v, p, n < 8 ? n : 8);
  return v;
}

inline uint32_t hash_tail(const uint8_t* p, size_t n) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
  for (size_t i = 0; i < n; i += 8) h = (h ^ load_upto8(p + i, n - i)) * 0xFF51AFD7ED558CCDull;
  return (uint32_t)(h >> (64 - kTableBits));
}

}  // namespace

RowDeflate::RowDeflate() : table_((size_t)1 << kTableBits) {}

void RowDeflate::match(size_t len, size_t dist) {
  // a match longer than 258 goes out in pieces, none of them shorter than 3
  while (len > 0) {
    size_t take = len > (size_t)kMaxMatch ? (size_t)kMaxMatch : len;
    if (len - take > 0 && len - take < (size_t)kMinMatch) take = len - kMinMatch;
    tok_ll_.push_back((uint16_t)take);
    tok_d_.push_back((uint16_t)dist);              // 1..32768
    ++freq_ll_[257 + kT.len_code[take]];
    ++freq_d_[kT.dist_code(dist)];
    len -= take;
  }
}

// Code lengths of a prefix code for the symbols with freq > 0, none longer than max_len, Kraft sum exactly one (zlib's
// inflate refuses an incomplete code).  Huffman's lengths by the two-queue construction; if the tree is deeper than
// max_len the lengths are clamped and the Kraft sum repaired: deepen the deepest shorter codes until it fits, then give
// what is left back to the longest ones.
void RowDeflate::build_lengths(const uint32_t* freq, int n, int max_len, uint8_t* len_out) {
  struct Node { uint64_t w; int parent; };
  int order[288];
  int m = 0;
  for (int s = 0; s < n; ++s) { len_out[s] = 0; if (freq[s]) order[m++] = s; }
  if (m == 0) return;
  if (m == 1) { len_out[order[0]] = 1; return; }
  std::sort(order, order + m, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
  Node nodes[2 * 288];
  for (int i = 0; i < m; ++i) nodes[i] = {freq[order[i]], -1};
  int leaf = 0, inner = m, made = m;
  auto take = [&]() {   // the lighter of the next unused leaf and the next unused internal node
    if (leaf < m && (inner >= made || nodes[leaf].w <= nodes[inner].w)) return leaf++;
    return inner++;
  };
  while ((m - leaf) + (made - inner) > 1) {
    const int a = take(), b = take();
    nodes[made] = {nodes[a].w + nodes[b].w, -1};
    nodes[a].parent = nodes[b].parent = made;
    ++made;
  }
  int depth[2 * 288];
  depth[made - 1] = 0;
  for (int i = made - 2; i >= 0; --i) depth[i] = depth[nodes[i].parent] + 1;
  // clamp, then repair (units of 2^-max_len)
  long long kraft = 0;
  const long long one = 1ll << max_len;
  int len[288];
  for (int i = 0; i < m; ++i) {
    len[i] = depth[i] > max_len ? max_len : depth[i];
    kraft += one >> len[i];
  }
  // leaves are sorted by ascending frequency: lengths are non-increasing along i up to ties
  while (kraft > one) {
    int pick = -1;
    for (int i = 0; i < m; ++i)
      if (len[i] < max_len && (pick < 0 || len[i] > len[pick])) pick = i;      // deepest code that can still grow; first = rarest
    kraft -= one >> (len[pick] + 1);
    ++len[pick];
  }
  while (kraft < one) {       // slack: shorten the longest code that may be shortened, the most frequent symbol of that length
    int pick = -1;
    for (int i = m - 1; i >= 0; --i)
      if (len[i] > 1 && kraft + (one >> len[i]) <= one && (pick < 0 || len[i] > len[pick])) pick = i;
    if (pick < 0) break;
    kraft += one >> len[pick];
    --len[pick];
  }
  for (int i = 0; i < m; ++i) len_out[order[i]] = (uint8_t)len[i];
}

// 3.2.2: canonical codes from the lengths; stored bit-reversed because DEFLATE packs Huffman codes MSB first into an
// LSB-first bit stream
void RowDeflate::make_codes(const uint8_t* len, int n, Code* codes) {
  int bl_count[16] = {0};
  for (int s = 0; s < n; ++s) ++bl_count[len[s]];
  bl_count[0] = 0;
  uint16_t next_code[16];
  uint16_t code = 0;
  for (int b = 1; b <= 15; ++b) {
    code = (uint16_t)((code + bl_count[b - 1]) << 1);
    next_code[b] = code;
  }
  for (int s = 0; s < n; ++s) {
    codes[s].len = len[s];
    codes[s].bits = len[s] ? reverse_bits(next_code[len[s]]++, len[s]) : 0;
  }
}

void RowDeflate::compress(const uint8_t* text, size_t n, const uint32_t* row_begin, const uint32_t* tail_begin, size_t n_rows,
                          std::vector<uint8_t>& out) {
  tok_ll_.clear();
  tok_d_.clear();
  tok_ll_.reserve(n_rows * 5 + 16);
  tok_d_.reserve(n_rows * 5 + 16);
  memset(freq_ll_, 0, sizeof freq_ll_);
  memset(freq_d_, 0, sizeof freq_d_);
  std::fill(table_.begin(), table_.end(), 0u);

  // ---- tokens --------------------------------------------------------------------------------------------------------
  size_t pos = 0;                                   // first byte no token covers yet
  for (size_t k = 0; k < n_rows; ++k) {
    const size_t rb = row_begin[k], tb = tail_begin[k], re = k + 1 < n_rows ? row_begin[k + 1] : n;
    if (pos < tb) {
      if (pos == rb && k > 0) {                     // the head of the row against the head of the row before it
        const size_t prev = row_begin[k - 1], dist = rb - prev;
        size_t l = 0;
        while (rb + l < tb && text[prev + l] == text[rb + l]) ++l;
        if (l >= (size_t)kMinMatch + 1 && dist <= (size_t)kWindow) { match(l, dist); pos += l; }
      }
      for (; pos < tb; ++pos) literal(text[pos]);
    }
    const size_t tl = re - tb;
    const uint32_t h = hash_tail(text + tb, tl);
    const uint32_t cand = table_[h];
    table_[h] = (uint32_t)tb + 1u;
    if (pos > tb) continue;                          // (a match ran across this tail: nothing left to decide here)
    bool matched = false;
    if (cand) {
      const size_t c = cand - 1, dist = tb - c;
      if (dist <= (size_t)kWindow && c + tl <= tb && memcmp(text + c, text + tb, tl) == 0) {
        size_t l = tl;                               // through the newline into the next row's head, as far as it agrees
        const size_t cap = std::min<size_t>(kMaxMatch, n - tb);
        while (l < cap && text[c + l] == text[tb + l]) ++l;
        if (l >= (size_t)kMinMatch) {
          match(l, dist);
          pos = tb + l;
          matched = true;
        }
      }
    }
    if (!matched)
      for (; pos < re; ++pos) literal(text[pos]);
  }
  for (; pos < n; ++pos) literal(text[pos]);         // (text behind the last row, if a caller has any)
  ++freq_ll_[256];

  // ---- codes ---------------------------------------------------------------------------------------------------------
  // zlib's deflate keeps two codes in the distance tree at the least; so does this (a lone or missing distance code is a
  // corner of the format not every inflater agrees on)
  int used_d = 0;
  for (int s = 0; s < 30; ++s) used_d += freq_d_[s] != 0;
  for (int s = 0; s < 2 && used_d < 2; ++s)
    if (!freq_d_[s]) { freq_d_[s] = 1; ++used_d; }
  uint8_t len_ll[288], len_d[32];
  build_lengths(freq_ll_, 286, 15, len_ll);
  build_lengths(freq_d_, 30, 15, len_d);
  int n_ll = 286, n_d = 30;
  while (n_ll > 257 && len_ll[n_ll - 1] == 0) --n_ll;
  while (n_d > 1 && len_d[n_d - 1] == 0) --n_d;
  Code code_ll[288], code_d[32];
  make_codes(len_ll, n_ll, code_ll);
  make_codes(len_d, n_d, code_d);
  // the code lengths themselves, Huffman coded (3.2.7) -- sent one by one: the run-length symbols 16-18 would save some
  // of ~150 bytes per member
  uint32_t freq_cl[19] = {0};
  for (int s = 0; s < n_ll; ++s) ++freq_cl[len_ll[s]];
  for (int s = 0; s < n_d; ++s) ++freq_cl[len_d[s]];
  int used_cl = 0;
  for (int s = 0; s < 19; ++s) used_cl += freq_cl[s] != 0;
  for (int s = 0; s < 2 && used_cl < 2; ++s)
    if (!freq_cl[s]) { freq_cl[s] = 1; ++used_cl; }
  uint8_t len_cl[19];
  build_lengths(freq_cl, 19, 7, len_cl);
  Code code_cl[19];
  make_codes(len_cl, 19, code_cl);
  int n_cl = 19;
  while (n_cl > 4 && len_cl[kClOrder[n_cl - 1]] == 0) --n_cl;

  // ---- bits ----------------------------------------------------------------------------------------------------------
  // room: a Huffman code is never worse than eight bits a symbol on average, matches only shorten it, the length limit
  // costs a fraction of a bit; the header is a few hundred bytes
  const size_t before = out.size();
  out.resize(before + n + n / 8 + 4096);
  at_ = out.data() + before;
  acc_ = 0;
  fill_ = 0;
  put(1, 1);                 // BFINAL
  put(2, 2);                 // BTYPE = dynamic Huffman
  put((uint32_t)(n_ll - 257), 5);
  put((uint32_t)(n_d - 1), 5);
  put((uint32_t)(n_cl - 4), 4);
  for (int i = 0; i < n_cl; ++i) put(len_cl[kClOrder[i]], 3);
  for (int s = 0; s < n_ll; ++s) put(code_cl[len_ll[s]].bits, code_cl[len_ll[s]].len);
  for (int s = 0; s < n_d; ++s) put(code_cl[len_d[s]].bits, code_cl[len_d[s]].len);
  const size_t n_tok = tok_ll_.size();
  for (size_t t = 0; t < n_tok; ++t) {
    const uint32_t d = tok_d_[t];
    if (d == 0) {
      const Code& c = code_ll[tok_ll_[t]];
      put(c.bits, c.len);
    } else {
      const size_t len = tok_ll_[t], dist = d;
      const int lc = kT.len_code[len];
      const Code& c = code_ll[257 + lc];
      put(c.bits, c.len);
      if (kLenExtra[lc]) put((uint32_t)(len - kLenBase[lc]), kLenExtra[lc]);
      const int dc = kT.dist_code(dist);
      const Code& e = code_d[dc];
      put(e.bits, e.len);
      if (kDistExtra[dc]) put((uint32_t)(dist - kDistBase[dc]), kDistExtra[dc]);
    }
  }
  put(code_ll[256].bits, code_ll[256].len);
  while (fill_ > 0) { *at_++ = (uint8_t)acc_; acc_ >>= 8; fill_ -= fill_ < 8 ? fill_ : 8; }
  out.resize((size_t)(at_ - out.data()));
  at_ = nullptr;
}

}  // namespace midas
