// Test program (tests/test_row_deflate.py compiles and runs it): RowDeflate::build_lengths under frequency tables that make
// Huffman's tree deeper than DEFLATE allows -- every code it returns must be complete (Kraft sum exactly one: zlib's
// inflate refuses anything else), within the limit, and give every used symbol a code.
#define private public
#include "../../midas_amd/csrc/row_deflate.h"
#undef private

#include <cstdio>
#include <cstdlib>
#include <random>

static long long g_cases = 0, g_limited = 0;

static bool check(midas::RowDeflate& rd, const uint32_t* freq, int n, int max_len, const char* what) {
  uint8_t len[288];
  rd.build_lengths(freq, n, max_len, len);
  int used = 0;
  long long kraft = 0;
  const long long one = 1ll << max_len;
  bool at_limit = false;
  for (int s = 0; s < n; ++s) {
    if (!freq[s]) { if (len[s]) { printf("%s: unused symbol %d got a code\n", what, s); return false; } continue; }
    ++used;
    if (len[s] < 1 || len[s] > max_len) { printf("%s: symbol %d has length %d (limit %d)\n", what, s, len[s], max_len); return false; }
    at_limit = at_limit || len[s] == max_len;
    kraft += one >> len[s];
  }
  ++g_cases;
  g_limited += at_limit;
  if (used >= 2 && kraft != one) { printf("%s: Kraft sum %lld / %lld with %d symbols\n", what, kraft, one, used); return false; }
  if (used == 1 && kraft != one / 2) { printf("%s: a lone symbol must get one bit\n", what); return false; }
  return true;
}

int main() {
  midas::RowDeflate rd;
  std::mt19937_64 rng(12345);
  const int shapes[3][2] = {{286, 15}, {30, 15}, {19, 7}};
  for (const auto& sh : shapes) {
    const int n = sh[0], max_len = sh[1];
    uint32_t freq[288];
    for (int trial = 0; trial < 4000; ++trial) {
      for (int s = 0; s < n; ++s) freq[s] = 0;
      const int used = 1 + (int)(rng() % (unsigned)n);
      const int kind = trial % 5;
      uint64_t a = 1, b = 1;
      for (int k = 0; k < used; ++k) {
        int s;
        do { s = (int)(rng() % (unsigned)n); } while (freq[s]);
        uint64_t f;
        if (kind == 0) f = 1 + rng() % 1000;                                   // flat
        else if (kind == 1) { f = a; const uint64_t c = a + b; a = b; b = c; if (b > 400000000ull) { a = b = 1; } }   // Fibonacci: the deepest tree there is
        else if (kind == 2) f = 1ull << (k % 31);                              // powers of two
        else if (kind == 3) f = (rng() % 4 == 0) ? 1 + rng() % 3 : 1000000 + rng() % 1000;   // a few rare among heavy ones
        else f = 1 + (uint64_t)((double)(rng() % 1000000) * (double)(rng() % 1000) / 1000.0);
        freq[s] = (uint32_t)(f > 0xFFFFFFFFull ? 0xFFFFFFFFull : f);
      }
      if (!check(rd, freq, n, max_len, n == 286 ? "literal/length" : (n == 30 ? "distance" : "code length"))) return 1;
    }
  }
  if (g_limited < g_cases / 20) { printf("only %lld of %lld cases reached the length limit: the test lost its teeth\n", g_limited, g_cases); return 1; }
  printf("ok %lld cases, %lld at the length limit\n", g_cases, g_limited);
  return 0;
}
