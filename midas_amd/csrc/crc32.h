// CRC-32 (gzip / BGZF: polynomial 0xEDB88320, reflected) on the device: the table-driven register update and the GF(2)
// arithmetic that moves a register over the bytes behind it, so that the lanes of a wavefront can each take a piece of a
// buffer from a zero register and the pieces be combined: crc(A || B) = shift(crc(A), |B|) ^ crc0(B).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace midas {
namespace crc {

constexpr uint32_t kPoly = 0xEDB88320u;

// a * b in GF(2)[x] / P (reflected: bit 31 is x^0)
__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int i = 0; i < 32; ++i) {
    p ^= (a & 0x80000000u) ? b : 0u;
    a <<= 1;
    b = (b >> 1) ^ ((b & 1u) ? kPoly : 0u);
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

// Tables of one workgroup (>= 256 threads, all of them call this): tab[k][b] = the register after byte b and k zero bytes
// (slicing by four), x2n[k] = x^(2^k) mod P.  Ends with a barrier.
struct Tables { uint32_t tab[4][256]; uint32_t x2n[32]; };
__device__ __forceinline__ void build_tables(Tables& T) {
  const int t = threadIdx.x;
  if (t < 256) {
    uint32_t c = (uint32_t)t;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? kPoly : 0u);
    T.tab[0][t] = c;
  }
  if (t == 0) {
    uint32_t p = 0x40000000u;      // x^1
    T.x2n[0] = p;
    for (int k = 1; k < 32; ++k) { p = gf_mul(p, p); T.x2n[k] = p; }
  }
  __syncthreads();
  for (int k = 1; k < 4; ++k) {
    if (t < 256) { const uint32_t c = T.tab[k - 1][t]; T.tab[k][t] = (c >> 8) ^ T.tab[0][c & 255u]; }
    __syncthreads();
  }
}
// the register (started at `c`) after the n bytes at p
__device__ __forceinline__ uint32_t update(const Tables& T, uint32_t c, const uint8_t* p, uint32_t n) {
  typedef uint32_t u32_a1 __attribute__((aligned(1)));
  uint32_t i = 0;
  for (; i + 4 <= n; i += 4) {
    c ^= *reinterpret_cast<const u32_a1*>(p + i);
    c = T.tab[3][c & 255u] ^ T.tab[2][(c >> 8) & 255u] ^ T.tab[1][(c >> 16) & 255u] ^ T.tab[0][c >> 24];
  }
  for (; i < n; ++i) c = T.tab[0][(c ^ p[i]) & 255u] ^ (c >> 8);
  return c;
}

}  // namespace crc
}  // namespace midas
