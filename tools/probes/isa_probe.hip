// Developer probe (not part of the library): checks on a real gfx950 the instruction behaviour pileup_direct.hip relies on
//   1. v_perm_b32 selector semantics for selectors >= 8 (sign replication, 0x00, 0xFF)
//   2. v_cmp_gt_u32_sdwa with independent byte selects on two VGPR operands
//   3. issue cost of plain / SDWA / perm VALU instructions per wave64 (cycles per instruction on one SIMD)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/isa_probe.hip -o /tmp/isa_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void perm_kernel(uint32_t s0, uint32_t s1, uint32_t* out) {
  const uint32_t sel = threadIdx.x * 0x01010101u;   // selector byte = lane id (0..63) in every byte
  out[threadIdx.x] = __builtin_amdgcn_perm(s0, s1, sel);
}

__global__ void sdwa_kernel(const uint32_t* a, const uint32_t* b, unsigned long long* out) {
  const uint32_t x = a[threadIdx.x], y = b[threadIdx.x];
  unsigned long long m0, m1, m2, m3;
  asm volatile("v_cmp_gt_u32_sdwa %0, %4, %5 src0_sel:BYTE_0 src1_sel:BYTE_3\n\t"
               "v_cmp_gt_u32_sdwa %1, %4, %5 src0_sel:BYTE_1 src1_sel:BYTE_2\n\t"
               "v_cmp_gt_u32_sdwa %2, %4, %5 src0_sel:BYTE_2 src1_sel:BYTE_1\n\t"
               "v_cmp_gt_u32_sdwa %3, %4, %5 src0_sel:BYTE_3 src1_sel:BYTE_0\n\t"
               : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(x), "v"(y));
  if (threadIdx.x == 0) { out[0] = m0; out[1] = m1; out[2] = m2; out[3] = m3; }
}

template <int KIND>
__global__ void issue_kernel(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a ^ 0x55u, d = a + 7u;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (KIND == 0) {        // plain VALU, four independent chains
        asm volatile("v_and_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_or_b32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
      } else if (KIND == 1) { // v_perm_b32
        asm volatile("v_perm_b32 %0, %0, %4, %1\n\tv_perm_b32 %1, %1, %4, %2\n\tv_perm_b32 %2, %2, %4, %3\n\tv_perm_b32 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
      } else if (KIND == 2) { // SDWA or
        asm volatile("v_or_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
                     "v_or_b32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
                     "v_or_b32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
                     "v_or_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
      } else if (KIND == 3) { // SDWA compare into SGPR pairs
        unsigned long long m0, m1, m2, m3;
        asm volatile("v_cmp_gt_u32_sdwa %0, %4, %5 src0_sel:BYTE_0 src1_sel:BYTE_3\n\t"
                     "v_cmp_gt_u32_sdwa %1, %4, %5 src0_sel:BYTE_1 src1_sel:BYTE_2\n\t"
                     "v_cmp_gt_u32_sdwa %2, %4, %5 src0_sel:BYTE_2 src1_sel:BYTE_1\n\t"
                     "v_cmp_gt_u32_sdwa %3, %4, %5 src0_sel:BYTE_3 src1_sel:BYTE_0"
                     : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(a), "v"(b));
        a += (uint32_t)(m0 ^ m1 ^ m2 ^ m3);
      } else if (KIND == 4) { // v_sad_u8
        asm volatile("v_sad_u8 %0, %1, %4, %0\n\tv_sad_u8 %1, %2, %4, %1\n\tv_sad_u8 %2, %3, %4, %2\n\tv_sad_u8 %3, %0, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[1 << 20] = (uint32_t)(t1 - t0); }
}

template <int KIND>
static void time_issue(const char* name, int waves_per_simd) {
  uint32_t* d; hipMalloc(&d, ((1 << 20) + 4) * 4);
  const int iters = 2000;
  const int threads = 256 * waves_per_simd;   // one CU: 4 SIMDs x waves_per_simd
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  issue_kernel<KIND><<<256, threads>>>(d, 10, 1);   // warm
  hipEventRecord(e0);
  issue_kernel<KIND><<<256, threads>>>(d, iters, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * 64;    // 16 x 4 per iteration (+1 for KIND 3, ignored)
  // every SIMD hosts waves_per_simd waves; time per wave-instruction on its SIMD = ms / (instr_per_wave * waves_per_simd)
  const double ns = ms * 1e6 / (instr_per_wave * waves_per_simd);
  printf("%-10s waves/SIMD %d: %.3f ms -> %.3f ns per wave-instruction per SIMD = %.2f cycles @2.4GHz\n", name, waves_per_simd, ms, ns, ns * 2.4);
  hipFree(d);
}

int main() {
  uint32_t* d; hipMalloc(&d, 64 * 4);
  const uint32_t s0 = 0x80112233u, s1 = 0x44F05566u;   // in[0..3] = s1 bytes 66 55 F0 44 ; in[4..7] = s0 bytes 33 22 11 80
  perm_kernel<<<1, 64>>>(s0, s1, d);
  std::vector<uint32_t> h(64); hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost);
  printf("v_perm_b32 S0=%08x S1=%08x: byte result per selector 0..15:", s0, s1);
  for (int i = 0; i < 16; ++i) printf(" %02x", h[i] & 0xFF);
  printf("\n  expect: 66 55 f0 44 33 22 11 80 | sign(in1)=00 sign(in3)=00 sign(in5)=00 sign(in7)=ff | 00 | ff ff ff  (in1=55,in3=44,in5=22,in7=80)\n");
  printf("  selectors 16,32,64,128,255:"); 
  for (int i : {16, 32, 63}) printf(" %02x", h[i] & 0xFF);
  printf("\n");
  uint32_t ha[64], hb[64];
  for (int i = 0; i < 64; ++i) { ha[i] = 0x04030201u * (uint32_t)(i + 1); hb[i] = 0x10203040u + (uint32_t)i * 0x01010101u; }
  uint32_t *da, *db; unsigned long long* dm; hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dm, 32);
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  sdwa_kernel<<<1, 64>>>(da, db, dm);
  unsigned long long m[4]; hipMemcpy(m, dm, 32, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int k = 0; k < 4; ++k) {
    unsigned long long want = 0;
    for (int i = 0; i < 64; ++i) {
      const uint32_t x = (ha[i] >> (8 * k)) & 0xFF, y = (hb[i] >> (8 * (3 - k))) & 0xFF;
      if (x > y) want |= 1ull << i;
    }
    if (want != m[k]) { ++bad; printf("sdwa cmp %d: got %016llx want %016llx\n", k, m[k], want); }
  }
  printf("v_cmp_gt_u32_sdwa byte/byte: %s\n", bad ? "MISMATCH" : "ok");
  for (int w : {1, 2, 4}) {
    time_issue<0>("valu", w); time_issue<1>("perm", w); time_issue<2>("sdwa_or", w); time_issue<3>("sdwa_cmp", w); time_issue<4>("sad_u8", w);
  }
  return 0;
}
