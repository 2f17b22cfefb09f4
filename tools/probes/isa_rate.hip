// Developer probe (not part of the library): wall time per wave64 instruction on one SIMD (4 waves per SIMD, every CU busy) for the
// instruction kinds the pileup kernels are made of.  The chip clocks to its power budget, so the figures are nanoseconds, not cycles.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/isa_rate.hip -o /tmp/isa_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define BODY4(A, B, C, D) asm volatile(A "\n\t" B "\n\t" C "\n\t" D : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed), "s"(sc))

template <int KIND>
__global__ void rate_kernel(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = a * 3u, c = a ^ 0x55u, d = a + 7u;
  const uint32_t sc = seed + 3u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (KIND == 0) BODY4("v_and_b32 %0, %0, %4", "v_xor_b32 %1, %1, %4", "v_or_b32 %2, %2, %4", "v_add_u32 %3, %3, %4");
      if (KIND == 1) BODY4("v_bfe_u32 %0, %0, 3, 8", "v_bfe_u32 %1, %1, 5, 9", "v_bfe_u32 %2, %2, 1, 7", "v_bfe_u32 %3, %3, 2, 11");
      if (KIND == 2) BODY4("v_and_or_b32 %0, %0, %4, %1", "v_and_or_b32 %1, %1, %4, %2", "v_and_or_b32 %2, %2, %4, %3", "v_and_or_b32 %3, %3, %4, %0");
      if (KIND == 3) BODY4("v_lshl_or_b32 %0, %0, 3, %1", "v_lshl_or_b32 %1, %1, 2, %2", "v_lshl_or_b32 %2, %2, 1, %3", "v_lshl_or_b32 %3, %3, 4, %0");
      if (KIND == 4) BODY4("v_add3_u32 %0, %0, %4, %1", "v_add3_u32 %1, %1, %4, %2", "v_add3_u32 %2, %2, %4, %3", "v_add3_u32 %3, %3, %4, %0");
      if (KIND == 5) BODY4("v_lshl_add_u32 %0, %0, 2, %1", "v_lshl_add_u32 %1, %1, 3, %2", "v_lshl_add_u32 %2, %2, 1, %3", "v_lshl_add_u32 %3, %3, 2, %0");
      if (KIND == 6) BODY4("v_cndmask_b32 %0, %0, %4, vcc", "v_cndmask_b32 %1, %1, %4, vcc", "v_cndmask_b32 %2, %2, %4, vcc", "v_cndmask_b32 %3, %3, %4, vcc");
      if (KIND == 7) asm volatile("v_cmp_lt_u32 vcc, %0, %4\n\tv_cmp_lt_u32 vcc, %1, %4\n\tv_cmp_lt_u32 vcc, %2, %4\n\tv_cmp_lt_u32 vcc, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed) : "vcc");
      if (KIND == 8) { unsigned long long m0, m1, m2, m3;
        asm volatile("v_cmp_lt_u32 %0, %4, %8\n\tv_cmp_lt_u32 %1, %5, %8\n\tv_cmp_lt_u32 %2, %6, %8\n\tv_cmp_lt_u32 %3, %7, %8" : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed)); }
      if (KIND == 9) { unsigned long long m0, m1, m2, m3;
        asm volatile("v_cmp_gt_u32_sdwa %0, %4, %8 src0_sel:BYTE_0 src1_sel:BYTE_3\n\tv_cmp_gt_u32_sdwa %1, %5, %8 src0_sel:BYTE_1 src1_sel:BYTE_2\n\t"
                     "v_cmp_gt_u32_sdwa %2, %6, %8 src0_sel:BYTE_2 src1_sel:BYTE_1\n\tv_cmp_gt_u32_sdwa %3, %7, %8 src0_sel:BYTE_3 src1_sel:BYTE_0"
                     : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed)); }
      if (KIND == 10) BODY4("v_mad_u32_u24 %0, %0, %4, %1", "v_mad_u32_u24 %1, %1, %4, %2", "v_mad_u32_u24 %2, %2, %4, %3", "v_mad_u32_u24 %3, %3, %4, %0");
      if (KIND == 11) BODY4("v_mul_lo_u32 %0, %0, %4", "v_mul_lo_u32 %1, %1, %4", "v_mul_lo_u32 %2, %2, %4", "v_mul_lo_u32 %3, %3, %4");
      if (KIND == 12) BODY4("v_med3_i32 %0, %0, %4, %1", "v_med3_i32 %1, %1, %4, %2", "v_min_i32 %2, %2, %4", "v_max_i32 %3, %3, %4");
      if (KIND == 13) BODY4("v_alignbit_b32 %0, %0, %1, %4", "v_alignbyte_b32 %1, %1, %2, %4", "v_alignbit_b32 %2, %2, %3, %4", "v_alignbyte_b32 %3, %3, %0, %4");
      if (KIND == 14) BODY4("v_perm_b32 %0, %0, %4, %1", "v_perm_b32 %1, %1, %4, %2", "v_perm_b32 %2, %2, %4, %3", "v_perm_b32 %3, %3, %4, %0");
      if (KIND == 15) BODY4("v_or_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v_or_b32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2",
                            "v_or_b32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3", "v_or_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0");
      if (KIND == 16) BODY4("v_sad_u8 %0, %1, %4, %0", "v_sad_u8 %1, %2, %4, %1", "v_sad_u8 %2, %3, %4, %2", "v_sad_u8 %3, %0, %4, %3");
      if (KIND == 17) BODY4("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %2 row_shr:2 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %3, %0, %3 row_shr:4 row_mask:0xf bank_mask:0xf");
      if (KIND == 18) BODY4("v_lshrrev_b32 %0, 3, %0", "v_lshlrev_b32 %1, 1, %1", "v_sub_u32 %2, %2, %4", "v_subrev_u32 %3, %5, %3");
      if (KIND == 19) BODY4("v_bfi_b32 %0, %4, %0, %1", "v_bfi_b32 %1, %4, %1, %2", "v_xad_u32 %2, %2, %4, %3", "v_or3_b32 %3, %3, %4, %0");
      if (KIND == 20) BODY4("v_lshrrev_b32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD", "v_and_b32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD",
                            "v_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD", "v_mov_b32_sdwa %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3");
      if (KIND == 22) { unsigned long long m = 0x5555555555555555ull ^ seed;
        asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n\tv_cndmask_b32_e64 %1, %1, %4, %5\n\tv_cndmask_b32_e64 %2, %2, %4, %5\n\tv_cndmask_b32_e64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed), "s"(m)); }
      if (KIND == 23) asm volatile("v_cmp_lt_u32 vcc, %0, %4\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cmp_lt_u32 vcc, %1, %4\n\tv_cndmask_b32 %1, %1, %4, vcc\n\t"
                                   "v_cmp_lt_u32 vcc, %2, %4\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cmp_lt_u32 vcc, %3, %4\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed) : "vcc");
      if (KIND == 24) { unsigned long long m0, m1, m2, m3;
        asm volatile("v_cmp_lt_u32 %4, %0, %8\n\tv_cmp_lt_u32 %5, %1, %8\n\tv_cmp_lt_u32 %6, %2, %8\n\tv_cmp_lt_u32 %7, %3, %8\n\t"
                     "v_cndmask_b32_e64 %0, %0, %8, %4\n\tv_cndmask_b32_e64 %1, %1, %8, %5\n\tv_cndmask_b32_e64 %2, %2, %8, %6\n\tv_cndmask_b32_e64 %3, %3, %8, %7"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(seed)); }
      if (KIND == 25) { unsigned long long m0, m1, m2, m3, sv;      // eight masked LDS adds the way the tally issues them
        asm volatile("v_cmp_lt_u32 %0, %5, %9\n\tv_cmp_lt_u32 %1, %6, %9\n\tv_cmp_lt_u32 %2, %7, %9\n\tv_cmp_lt_u32 %3, %8, %9\n\t"
                     "s_mov_b64 %4, exec\n\ts_mov_b64 exec, %0\n\tds_add_u32 %10, %11\n\ts_mov_b64 exec, %1\n\tds_add_u32 %10, %11 offset:16\n\t"
                     "s_mov_b64 exec, %2\n\tds_add_u32 %10, %11 offset:32\n\ts_mov_b64 exec, %3\n\tds_add_u32 %10, %11 offset:48\n\ts_mov_b64 exec, %4"
                     : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(sv) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed), "v"((threadIdx.x & 63u) * 480u + (threadIdx.x >> 6) * 64u), "v"(1u) : "memory"); }
      if (KIND == 26) {      // the same adds, unconditional, value 0 / 1 from a bit field
        uint32_t v0, v1, v2, v3;
        asm volatile("v_bfe_u32 %0, %4, 0, 1\n\tv_bfe_u32 %1, %5, 1, 1\n\tv_bfe_u32 %2, %6, 2, 1\n\tv_bfe_u32 %3, %7, 3, 1\n\t"
                     "ds_add_u32 %8, %0\n\tds_add_u32 %8, %1 offset:16\n\tds_add_u32 %8, %2 offset:32\n\tds_add_u32 %8, %3 offset:48"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a), "v"(b), "v"(c), "v"(d), "v"((threadIdx.x & 63u) * 480u + (threadIdx.x >> 6) * 64u) : "memory"); }
      if (KIND == 21) { unsigned long long m0, m1, m2, m3, sv;      // the tally's shape: compare -> exec -> (no LDS op) -> restore
        asm volatile("v_cmp_lt_u32 %0, %5, %9\n\tv_cmp_lt_u32 %1, %6, %9\n\tv_cmp_lt_u32 %2, %7, %9\n\tv_cmp_lt_u32 %3, %8, %9\n\t"
                     "s_mov_b64 %4, exec\n\ts_mov_b64 exec, %0\n\ts_mov_b64 exec, %1\n\ts_mov_b64 exec, %2\n\ts_mov_b64 exec, %3\n\ts_mov_b64 exec, %4"
                     : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(sv) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed)); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int KIND>
static void run(const char* name) {
  uint32_t* d; hipMalloc(&d, (size_t)256 * 8 * 1024 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w : {4}) {
    const int threads = 256 * w, blocks = 256 * 4;      // four workgroups per CU's worth of waves in flight, all CUs
    rate_kernel<KIND><<<blocks, threads, 32768>>>(d, 10, 1);
    hipEventRecord(e0);
    rate_kernel<KIND><<<blocks, threads, 32768>>>(d, iters, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // every SIMD hosts w waves of each of the four workgroups that queue on its CU (one resident set at a time at 1024 threads)
    const double per = (double)ms * 1e6 / ((double)iters * 64 * w * 4);
    printf("%-22s %8.3f ms  %.3f ns per wave-instruction per SIMD\n", name, ms, per);
  }
  hipFree(d);
}

int main() {
  run<0>("plain and/xor/or/add"); run<18>("shift / sub"); run<1>("v_bfe_u32"); run<2>("v_and_or_b32"); run<3>("v_lshl_or_b32"); run<4>("v_add3_u32"); run<5>("v_lshl_add_u32");
  run<19>("bfi / xad / or3"); run<6>("v_cndmask (vcc)"); run<7>("v_cmp e32 -> vcc"); run<8>("v_cmp e64 -> sgpr"); run<9>("v_cmp_sdwa -> sgpr"); run<21>("v_cmp e64 + exec moves");
  run<22>("v_cndmask e64 (sgpr)"); run<23>("cmp->vcc + cndmask"); run<24>("cmp->sgpr + cndmask"); run<25>("cmp + exec + ds_add"); run<26>("bfe + ds_add (uncond)");
  run<10>("v_mad_u32_u24"); run<11>("v_mul_lo_u32"); run<12>("med3 / min / max"); run<13>("alignbit / alignbyte"); run<14>("v_perm_b32"); run<15>("v_or_b32_sdwa");
  run<20>("other sdwa ops"); run<16>("v_sad_u8"); run<17>("dpp mov / add");
  return 0;
}
