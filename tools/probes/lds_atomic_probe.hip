// Developer probe (not part of the library): what a returnless LDS atomic costs on gfx950 in the pileup kernel's access
// patterns -- 16 wavefronts per CU (two 512-thread workgroups with 64 KiB of tallies each), every lane (read g, chunk c)
// adding at site pos_g + 30 c + j in instruction j = 0..29.
//   0 conflict-free                 address = lane * 4
//   1 [site][A,C,G,T] u32 (16 B per site, what the kernel does)
//   2 [site][AC | GT] two u16 per dword (8 B per site)
//   3 [site] four u8 per dword (4 B per site)
//   4 pattern 1, all lanes of the wave on ONE read-like run (c = lane, no second read): the within-read spacing alone
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/lds_atomic_probe.hip -o /tmp/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int PAT, int J>
__device__ __forceinline__ void one_add(uint32_t abase, uint32_t codes, uint32_t one) {
  const uint32_t code = (codes >> (2 * (J & 15))) & 3u;
  if (PAT == 0) {
    asm volatile("ds_add_u32 %0, %1" :: "v"(abase), "v"(one) : "memory");
  } else if (PAT == 1 || PAT == 4) {
    const uint32_t a = abase | (code << 2);
    asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(a), "v"(one), "n"(16 * J) : "memory");
  } else if (PAT == 2) {
    const uint32_t a = abase | ((code >> 1) << 2);
    const uint32_t v = 1u << (16 * (code & 1u));
    asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(a), "v"(v), "n"(8 * J) : "memory");
  } else {
    const uint32_t v = 1u << (8 * code);
    asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(abase), "v"(v), "n"(4 * J) : "memory");
  }
}
template <int PAT, int J>
struct Unroll {
  static __device__ __forceinline__ void run(uint32_t abase, uint32_t codes, uint32_t one) {
    one_add<PAT, 30 - J>(abase, codes, one);
    Unroll<PAT, J - 1>::run(abase, codes, one);
  }
};
template <int PAT>
struct Unroll<PAT, 0> {
  static __device__ __forceinline__ void run(uint32_t, uint32_t, uint32_t) {}
};

template <int PAT>
__global__ __launch_bounds__(512, 4) void probe(int iters, uint32_t seed, unsigned long long* cycles, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 4096];
  for (int i = threadIdx.x; i < 4 * 4096; i += 512) lds[i] = 0u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / 5, c = lane - 5 * g;
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds;
  uint32_t state = seed ^ (blockIdx.x * 9781u + wave * 7919u);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    state = state * 1664525u + 1013904223u;
    // one start per read of the wave: a hash of (state, g), the same in the five lanes of a read
    uint32_t h = (state ^ (g * 0x9E3779B9u)) * 0x85EBCA6Bu;
    h ^= h >> 13;
    const uint32_t pos = (h >> 8) % 3900u;
    const uint32_t site = PAT == 4 ? (state >> 8) % 2000u + 30u * lane : pos + 30u * c;
    uint32_t codes = (h ^ (lane * 0x27D4EB2Fu)) * 0xC2B2AE35u;      // 16 two-bit codes per lane
    codes ^= codes >> 15;
    uint32_t abase;
    if (PAT == 0) abase = lds_base + 4u * lane + 256u * (it & 31);
    else if (PAT == 1 || PAT == 4) abase = lds_base + (site << 4);
    else if (PAT == 2) abase = lds_base + (site << 3);
    else abase = lds_base + (site << 2);
    if (g < 12 || PAT == 0 || PAT == 4) Unroll<PAT, 30>::run(abase, codes, 1u);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
  uint32_t acc = 0;
  for (int i = threadIdx.x; i < 4 * 4096; i += 512) acc += lds[i];
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <int PAT>
void run(const char* name, int iters) {
  unsigned long long* d_cycles;
  uint32_t* d_sink;
  const int grid = 512;
  hipMalloc(&d_cycles, grid * 8 * 8);
  hipMalloc(&d_sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<PAT>, dim3(grid), dim3(512), 0, 0, 10, 1u, d_cycles, d_sink);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<PAT>, dim3(grid), dim3(512), 0, 0, iters, 12345u, d_cycles, d_sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> cyc(grid * 8);
  hipMemcpy(cyc.data(), d_cycles, grid * 8 * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : cyc) mean += (double)v;
  mean /= cyc.size();
  // per CU: 16 waves x iters x 30 wave-instructions share one LDS
  const double per_cu = 16.0 * iters * 30.0;
  printf("%-44s %8.3f ms   wave cycles %10.0f   LDS cycles per ds_add wave-instruction (per CU): %.2f (by s_memtime)  %.2f (by wall clock at 2.4 GHz)\n",
         name, ms, mean, mean / per_cu, ms * 1e-3 * 2.4e9 / per_cu);
  hipFree(d_cycles);
  hipFree(d_sink);
}

int main() {
  const int iters = 2000;
  run<0>("0 conflict-free", iters);
  run<1>("1 [site][A,C,G,T] u32, 12 reads x 5 lanes", iters);
  run<2>("2 [site] two u16 pairs (8 B / site)", iters);
  run<3>("3 [site] four u8 (4 B / site)", iters);
  run<4>("4 one run, lanes 30 sites apart, u32", iters);
  return 0;
}
