// dev probe: does issuing LDS-DMA beyond some number of outstanding instructions BLOCK the issuing wave?
// one workgroup of W waves per CU (grid 256 so that HBM is under realistic load), each wave issues N back-to-back
// buffer_load_dwordx4 ... lds (1 KB each) to cold (HBM) or warm (L2) addresses and stamps s_memtime before the first,
// after the last issue and after vmcnt(0).  Reported: cycles to ISSUE N, cycles until all N landed (wave 0 of block 0..3 avg).
// Also the same with plain buffer_load_dwordx4 into VGPRs.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_issue dma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)base;
  return i32x4{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void dma16s(i32x4 rsrc, uint32_t lds_byte, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
               :: "v"(voff), "s"(lds_byte), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ uint64_t now() { return __builtin_amdgcn_s_memtime(); }

template <int N, bool TOVGPR>
__global__ __launch_bounds__(1024) void k_issue(const unsigned char* src, uint32_t bytes, int warm, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nw = blockDim.x >> 6;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const i32x4 rs = make_rsrc(src, bytes);
  // every (block, wave, i) gets its own 1 KB; warm: touch them first with normal loads so that they sit in L2
  const uint32_t base = ((uint32_t)blockIdx.x * nw + wave) * (uint32_t)(N * 1024);
  float acc = 0.f;
  if (warm) {
    for (int i = 0; i < N; ++i) acc += *reinterpret_cast<const float*>(src + base + i * 1024 + lane * 16);
    if (acc == 1234.5f) sink[0] = acc;
  }
  __syncthreads();
  f32x4 r[N];
  const uint64_t t0 = now();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if constexpr (TOVGPR) {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r[i]) : "v"((uint32_t)lane * 16u), "s"(rs), "s"(base + i * 1024) : "memory");
    } else {
      dma16s(rs, lds0 + (uint32_t)(wave * N + i) * 1024u, (uint32_t)lane * 16u, base + i * 1024);
    }
  }
  const uint64_t t1 = now();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint64_t t2 = now();
  if constexpr (TOVGPR) {
#pragma unroll
    for (int i = 0; i < N; ++i) acc += r[i].x;
    if (acc == 1234.5f) sink[1] = acc;
  }
  if (lane == 0) {
    out[((size_t)blockIdx.x * nw + wave) * 2 + 0] = t1 - t0;
    out[((size_t)blockIdx.x * nw + wave) * 2 + 1] = t2 - t0;
  }
}

template <int N, bool V>
static void go(const unsigned char* x, size_t bytes, unsigned long long* out, unsigned long long* hout, float* sink, int waves, int warm) {
  const int grid = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_issue<N, V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t lds = V ? 0 : (size_t)waves * N * 1024;
  if (lds > 160 * 1024) return;
  double si = 0, sl = 0; int cnt = 0;
  for (int rep = 0; rep < 4; ++rep) {
    // flush L2 / MALL between cold runs: touch another 1 GB
    hipMemsetAsync((void*)(x + (1ull << 30)), rep, 1ull << 30, 0);
    hipLaunchKernelGGL((k_issue<N, V>), dim3(grid), dim3(64 * waves), lds, 0, x, (uint32_t)(512u << 20), warm, out, sink);
    hipMemcpy(hout, out, sizeof(unsigned long long) * 2 * grid * waves, hipMemcpyDeviceToHost);
    if (rep == 0) continue;
    for (int i = 0; i < grid * waves; ++i) { si += hout[2 * i]; sl += hout[2 * i + 1]; ++cnt; }
  }
  printf("%s N=%2d waves=%2d %s : issue %7.0f clk (%5.0f per instr)   all landed %7.0f clk  \n",
         V ? "vgpr-load" : "lds-dma  ", N, waves, warm ? "warm(L2)" : "cold(HBM)", si / cnt, si / cnt / N, sl / cnt);
  fflush(stdout);
}

int main() {
  unsigned char* x; hipMalloc(&x, 2ull << 30); hipMemset(x, 1, 2ull << 30);
  unsigned long long *out, *hout; hipMalloc(&out, 8 * 2 * 256 * 16); hout = (unsigned long long*)malloc(8 * 2 * 256 * 16);
  float* sink; hipMalloc(&sink, 16);
  printf("s_memtime tick = shader cycle (guide)\n");
  for (int warm : {0, 1}) for (int waves : {1, 4, 8}) {
    go<1, false>(x, 0, out, hout, sink, waves, warm);
    go<2, false>(x, 0, out, hout, sink, waves, warm);
    go<4, false>(x, 0, out, hout, sink, waves, warm);
    go<6, false>(x, 0, out, hout, sink, waves, warm);
    go<8, false>(x, 0, out, hout, sink, waves, warm);
    go<12, false>(x, 0, out, hout, sink, waves, warm);
    go<16, false>(x, 0, out, hout, sink, waves, warm);
  }
  for (int warm : {0, 1}) for (int waves : {1, 4, 8}) {
    go<1, true>(x, 0, out, hout, sink, waves, warm);
    go<4, true>(x, 0, out, hout, sink, waves, warm);
    go<8, true>(x, 0, out, hout, sink, waves, warm);
    go<16, true>(x, 0, out, hout, sink, waves, warm);
  }
  return 0;
}
