// probe: do ds_read_b128 returns steal time from the matrix pipe?  Each wave loops over groups of 12 MFMAs (32x32x16 f16,
// two accumulators) with R independent ds_read_b128 issued in front of each group (consumed only at the end), no barriers.
// hipcc --offload-arch=gfx950 -O3 mfma_lds_mix.hip -o mfma_lds_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int R, int USE>
__global__ __launch_bounds__(512) void k(int n, float* out) {
  extern __shared__ unsigned char lds[];
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
  f32x16 acc0 = {0}, acc1 = {0};
  f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 2, 2, 2, 2};
  u32x4 sink = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    u32x4 r[R > 0 ? R : 1];
#pragma unroll
    for (int j = 0; j < R; ++j) r[j] = *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)(base + j * 1024 * 0 + ((i + j) & 3) * 1024);
    __builtin_amdgcn_sched_barrier(0);
    f16x8 aa = a, bb = b;
    if (USE) {   // operands come from the reads of the PREVIOUS iteration (sink), as in the kernels
      aa = __builtin_bit_cast(f16x8, sink);
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa, bb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bb, aa, acc1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < R; ++j) sink ^= r[j];
  }
  if (acc0[0] + acc1[3] == 12345.f || sink[0] == 77) out[0] = acc0[1] + sink[1];
}
template <int R, int USE>
void run(int waves, int n, float* out) {
  hipFuncSetAttribute((const void*)k<R, USE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  k<R, USE><<<256, waves * 64, 65536>>>(n, out);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<R, USE><<<256, waves * 64, 65536>>>(n, out); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)n * 12 * (waves / 4.0);   // MFMAs per SIMD
  printf("waves/CU %d  reads/12 MFMA %2d use %d: %.1f us, %.1f ns per MFMA per SIMD (32 cycles = %.1f ns at 2.1 GHz)\n", waves, R, USE, ms * 1e3, ms * 1e6 / mf, 32 / 2.1);
}
int main() {
  float* out; (void)hipMalloc(&out, 4);
  const int n = 2000;
  for (int waves : {4, 8}) {
    run<0, 0>(waves, n, out); run<6, 0>(waves, n, out); run<12, 0>(waves, n, out); run<24, 0>(waves, n, out);
    run<12, 1>(waves, n, out);
  }
  return 0;
}
