// dev probe: sustained v_mfma_f32_32x32x16_bf16 rate from registers (no memory traffic), random operands.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_bf16_peak.hip -o /tmp/mfma_peak ; run: /tmp/mfma_peak [waves_per_simd] [seconds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int F32>
__global__ __launch_bounds__(256) void k(const u32x4* in, float* out, int iters) {
  u32x4 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 256], b0 = in[threadIdx.x + 512], b1 = in[threadIdx.x + 768];
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (F32) {
        c[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a0.x), __builtin_bit_cast(float, b0.x), c[0], 0, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a0.y), __builtin_bit_cast(float, b1.x), c[1], 0, 0, 0);
        c[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a1.x), __builtin_bit_cast(float, b0.y), c[2], 0, 0, 0);
        c[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a1.y), __builtin_bit_cast(float, b1.y), c[3], 0, 0, 0);
      } else {
        c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), c[0], 0, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b1), c[1], 0, 0, 0);
        c[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b0), c[2], 0, 0, 0);
        c[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), c[3], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  int wg_per_cu = argc > 1 ? atoi(argv[1]) : 1;
  double secs = argc > 2 ? atof(argv[2]) : 3.0;
  int f32 = argc > 3 ? atoi(argv[3]) : 0;
  unsigned* h = (unsigned*)malloc(1024 * 16);
  srand(1);
  for (int i = 0; i < 4096; ++i) {
    // random bf16 pairs / floats with exponents around 1.0
    unsigned m0 = rand() & 0x7f, m1 = rand() & 0x7f, s0 = rand() & 1, s1 = rand() & 1;
    unsigned lo = (s0 << 15) | (0x3f80 - ((rand() & 3) << 7)) | m0, hi = (s1 << 15) | (0x3f80 - ((rand() & 3) << 7)) | m1;
    h[i] = f32 ? ((hi << 16) | (rand() & 0xffff)) : ((hi << 16) | lo);
  }
  u32x4* din; float* dout;
  hipMalloc(&din, 1024 * 16); hipMalloc(&dout, 256 * 8 * 256 * 4);
  hipMemcpy(din, h, 1024 * 16, hipMemcpyHostToDevice);
  const int grid = 256 * wg_per_cu, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto t0 = std::chrono::steady_clock::now();
  double best = 0, last = 0; int n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    hipEventRecord(e0);
    if (f32) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, din, dout, iters);
    else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, din, dout, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 16 * (f32 ? 4096.0 : 32768.0);
    last = flops / (ms * 1e-3) / 1e12; if (last > best) best = last; ++n;
  }
  printf("%s wg/cu=%d: first-best %.1f TF, sustained (last of %d launches) %.1f TF\n", f32 ? "f32 32x32x2" : "bf16 32x32x16", wg_per_cu, best, n, last);
  return 0;
}
