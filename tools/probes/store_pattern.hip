// dev probe: HBM write rate of the igemm epilogue's store pattern (per instruction: 32 pixels x 32 contiguous bytes,
// 1 KB apart; four instructions complete a 128-byte line) against a linear pattern (1 KB contiguous per instruction).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/store_pattern.hip -o tools/probes/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// tile: 128 pixels x 128 channels (fp32), row pitch C floats; one workgroup of 256 threads per tile
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int C, int tiles_n, float v) {
  const int tile = blockIdx.x, tn = tile % tiles_n, tm = tile / tiles_n;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const f32x4 val = {v, v + 1, v + 2, v + 3};
  if (MODE == 0) {  // igemm epilogue pattern
    for (int a = 0; a < 2; ++a) {
      const size_t row = (size_t)tm * 128 + wm * 64 + a * 32 + li;
      for (int b = 0; b < 2; ++b)
        for (int r4 = 0; r4 < 4; ++r4) {
          const int col = tn * 128 + wn * 64 + b * 32 + 8 * r4 + 4 * lh;
          *reinterpret_cast<f32x4*>(out + row * C + col) = val;
        }
    }
  } else {  // same bytes, each instruction writes whole rows: 16 lanes x 16 B = 256 B contiguous per pixel row
    for (int j = 0; j < 16; ++j) {
      const size_t row = (size_t)tm * 128 + wm * 64 + j * 4 + (lane >> 4);
      const int col = tn * 128 + wn * 64 + (lane & 15) * 4;
      *reinterpret_cast<f32x4*>(out + row * C + col) = val;
    }
  }
}
int main() {
  const int M = 262144, C = 256;
  float* d; hipMalloc(&d, (size_t)M * C * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int tiles = (M / 128) * (C / 128);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(tiles), dim3(256), 0, 0, d, C, C / 128, (float)i);
        else hipLaunchKernelGGL(k<1>, dim3(tiles), dim3(256), 0, 0, d, C, C / 128, (float)i);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d: %.1f us per 268 MB  = %.2f TB/s\n", mode, ms * 100, (double)M * C * 4 / (ms / 10 * 1e-3) / 1e12);
    }
  return 0;
}
