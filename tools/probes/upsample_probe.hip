// dev probe: where does the x2 bilinear forward on [16][64][64][256] -> [16][128][128][256] lose its time?
// build: hipcc --offload-arch=gfx950 -O3 -w -o upsample_probe upsample_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave per output pixel; NL = number of input loads (0, 1, 2, 4); REMAP = XCD-contiguous pixel order
template <int NL, int REMAP>
__global__ __launch_bounds__(256) void k_up(const float* __restrict__ x, float* __restrict__ y, int npix) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  int blk = blockIdx.x;
  if (REMAP) { const int q = gridDim.x >> 3; blk = (blk & 7) * q + (blk >> 3); }
  const int pix = blk * 4 + wave;
  const int ox = pix & 127, oy = (pix >> 7) & 127, n = pix >> 14;
  const int x0 = ox >> 1, y0 = oy >> 1, x1 = min(x0 + 1, 63), y1 = min(y0 + 1, 63);
  const float* b = x + (size_t)n * 64 * 64 * 256 + lane * 4;
  f32x4 v = {1.f, 2.f, 3.f, 4.f};
  if (NL >= 1) v = *reinterpret_cast<const f32x4*>(b + (y0 * 64 + x0) * 256);
  if (NL >= 2) v += *reinterpret_cast<const f32x4*>(b + (y0 * 64 + x1) * 256);
  if (NL >= 4) {
    v += *reinterpret_cast<const f32x4*>(b + (y1 * 64 + x0) * 256);
    v += *reinterpret_cast<const f32x4*>(b + (y1 * 64 + x1) * 256);
  }
  *reinterpret_cast<f32x4*>(y + (size_t)pix * 256 + lane * 4) = v;
}
// one THREAD per 16 bytes, linear order (the element-per-thread form), 4 loads
__global__ __launch_bounds__(256) void k_up_thread(const float* __restrict__ x, float* __restrict__ y) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int cb = i & 63, pix = (int)(i >> 6);
  const int ox = pix & 127, oy = (pix >> 7) & 127, n = pix >> 14;
  const int x0 = ox >> 1, y0 = oy >> 1, x1 = min(x0 + 1, 63), y1 = min(y0 + 1, 63);
  const float* b = x + (size_t)n * 64 * 64 * 256 + cb * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(b + (y0 * 64 + x0) * 256);
  v += *reinterpret_cast<const f32x4*>(b + (y0 * 64 + x1) * 256);
  v += *reinterpret_cast<const f32x4*>(b + (y1 * 64 + x0) * 256);
  v += *reinterpret_cast<const f32x4*>(b + (y1 * 64 + x1) * 256);
  *reinterpret_cast<f32x4*>(y + i * 4) = v;
}
// output tile per workgroup: 2 output rows x 32 output pixels (both rows read the same input rows)
__global__ __launch_bounds__(256) void k_up_tile(const float* __restrict__ x, float* __restrict__ y) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  // block -> (n, oy pair, ox group of 32); wave -> 8 consecutive ox of both rows
  const int blk = blockIdx.x;
  const int gx = blk & 3, oyp = (blk >> 2) & 63, n = blk >> 8;
  const float* b = x + (size_t)n * 64 * 64 * 256 + lane * 4;
  for (int r = 0; r < 2; ++r) {
    const int oy = oyp * 2 + r, y0 = oy >> 1, y1 = min(y0 + 1, 63);
    for (int k = 0; k < 8; ++k) {
      const int ox = gx * 32 + wave * 8 + k, x0 = ox >> 1, x1 = min(x0 + 1, 63);
      f32x4 v = *reinterpret_cast<const f32x4*>(b + (y0 * 64 + x0) * 256);
      v += *reinterpret_cast<const f32x4*>(b + (y0 * 64 + x1) * 256);
      v += *reinterpret_cast<const f32x4*>(b + (y1 * 64 + x0) * 256);
      v += *reinterpret_cast<const f32x4*>(b + (y1 * 64 + x1) * 256);
      *reinterpret_cast<f32x4*>(y + ((size_t)(n * 128 + oy) * 128 + ox) * 256 + lane * 4) = v;
    }
  }
}

template <class F> static void run(const char* name, F launch) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(s);
  for (int i = 0; i < 20; ++i) launch();
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  printf("%-34s %8.1f us\n", name, ms / 20 * 1e3);
}
int main() {
  const size_t nin = (size_t)16 * 64 * 64 * 256, nout = nin * 4;
  float *x, *y; hipMalloc(&x, nin * 4); hipMalloc(&y, nout * 4);
  hipMemset(x, 0, nin * 4);
  const int npix = 16 * 128 * 128;
  run("wave/pixel store only", [&] { hipLaunchKernelGGL((k_up<0, 0>), dim3(npix / 4), dim3(256), 0, 0, x, y, npix); });
  run("wave/pixel 1 load", [&] { hipLaunchKernelGGL((k_up<1, 0>), dim3(npix / 4), dim3(256), 0, 0, x, y, npix); });
  run("wave/pixel 2 loads", [&] { hipLaunchKernelGGL((k_up<2, 0>), dim3(npix / 4), dim3(256), 0, 0, x, y, npix); });
  run("wave/pixel 4 loads", [&] { hipLaunchKernelGGL((k_up<4, 0>), dim3(npix / 4), dim3(256), 0, 0, x, y, npix); });
  run("wave/pixel 4 loads, XCD remap", [&] { hipLaunchKernelGGL((k_up<4, 1>), dim3(npix / 4), dim3(256), 0, 0, x, y, npix); });
  run("wave/pixel 1 load, XCD remap", [&] { hipLaunchKernelGGL((k_up<1, 1>), dim3(npix / 4), dim3(256), 0, 0, x, y, npix); });
  run("wave/pixel store only, XCD remap", [&] { hipLaunchKernelGGL((k_up<0, 1>), dim3(npix / 4), dim3(256), 0, 0, x, y, npix); });
  run("thread/16B linear, 4 loads", [&] { hipLaunchKernelGGL(k_up_thread, dim3(nout / 4 / 256), dim3(256), 0, 0, x, y); });
  run("tile 2x32 per workgroup", [&] { hipLaunchKernelGGL(k_up_tile, dim3(16 * 64 * 4), dim3(256), 0, 0, x, y); });
  return 0;
}
