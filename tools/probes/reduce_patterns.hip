// dev probe: per-channel reduction over two [rows][C] fp32 streams (the bn_bwd_partial shape) by work distribution.
// build: hipcc --offload-arch=gfx950 -O3 -w -o reduce_patterns reduce_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: span (block owns contiguous rows), U rows in flight.  MODE 1: interleaved (row groups dealt round-robin).
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_reduce(const float* __restrict__ dy, const float* __restrict__ x,
                                                float* __restrict__ partial, long rows, int C, long rows_per_blk,
                                                int tpc, int rl) {
  __shared__ f32x4 red[2][256];
  const int c4 = C >> 2;
  const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
  for (int cb = tc; cb < c4; cb += tpc) {
    f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
    long r, r1, st;
    if (MODE == 0) {
      const long r0 = (long)blockIdx.x * rows_per_blk;
      r1 = min(rows, r0 + rows_per_blk); r = r0 + tr; st = rl;
    } else {
      r = (long)blockIdx.x * rl + tr; r1 = rows; st = (long)gridDim.x * rl;
    }
    for (; r + (U - 1) * st < r1; r += U * st) {
      f32x4 g[U], v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        g[j] = *reinterpret_cast<const f32x4*>(dy + (r + j * st) * C + cb * 4);
        v[j] = *reinterpret_cast<const f32x4*>(x + (r + j * st) * C + cb * 4);
      }
#pragma unroll
      for (int j = 0; j < U; ++j) { s += g[j]; q += g[j] * v[j]; }
    }
    for (; r < r1; r += st) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * C + cb * 4);
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + cb * 4);
      s += g; q += g * v;
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    if (tr == 0) {
      for (int k = 1; k < rl; ++k) { s += red[0][k * tpc + tc]; q += red[1][k * tpc + tc]; }
      float* o = partial + (size_t)blockIdx.x * 2 * C;
      *reinterpret_cast<f32x4*>(o + cb * 4) = s;
      *reinterpret_cast<f32x4*>(o + C + cb * 4) = q;
    }
    __syncthreads();
  }
}

template <class F> static void run(const char* name, F launch, double bytes) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(s);
  const int R = 20;
  for (int i = 0; i < R; ++i) launch();
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  printf("  %-26s %8.1f us  %6.2f TB/s\n", name, ms / R * 1e3, bytes / (ms / R * 1e-3) / 1e12);
}

int main() {
  const int Cs[] = {64, 256, 512, 1024};
  const long Rs[] = {1048576, 262144, 65536, 16384};
  float *dy, *x, *partial;
  hipMalloc(&dy, 268435456); hipMalloc(&x, 268435456); hipMalloc(&partial, 8192 * 2 * 2048 * 4);
  hipMemset(dy, 0, 268435456); hipMemset(x, 0, 268435456);
  for (int t = 0; t < 4; ++t) {
    const int C = Cs[t]; const long rows = Rs[t];
    const int c4 = C / 4, tpc = c4 < 256 ? c4 : 256, rl = 256 / tpc;
    const double bytes = 2.0 * rows * C * 4;
    printf("C=%d rows=%ld (%.0f MB x 2 streams)\n", C, rows, bytes / 2e6);
    for (int G : {1024, 2048, 4096, 8192}) {
      long rpb = (rows + G - 1) / G; rpb = (rpb + rl - 1) / rl * rl;
      const int nb = (int)((rows + rpb - 1) / rpb);
      char nm[64];
      snprintf(nm, 64, "span U=4 G=%d", nb);
      run(nm, [&] { hipLaunchKernelGGL((k_reduce<0, 4>), dim3(nb), dim3(256), 0, 0, dy, x, partial, rows, C, rpb, tpc, rl); }, bytes);
      snprintf(nm, 64, "span U=2 G=%d", nb);
      run(nm, [&] { hipLaunchKernelGGL((k_reduce<0, 2>), dim3(nb), dim3(256), 0, 0, dy, x, partial, rows, C, rpb, tpc, rl); }, bytes);
      snprintf(nm, 64, "interleaved U=1 G=%d", G);
      run(nm, [&] { hipLaunchKernelGGL((k_reduce<1, 1>), dim3(G), dim3(256), 0, 0, dy, x, partial, rows, C, rpb, tpc, rl); }, bytes);
      snprintf(nm, 64, "interleaved U=2 G=%d", G);
      run(nm, [&] { hipLaunchKernelGGL((k_reduce<1, 2>), dim3(G), dim3(256), 0, 0, dy, x, partial, rows, C, rpb, tpc, rl); }, bytes);
      snprintf(nm, 64, "interleaved U=4 G=%d", G);
      run(nm, [&] { hipLaunchKernelGGL((k_reduce<1, 4>), dim3(G), dim3(256), 0, 0, dy, x, partial, rows, C, rpb, tpc, rl); }, bytes);
    }
  }
  return 0;
}
