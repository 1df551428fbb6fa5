// dev probe: what does a 1-read + 1-write streaming kernel reach on 268 MB maps, by work distribution?
// build: hipcc --offload-arch=gfx950 -O3 -o copy_patterns copy_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 op(f32x4 v) { return v * 1.0001f + 0.5f; }

// V0: grid-stride, 4 loads S apart (the bn_apply structure)
__global__ __launch_bounds__(256) void k_gridstride4(const f32x4* x, f32x4* y, size_t n4) {
  const size_t S = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * S < n4; i += 4 * S) {
    const f32x4 a0 = x[i], a1 = x[i + S], a2 = x[i + 2 * S], a3 = x[i + 3 * S];
    y[i] = op(a0); y[i + S] = op(a1); y[i + 2 * S] = op(a2); y[i + 3 * S] = op(a3);
  }
  for (; i < n4; i += S) y[i] = op(x[i]);
}
// V1: one shot, workgroup owns U*4 KB contiguous
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_oneshot(const f32x4* x, f32x4* y, size_t n4) {
  const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  f32x4 a[U];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const size_t i = base + 256 * j;
    if (i < n4) a[j] = NT ? __builtin_nontemporal_load(x + i) : x[i];
  }
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const size_t i = base + 256 * j;
    if (i < n4) { if (NT) __builtin_nontemporal_store(op(a[j]), y + i); else y[i] = op(a[j]); }
  }
}
// V2: persistent, each workgroup walks its own contiguous span, U in flight
template <int U>
__global__ __launch_bounds__(256) void k_span(const f32x4* x, f32x4* y, size_t n4) {
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
  size_t i = lo + threadIdx.x;
  for (; i + 256 * (U - 1) < hi; i += 256 * U) {
    f32x4 a[U];
#pragma unroll
    for (int j = 0; j < U; ++j) a[j] = x[i + 256 * j];
#pragma unroll
    for (int j = 0; j < U; ++j) y[i + 256 * j] = op(a[j]);
  }
  for (; i < hi; i += 256) y[i] = op(x[i]);
}
// V3: grid-stride, adjacent trips (stride S but grid covers chip exactly: 256 CUs x 8)
__global__ __launch_bounds__(256) void k_gridstride1(const f32x4* x, f32x4* y, size_t n4) {
  const size_t S = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += S) y[i] = op(x[i]);
}

template <class F> static void run(const char* name, F launch, size_t bytes) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(s);
  const int R = 20;
  for (int i = 0; i < R; ++i) launch();
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  printf("%-28s %8.1f us  %6.2f TB/s\n", name, ms / R * 1e3, 2.0 * bytes / (ms / R * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 268;
  const size_t n4 = mb * 1000000 / 16, bytes = n4 * 16;
  f32x4 *x, *y; hipMalloc(&x, bytes); hipMalloc(&y, bytes);
  hipMemset(x, 0, bytes); hipMemset(y, 0, bytes);
  printf("%zu MB in + %zu MB out\n", bytes / 1000000, bytes / 1000000);
  for (int g : {1024, 2048, 4096, 8192}) {
    char nm[64]; snprintf(nm, 64, "gridstride4 grid=%d", g);
    run(nm, [&] { hipLaunchKernelGGL(k_gridstride4, dim3(g), dim3(256), 0, 0, x, y, n4); }, bytes);
  }
  for (int g : {2048, 4096, 16384}) {
    char nm[64]; snprintf(nm, 64, "gridstride1 grid=%d", g);
    run(nm, [&] { hipLaunchKernelGGL(k_gridstride1, dim3(g), dim3(256), 0, 0, x, y, n4); }, bytes);
  }
  run("oneshot U=1", [&] { hipLaunchKernelGGL((k_oneshot<1, false>), dim3((n4 + 255) / 256), dim3(256), 0, 0, x, y, n4); }, bytes);
  run("oneshot U=4", [&] { hipLaunchKernelGGL((k_oneshot<4, false>), dim3((n4 + 1023) / 1024), dim3(256), 0, 0, x, y, n4); }, bytes);
  run("oneshot U=8", [&] { hipLaunchKernelGGL((k_oneshot<8, false>), dim3((n4 + 2047) / 2048), dim3(256), 0, 0, x, y, n4); }, bytes);
  run("oneshot U=4 nt", [&] { hipLaunchKernelGGL((k_oneshot<4, true>), dim3((n4 + 1023) / 1024), dim3(256), 0, 0, x, y, n4); }, bytes);
  run("oneshot U=8 nt", [&] { hipLaunchKernelGGL((k_oneshot<8, true>), dim3((n4 + 2047) / 2048), dim3(256), 0, 0, x, y, n4); }, bytes);
  for (int g : {1024, 2048, 4096}) {
    char nm[64]; snprintf(nm, 64, "span U=4 grid=%d", g);
    run(nm, [&] { hipLaunchKernelGGL((k_span<4>), dim3(g), dim3(256), 0, 0, x, y, n4); }, bytes);
  }
  run("span U=8 grid=2048", [&] { hipLaunchKernelGGL((k_span<8>), dim3(2048), dim3(256), 0, 0, x, y, n4); }, bytes);
  run("hipMemcpyDtoD", [&] { hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0); }, bytes);
  return 0;
}
