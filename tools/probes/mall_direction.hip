// dev probe (GPU box): does a streaming pass that walks its input in the direction OPPOSITE to the pass that wrote it find the
// producer's tail in the 256 MB memory-side cache?  writer: block b writes chunk b (ascending dispatch order); reader: block b
// reads chunk b (same direction) or chunk nblk-1-b (opposite), one f4 per thread, sums into a sink.  A third mode reads and
// writes (the shape of a BatchNorm apply pass).  build: hipcc --offload-arch=gfx950 -O3 mall_direction.hip -o mall_direction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NTS>
__global__ __launch_bounds__(256) void writer(f4* p, size_t n4, float v) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) { if (NTS) __builtin_nontemporal_store(f4{v, v + 1, v + 2, v + 3}, p + i); else p[i] = f4{v, v + 1, v + 2, v + 3}; }
}
template <bool NT>
__global__ __launch_bounds__(256) void reader(const f4* p, size_t n4, int rev, float* sink) {
  const size_t b = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const size_t i = b * 256 + threadIdx.x;
  f4 v = f4{0, 0, 0, 0};
  if (i < n4) v = NT ? __builtin_nontemporal_load(p + i) : p[i];
  if (v.x + v.y + v.z + v.w == 12345.678f) sink[0] = 1.f;
}
template <bool NT, bool NTS = false>
__global__ __launch_bounds__(256) void rw(const f4* p, f4* q, size_t n4, int rev) {
  const size_t b = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const size_t i = b * 256 + threadIdx.x;
  if (i < n4) {
    f4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
    v.x = v.x * 1.5f + 1.f; v.y = v.y * 1.5f + 1.f; v.z = v.z * 1.5f + 1.f; v.w = v.w * 1.5f + 1.f;
    if (NTS) __builtin_nontemporal_store(v, q + i); else q[i] = v;
  }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const size_t maxb = (size_t)1 << 30;
  f4 *a, *b; float* sink;
  CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t mb : {64, 128, 256, 384, 512}) {
    const size_t n4 = mb * 1024 * 1024 / 16; const unsigned nblk = (unsigned)((n4 + 255) / 256);
    for (int wnt = 0; wnt < 2; ++wnt)
    for (int mode = 0; mode < 5; ++mode)
      for (int rev = 0; rev < 2; ++rev) {
        std::vector<float> t;
        for (int it = 0; it < 9; ++it) {
          if (wnt) hipLaunchKernelGGL(writer<true>, dim3(nblk), dim3(256), 0, 0, a, n4, (float)it); else hipLaunchKernelGGL(writer<false>, dim3(nblk), dim3(256), 0, 0, a, n4, (float)it);
          CK(hipEventRecord(e0));
          if (mode == 0) hipLaunchKernelGGL(reader<false>, dim3(nblk), dim3(256), 0, 0, a, n4, rev, sink);
          else if (mode == 1) hipLaunchKernelGGL(rw<false>, dim3(nblk), dim3(256), 0, 0, a, b, n4, rev);
          else if (mode == 2) hipLaunchKernelGGL(reader<true>, dim3(nblk), dim3(256), 0, 0, a, n4, rev, sink);
          else if (mode == 3) hipLaunchKernelGGL(rw<true>, dim3(nblk), dim3(256), 0, 0, a, b, n4, rev);
          else hipLaunchKernelGGL((rw<true, true>), dim3(nblk), dim3(256), 0, 0, a, b, n4, rev);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        const double bytes = (double)mb * 1048576 * ((mode & 1) ? 2 : 1);
        printf("%5zu MB producer %s | %s %s: min %7.1f med %7.1f us  %.2f TB/s (med)\n", mb, wnt ? "nt store" : "store   ", mode == 0 ? "read         " : mode == 1 ? "read+write   " : mode == 2 ? "read nt      " : mode == 3 ? "read+write nt" : "rd nt + wr nt", rev ? "opposite" : "same    ",
               t[0], t[t.size() / 2], bytes / t[t.size() / 2] / 1e6);
      }
  }
  return 0;
}
