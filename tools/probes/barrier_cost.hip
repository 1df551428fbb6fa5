// probe: cost of one s_barrier round (all waves of a workgroup, nothing else) as a function of waves per workgroup,
// workgroups per CU and a little SALU / VALU work between barriers.  hipcc --offload-arch=gfx950 -O3 barrier_cost.hip -o barrier_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(int n, int work, float* out, long long* cyc) {
  extern __shared__ float lds[];
  float a = threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int w = 0; w < work; ++w) a = a * 1.0001f + 0.5f;
  }
  long long t1 = clock64();
  if (a == 12345.f) out[0] = a + lds[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4); hipMalloc(&cyc, 8);
  const int n = 4096;
  for (int lds : {0, 65536, 160 * 1024}) for (int waves : {4, 8, 12, 16}) for (int wgs : {256, 512}) for (int work : {0, 32}) {
    if (lds > 65536 && wgs > 256) continue;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<<<wgs, waves * 64, lds>>>(n, work, out, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<<<wgs, waves * 64, lds>>>(n, work, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("lds %6d waves %2d wgs %3d work %2d: %.1f ns per barrier round, %lld clock64 ticks per round\n", lds, waves, wgs, work, ms * 1e6 / n, c / n);
  }
  return 0;
}
