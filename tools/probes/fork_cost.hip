// probe: what a cross-stream fork costs the stream that is forked FROM.  A chain of N dependent ~T us kernels on stream A
// (a) alone, (b) with hipEventRecord after every kernel, (c) + a second stream waiting for each event and running a small
// kernel behind it (the weight-gradient fork of DESIGN 2.8), (d) the same fork with the event attached to the kernel's own
// completion signal (hipExtLaunchKernelGGL stopEvent) instead of a marker packet, (e) one fork per 4 kernels.
// hipcc --offload-arch=gfx950 -O3 fork_cost.hip -o fork_cost && ./fork_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <chrono>
#include <vector>
__global__ void spin(float* p, int iters) {
  float a = p[threadIdx.x & 63];
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) p[0] = a;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  float* buf; hipMalloc(&buf, 4096); hipMemset(buf, 0, 4096);
  hipStream_t A, B; hipStreamCreateWithFlags(&A, hipStreamNonBlocking); hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
  const int N = 400;
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (int iters : {2000, 10000}) {
    for (int grid : {256, 2048}) {
      auto run = [&](int mode) {
        hipDeviceSynchronize();
        double t0 = now();
        for (int i = 0; i < N; ++i) {
          const bool fork = mode == 4 ? (i % 4 == 3) : (mode >= 1);
          if (mode == 3 && fork) {
            hipExtLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, A, nullptr, ev[i], 0, buf, iters);
          } else {
            hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, A, buf, iters);
            if (fork) hipEventRecord(ev[i], A);
          }
          if (fork && mode >= 2) {
            hipStreamWaitEvent(B, ev[i], 0);
            hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, B, buf + 512, 200);
          }
        }
        hipStreamSynchronize(A);
        double t1 = now();
        hipDeviceSynchronize();
        return (t1 - t0) / N;
      };
      run(0);
      const char* names[5] = {"chain alone", "+ event record each", "+ fork each (record, B waits, B kernel)", "fork each via ext-launch stopEvent",
                              "fork every 4th (record)"};
      for (int m = 0; m < 5; ++m) {
        double best = 1e9;
        for (int r = 0; r < 3; ++r) { double t = run(m); if (t < best) best = t; }
        printf("iters %5d grid %4d  %-44s %.2f us per kernel\n", iters, grid, names[m], best);
      }
    }
  }
  return 0;
}
