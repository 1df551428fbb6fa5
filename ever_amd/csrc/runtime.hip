// Error reporting and ABI identification for libever_hip.so.
#include "common.hpp"
#include <string.h>

namespace evk {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace evk

extern "C" const char* evk_last_error(void) { return evk::g_err; }
extern "C" int evk_abi_version(void) { return 21; }
extern "C" const char* evk_build_arch(void) { return "gfx950"; }

// Fork of one stream from another without a torch Event object per call (53 weight gradients per step fork the side
// stream from the backward's stream): a ring of timing-less events per device, recorded on `from` and waited for on `to`.
// A wait refers to the record that preceded it, so an event of the ring can be recorded again at once.
namespace {
constexpr int kForkEvents = 64;
struct ForkRing {
  hipEvent_t ev[kForkEvents];
  int next = 0;
  bool made = false;
};
ForkRing g_fork[16];
}  // namespace

extern "C" int evk_stream_fork(void* from, void* to) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
    evk::set_error("evk_stream_fork: no current device");
    return 1;
  }
  ForkRing& r = g_fork[dev];
  if (!r.made) {
    for (int i = 0; i < kForkEvents; ++i)
      if (hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
        evk::set_error("evk_stream_fork: hipEventCreateWithFlags failed");
        return 1;
      }
    r.made = true;
  }
  hipEvent_t e = r.ev[r.next];
  r.next = (r.next + 1) % kForkEvents;
  hipError_t err = hipEventRecord(e, static_cast<hipStream_t>(from));
  if (err == hipSuccess) err = hipStreamWaitEvent(static_cast<hipStream_t>(to), e, 0);
  if (err != hipSuccess) {
    evk::set_error("evk_stream_fork: %s", hipGetErrorString(err));
    return 1;
  }
  return 0;
}

// Do two streams of this process run kernels at the same time?  HIP multiplexes its streams onto a few hardware queues
// (4 by default); two streams that share one are serialised, whatever their events say.  The weight-gradient side stream
// is only worth having on another queue than the backward's stream, and which stream objects share a queue depends on
// how many streams the process (RCCL, torch) has made before — so it is measured: one single-workgroup kernel that spins
// for `us` microseconds of the constant-rate wall clock on each stream, forked and joined by events; together they take
// `us` when the streams overlap and 2 x `us` when they do not.
namespace {
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

extern "C" int evk_streams_overlap(void* a, void* b, int32_t us, float* elapsed_us) {
  int dev = 0, khz = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr, ef = nullptr, ej = nullptr;
  hipStream_t sa = static_cast<hipStream_t>(a), sb = static_cast<hipStream_t>(b);
  hipError_t err = hipGetDevice(&dev);
  if (err == hipSuccess) err = hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
  if (err == hipSuccess && khz <= 0) khz = 100000;
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  if (err == hipSuccess) err = hipEventCreateWithFlags(&ef, hipEventDisableTiming);
  if (err == hipSuccess) err = hipEventCreateWithFlags(&ej, hipEventDisableTiming);
  float ms = 0.f;
  if (err == hipSuccess) {
    const long long ticks = static_cast<long long>(khz) * us / 1000;
    for (int rep = 0; rep < 2 && err == hipSuccess; ++rep) {      // (the first round pays the kernel's load)
      err = hipEventRecord(e0, sa);
      if (err == hipSuccess) err = hipEventRecord(ef, sa);
      if (err == hipSuccess) err = hipStreamWaitEvent(sb, ef, 0);
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, sa, ticks);
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, sb, ticks);
      if (err == hipSuccess) err = hipEventRecord(ej, sb);
      if (err == hipSuccess) err = hipStreamWaitEvent(sa, ej, 0);
      if (err == hipSuccess) err = hipEventRecord(e1, sa);
      if (err == hipSuccess) err = hipEventSynchronize(e1);
      if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    }
  }
  for (hipEvent_t e : {e0, e1, ef, ej})
    if (e) (void)hipEventDestroy(e);
  if (err != hipSuccess) {
    evk::set_error("evk_streams_overlap: %s", hipGetErrorString(err));
    return -1;
  }
  if (elapsed_us) *elapsed_us = ms * 1e3f;
  return ms * 1e3f < 1.6f * us ? 1 : 0;
}
