// Shared helpers for the gfx950 kernels of libever_hip.so.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ever_hip.h"

namespace evk {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return EVK_E_LAUNCH;
  }
  return EVK_OK;
}

#define EVK_REQUIRE(cond, code, ...)     \
  do {                                   \
    if (!(cond)) {                       \
      ::evk::set_error(__VA_ARGS__);     \
      return (code);                     \
    }                                    \
  } while (0)

constexpr int kWave = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Unsigned division by a runtime-invariant divisor (n < 2^31): q = (umulhi(n, mul) + n) >> shift.
struct FastDiv {
  uint32_t div, mul, shift;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.div = d;
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  f.shift = s;
  f.mul = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return (uint32_t)(((uint64_t)__umulhi(n, f.mul) + n) >> f.shift);
}

// XCD-aware tile order: consecutive tile ids (sharing A rows / weights) stay on one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace evk
