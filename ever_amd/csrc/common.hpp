// Shared helpers for the gfx950 kernels of libever_hip.so.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
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

// "Once per DEVICE" for hipFuncSetAttribute: a function attribute belongs to the device's copy of the code object, so a
// process that drives several devices has to set it on each (one flag per process left the second device's kernels at the
// 64 KB default: launch failure with a 98-160 KB ring).
struct PerDeviceOnce {
  std::atomic<uint64_t> mask{0};
  bool first() {
    int d = 0;
    (void)hipGetDevice(&d);
    const uint64_t b = 1ull << (d & 63);
    return !(mask.fetch_or(b) & b);
  }
};

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

// ---- ReLU bits of an fp32 tensor: one bit per element ("the activation's output was positive"), written by the
// BatchNorm + add + ReLU pass of a residual block and read wherever its backward needs the mask — 1/32 of the bytes of the
// output tensor it replaces there (csrc/bn.hip).  Layout: the tensor is cut into 16-byte elements j (4 floats), 64 of
// them make a chunk of 8 words: word (j >> 6) * 8 + ((j >> 5) & 1) * 4 + e holds, at bit (j & 31), the bit of float e of
// element j — so a writer that owns element j = wave-chunk * 64 + lane gets its four words from four wave ballots, and a
// reader with ANY thread-to-element mapping fetches one aligned 16-byte group and shifts.
__host__ __device__ inline size_t relu_bits_words(size_t n4) { return ((n4 + 63) / 64) * 8; }
typedef uint32_t relu_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 relu_bits_mask(f32x4 g, const uint32_t* __restrict__ bits, size_t j) {
  const relu_u32x4 w = *reinterpret_cast<const relu_u32x4*>(bits + (j >> 6) * 8 + ((j >> 5) & 1) * 4);
  const uint32_t b = (uint32_t)(j & 31);
  f32x4 o;
  o.x = ((w.x >> b) & 1u) ? g.x : 0.f;
  o.y = ((w.y >> b) & 1u) ? g.y : 0.f;
  o.z = ((w.z >> b) & 1u) ? g.z : 0.f;
  o.w = ((w.w >> b) & 1u) ? g.w : 0.f;
  return o;
}
// writer: every lane of the wave calls this with ITS element j = chunk * 64 + lane (inactive elements pass pos = 0 bits)
__device__ __forceinline__ void relu_bits_store(uint32_t* __restrict__ bits, size_t chunk, bool p0, bool p1, bool p2, bool p3) {
  const uint64_t b0 = __ballot(p0), b1 = __ballot(p1), b2 = __ballot(p2), b3 = __ballot(p3);
  const int lane = threadIdx.x & 63;
  if (lane == 0)
    *reinterpret_cast<relu_u32x4*>(bits + chunk * 8) = relu_u32x4{(uint32_t)b0, (uint32_t)b1, (uint32_t)b2, (uint32_t)b3};
  if (lane == 32)
    *reinterpret_cast<relu_u32x4*>(bits + chunk * 8 + 4) =
        relu_u32x4{(uint32_t)(b0 >> 32), (uint32_t)(b1 >> 32), (uint32_t)(b2 >> 32), (uint32_t)(b3 >> 32)};
}

}  // namespace evk
