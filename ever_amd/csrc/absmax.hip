// max |x| of a tensor as the bit image of a non-negative float (unsigned order = float order), left in device memory:
// the per-tensor operand scale of the f16x2 convolution arithmetic (x3_common.hpp: op_scale).  No reference call site —
// operand preparation of this engine's arithmetic, like the weight planes.  HBM-bound: one read pass.
//   evk_absmax        one tensor -> an ACTIVATION scale buffer of evk_absmax_words() words (slot 0 = the maximum, the other
//                     slots zero: x3_common.hpp act_absmax); partial maxima per workgroup, folded by the LAST-ARRIVING
//                     workgroup (no second launch, no pre-zeroed output; the caller's workspace holds the partials and a ticket counter that must be
//                     zero before the first call and is left zero by every call — calls on ONE stream only)
//   evk_absmax_multi  n tensors in one launch (the convolution weights, once per optimiser step): atomic max into a
//                     zeroed output array
#include "x3_common.hpp"

namespace evk {

constexpr int kAbsBlocks = 512;

__device__ __forceinline__ uint32_t abs_bits(float v) { return __builtin_bit_cast(uint32_t, v) & 0x7fffffffu; }
__device__ __forceinline__ uint32_t max4(uint32_t m, const f32x4 v) {
  m = max(m, abs_bits(v.x)); m = max(m, abs_bits(v.y));
  m = max(m, abs_bits(v.z)); m = max(m, abs_bits(v.w));
  return m;
}
__device__ __forceinline__ uint32_t block_max(uint32_t m) {
  __shared__ uint32_t red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  return max(max(red[0], red[1]), max(red[2], red[3]));
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n, size_t span4,
                                                     uint32_t* __restrict__ out, uint32_t* __restrict__ ws) {
  const size_t n4 = n >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const size_t b0 = (size_t)blockIdx.x * span4, b1 = min(n4, b0 + span4);
  uint32_t m = 0;
  size_t i = b0 + threadIdx.x;
  for (; i + 1792 < b1; i += 2048) {   // eight independent 16-byte loads in flight per lane
    const f32x4 v0 = x4[i], v1 = x4[i + 256], v2 = x4[i + 512], v3 = x4[i + 768];
    const f32x4 v4 = x4[i + 1024], v5 = x4[i + 1280], v6 = x4[i + 1536], v7 = x4[i + 1792];
    m = max4(max4(max4(max4(m, v0), v1), v2), v3);
    m = max4(max4(max4(max4(m, v4), v5), v6), v7);
  }
  for (; i < b1; i += 256) m = max4(m, x4[i]);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, abs_bits(x[(n4 << 2) + threadIdx.x]));
  m = block_max(m);
  __shared__ bool last;
  if (threadIdx.x == 0) {
    ws[blockIdx.x] = m;
    __threadfence();
    last = atomicAdd(&ws[kAbsBlocks], 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  uint32_t t = 0;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) t = max(t, __builtin_nontemporal_load(ws + b));
  __syncthreads();   // block_max's shared array is reused
  t = block_max(t);
  if (threadIdx.x < kAmaxSlots) out[threadIdx.x * kAmaxStride] = threadIdx.x == 0 ? t : 0u;   // slot 0; the others empty
  if (threadIdx.x == 0) ws[kAbsBlocks] = 0;   // ready for the next call on this stream
}

__global__ __launch_bounds__(256) void absmax_multi_kernel(const float* const* __restrict__ ptrs,
                                                           const int64_t* __restrict__ sizes, uint32_t* __restrict__ out) {
  const int t = blockIdx.y;
  const float* x = ptrs[t];
  const size_t n = (size_t)sizes[t];
  uint32_t m = 0;
  if ((((uintptr_t)x) & 15) == 0) {
    const size_t n4 = n >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) m = max4(m, x4[i]);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, abs_bits(x[(n4 << 2) + threadIdx.x]));
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = max(m, abs_bits(x[i]));
  }
  m = block_max(m);
  if (threadIdx.x == 0 && m) atomicMax(out + t, m);
}

// x -> packed words of x / s (x3_common.hpp: pack_hl), s from the tensor's scale buffer; and back (h + l) * s
__global__ __launch_bounds__(256) void pack_f16x2_kernel(const float* __restrict__ x, size_t n4,
                                                         const uint32_t* __restrict__ slots, uint32_t* __restrict__ out) {
  const float inv = op_scale(act_absmax(slots)).inv;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  reinterpret_cast<u32x4*>(out)[i] = pack_hl4(v, inv);
}
__global__ __launch_bounds__(256) void unpack_f16x2_kernel(const uint32_t* __restrict__ in, size_t n4,
                                                           const uint32_t* __restrict__ slots, float* __restrict__ out) {
  const float s = op_scale(act_absmax(slots)).s;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const u32x4 w = reinterpret_cast<const u32x4*>(in)[i];
  const f32x4 v = {unpack_hl(w.x) * s, unpack_hl(w.y) * s, unpack_hl(w.z) * s, unpack_hl(w.w) * s};
  reinterpret_cast<f32x4*>(out)[i] = v;
}

}  // namespace evk

using namespace evk;

extern "C" int evk_pack_f16x2(const float* x, int64_t n, const uint32_t* x_absmax, uint32_t* out, void* stream) {
  EVK_REQUIRE(x && x_absmax && out && n > 0 && n % 4 == 0, EVK_E_INVALID, "pack_f16x2: bad argument (n %% 4 == 0)");
  const size_t n4 = (size_t)n >> 2;
  hipLaunchKernelGGL(pack_f16x2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n4,
                     x_absmax, out);
  return check_launch("pack_f16x2");
}
extern "C" int evk_unpack_f16x2(const uint32_t* packed, int64_t n, const uint32_t* x_absmax, float* out, void* stream) {
  EVK_REQUIRE(packed && x_absmax && out && n > 0 && n % 4 == 0, EVK_E_INVALID, "unpack_f16x2: bad argument (n %% 4 == 0)");
  const size_t n4 = (size_t)n >> 2;
  hipLaunchKernelGGL(unpack_f16x2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, packed,
                     n4, x_absmax, out);
  return check_launch("unpack_f16x2");
}

extern "C" size_t evk_absmax_words(void) { return (size_t)kAmaxWords; }
extern "C" size_t evk_absmax_workspace_bytes(void) { return (size_t)(kAbsBlocks + 1) * sizeof(uint32_t); }

extern "C" int evk_absmax(const float* x, int64_t n, uint32_t* out_bits, void* workspace, void* stream) {
  EVK_REQUIRE(x && out_bits && workspace && n > 0, EVK_E_INVALID, "absmax: bad argument");
  if (((uintptr_t)x & 15) != 0) {   // (every tensor the path produces is 16-byte aligned; odd views are refused)
    set_error("absmax: tensor must be 16-byte aligned");
    return EVK_E_INVALID;
  }
  const size_t n4 = (size_t)n >> 2;
  // every workgroup ends with one ticket atomic on ONE word (~20 ns each, serialised): few, long spans
  size_t blocks = (n4 + 4095) / 4096;            // >= 16 float4 per thread
  if (blocks > (size_t)kAbsBlocks) blocks = kAbsBlocks;
  if (blocks < 1) blocks = 1;
  size_t span4 = (n4 + blocks - 1) / blocks;
  span4 = (span4 + 255) & ~(size_t)255;
  blocks = n4 ? (n4 + span4 - 1) / span4 : 1;
  if (span4 == 0) span4 = 256;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, span4, out_bits,
                     (uint32_t*)workspace);
  return check_launch("absmax");
}

extern "C" int evk_absmax_multi(const float* const* ptrs_dev, const int64_t* sizes_dev, int32_t n_tensors,
                                uint32_t* out_bits, void* stream) {
  EVK_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (ptrs_dev && sizes_dev && out_bits)), EVK_E_INVALID,
              "absmax_multi: bad argument");
  if (n_tensors == 0) return EVK_OK;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out_bits, 0, (size_t)n_tensors * sizeof(uint32_t), st) != hipSuccess) {
    set_error("absmax_multi: memset failed");
    return EVK_E_LAUNCH;
  }
  hipLaunchKernelGGL(absmax_multi_kernel, dim3(32, (unsigned)n_tensors), dim3(256), 0, st, ptrs_dev, sizes_dev, out_bits);
  return check_launch("absmax_multi");
}
