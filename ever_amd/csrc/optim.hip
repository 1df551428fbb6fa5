// Multi-tensor SGD step and global gradient-norm clipping (gfx950), one launch over a device-side
// table of tensors.  Replaces the per-tensor loops of torch.optim.SGD (reference
// ever/opt/optimizer.py:7) and clip_grad_norm_ (ever/interface/module.py:96-108).  HBM-bound.
#include "common.hpp"

namespace evk {

// workgroups per tensor: the 20 largest of FarSeg-R50's 238 parameter tensors hold most of the bytes, and 32
// workgroups each left the chip under-filled (sgd_multi at 1.9 TB/s); small tensors' surplus workgroups exit at once
constexpr int kOptBlocksPerTensor = 128;

__global__ __launch_bounds__(256) void sqnorm_multi_kernel(const float* const* __restrict__ grads,
                                                           const int64_t* __restrict__ sizes,
                                                           double* __restrict__ partial) {
  __shared__ double red[4];
  const int t = blockIdx.y;
  const float* g = grads[t];
  const int64_t n = sizes[t];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = g[i];
    s += (double)v * (double)v;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)t * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// total_norm = sqrt(sum partial); coef = min(1, max_norm / (total_norm + 1e-6))   (torch semantics)
__global__ void clip_coef_kernel(const double* __restrict__ partial, int n, float max_norm, float* total_norm,
                                 float* coef) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tn = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
    *total_norm = tn;
    const float c = max_norm / (tn + 1e-6f);
    *coef = c < 1.f ? c : 1.f;
  }
}

__global__ __launch_bounds__(256) void sgd_multi_kernel(float* const* __restrict__ params,
                                                        const float* const* __restrict__ grads,
                                                        float* const* __restrict__ bufs,
                                                        const int64_t* __restrict__ sizes, float lr, float momentum,
                                                        float dampening, float wd, int nesterov, int first_step,
                                                        const float* __restrict__ clip_coef,
                                                        const float* __restrict__ lr_dev) {
  if (lr_dev) lr = *lr_dev;   // a captured launch (hipGraph replay) must not bake the schedule's value in
  const int t = blockIdx.y;
  float* p = params[t];
  const float* g = grads[t];
  float* b = bufs ? bufs[t] : nullptr;
  const int64_t n = sizes[t];
  const float cc = clip_coef ? *clip_coef : 1.f;
  // explicit fused multiply-adds: the 16-byte and the scalar loop must round identically (a DDP replica whose
  // gradients are bucket views takes the scalar loop where an unwrapped model takes the vector one)
  auto upd = [&](float gi, float w, float& m) {
    float d = gi * cc;
    if (wd != 0.f) d = __fmaf_rn(wd, w, d);
    if (momentum != 0.f) {
      m = first_step ? d : __fmaf_rn(momentum, m, (1.f - dampening) * d);
      d = nesterov ? __fmaf_rn(momentum, m, d) : m;
    }
    return __fmaf_rn(-lr, d, w);
  };
  // 16-byte main part when the three tensors allow it (gradients may be views into a DDP bucket at any 4-byte offset)
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)b) & 15) == 0;
  const int64_t n4 = vec ? n >> 2 : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 w = reinterpret_cast<f32x4*>(p)[i];
    f32x4 m = {0.f, 0.f, 0.f, 0.f};
    if (momentum != 0.f && !first_step) m = reinterpret_cast<f32x4*>(b)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float me = m[e];
      w[e] = upd(gv[e], w[e], me);
      m[e] = me;
    }
    if (momentum != 0.f) reinterpret_cast<f32x4*>(b)[i] = m;
    reinterpret_cast<f32x4*>(p)[i] = w;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float m = (momentum != 0.f && !first_step) ? b[i] : 0.f;
    const float w = upd(g[i], p[i], m);
    if (momentum != 0.f) b[i] = m;
    p[i] = w;
  }
}

// torch.optim.Adam / AdamW (single-tensor form, no amsgrad): decoupled = 1 is AdamW (p *= 1 - lr*wd), 0 is Adam
// (g += wd*p).  bc1 = 1 - beta1^step, bc2s = sqrt(1 - beta2^step) are computed by the caller from the step count.
__global__ __launch_bounds__(256) void adam_multi_kernel(float* const* __restrict__ params,
                                                         const float* const* __restrict__ grads,
                                                         float* const* __restrict__ exp_avg,
                                                         float* const* __restrict__ exp_avg_sq,
                                                         const int64_t* __restrict__ sizes, float lr, float beta1,
                                                         float beta2, float eps, float wd, int decoupled, float bc1,
                                                         float bc2s, const float* __restrict__ clip_coef) {
  const int t = blockIdx.y;
  float* p = params[t];
  const float* g = grads[t];
  float* m = exp_avg[t];
  float* v = exp_avg_sq[t];
  const int64_t n = sizes[t];
  const float cc = clip_coef ? *clip_coef : 1.f;
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float w = p[i];
    float gi = g[i] * cc;
    if (wd != 0.f) {
      if (decoupled) w *= 1.f - lr * wd;
      else gi = __fmaf_rn(wd, w, gi);
    }
    const float mi = __fmaf_rn(gi - m[i], 1.f - beta1, m[i]);            // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = __fmaf_rn(gi * gi, 1.f - beta2, v[i] * beta2);      // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(vi) / bc2s + eps;
    m[i] = mi;
    v[i] = vi;
    p[i] = w - step_size * (mi / denom);
  }
}

}  // namespace evk
using namespace evk;

extern "C" int evk_adam_multi(float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const int64_t* sizes, int32_t ntensors, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int32_t decoupled, float bias_correction1,
                              float sqrt_bias_correction2, const float* clip_coef, void* stream) {
  EVK_REQUIRE(params && grads && exp_avg && exp_avg_sq && sizes && ntensors > 0, EVK_E_INVALID, "adam_multi: bad argument");
  EVK_REQUIRE(bias_correction1 > 0.f && sqrt_bias_correction2 > 0.f, EVK_E_INVALID, "adam_multi: bad bias correction");
  hipLaunchKernelGGL(adam_multi_kernel, dim3(kOptBlocksPerTensor, ntensors), dim3(256), 0, (hipStream_t)stream, params,
                     grads, exp_avg, exp_avg_sq, sizes, lr, beta1, beta2, eps, weight_decay, decoupled, bias_correction1,
                     sqrt_bias_correction2, clip_coef);
  return check_launch("adam_multi");
}

extern "C" int32_t evk_opt_blocks_per_tensor(void) { return kOptBlocksPerTensor; }

extern "C" int evk_sqnorm_multi(const float* const* grads, const int64_t* sizes, int32_t ntensors, double* partial,
                                float max_norm, float* total_norm, float* clip_coef, void* stream) {
  EVK_REQUIRE(grads && sizes && partial && total_norm && clip_coef && ntensors > 0, EVK_E_INVALID,
              "sqnorm_multi: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_multi_kernel, dim3(kOptBlocksPerTensor, ntensors), dim3(256), 0, st, grads, sizes, partial);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, st, (const double*)partial,
                     kOptBlocksPerTensor * ntensors, max_norm, total_norm, clip_coef);
  return check_launch("sqnorm_multi");
}

extern "C" int evk_sgd_multi_lr(float* const* params, const float* const* grads, float* const* momentum_bufs,
                                const int64_t* sizes, int32_t ntensors, float lr, const float* lr_dev, float momentum,
                                float dampening, float weight_decay, int32_t nesterov, int32_t first_step,
                                const float* clip_coef, void* stream) {
  EVK_REQUIRE(params && grads && sizes && ntensors > 0, EVK_E_INVALID, "sgd_multi: bad argument");
  EVK_REQUIRE(momentum == 0.f || momentum_bufs, EVK_E_INVALID, "sgd_multi: momentum needs buffers");
  hipLaunchKernelGGL(sgd_multi_kernel, dim3(kOptBlocksPerTensor, ntensors), dim3(256), 0, (hipStream_t)stream, params,
                     grads, momentum_bufs, sizes, lr, momentum, dampening, weight_decay, nesterov, first_step,
                     clip_coef, lr_dev);
  return check_launch("sgd_multi");
}
extern "C" int evk_sgd_multi(float* const* params, const float* const* grads, float* const* momentum_bufs,
                             const int64_t* sizes, int32_t ntensors, float lr, float momentum, float dampening,
                             float weight_decay, int32_t nesterov, int32_t first_step, const float* clip_coef,
                             void* stream) {
  return evk_sgd_multi_lr(params, grads, momentum_bufs, sizes, ntensors, lr, nullptr, momentum, dampening, weight_decay,
                          nesterov, first_step, clip_coef, stream);
}

// dst[offsets[t] + i] = srcs[t][i] * scale  (srcs[t] == NULL: zeros) — one launch packs every gradient of a
// bucket into the flat all-reduce buffer, pre-divided by the world size (replaces the per-parameter
// copy + div kernels of torch DDP's reducer: 238 launches / 2.3 ms per step on FarSeg-R50).
namespace evk {
__global__ __launch_bounds__(256) void pack_multi_kernel(const float* const* __restrict__ srcs,
                                                         const int64_t* __restrict__ sizes,
                                                         const int64_t* __restrict__ offsets, float scale,
                                                         float* dst) {   // (no __restrict__: a source may BE its slot)
  const int t = blockIdx.y;
  const float* s = srcs[t];
  float* d = dst + offsets[t];
  const int64_t n = sizes[t];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    d[i] = s ? s[i] * scale : 0.f;
}
}  // namespace evk

extern "C" int evk_pack_multi(const float* const* srcs, const int64_t* sizes, const int64_t* offsets, int32_t ntensors,
                              float scale, float* dst, void* stream) {
  EVK_REQUIRE(srcs && sizes && offsets && dst && ntensors > 0, EVK_E_INVALID, "pack_multi: bad argument");
  hipLaunchKernelGGL(evk::pack_multi_kernel, dim3(evk::kOptBlocksPerTensor, ntensors), dim3(256), 0, (hipStream_t)stream,
                     srcs, sizes, offsets, scale, dst);
  return evk::check_launch("pack_multi");
}
