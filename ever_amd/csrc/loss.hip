// Pixel-wise losses with ignore_index on NHWC logits / int64 labels (gfx950), forward statistics and
// gradients.  Replaces the masked_select compaction + reductions of reference
// ever/module/loss.py:10-17 (_masked_ignore), :26-37 (select), :40-75 (dice), :207-219 (LS-CE),
// :229-235 (BCE) and F.cross_entropy(ignore_index=...) in user model code.
// HBM-bound masked reductions: fp64 accumulation, per-workgroup partials then a single-workgroup
// finalisation (fixed order => reproducible).  stats layout: [K finals][kLossBlocks x K partials].
#include "common.hpp"

namespace evk {

constexpr int kLossBlocks = 2048;   // 256 left the 4-million-pixel losses on a quarter of the chip (62 us for 50 MB)
constexpr int kMaxClasses = 64;

static inline int loss_grid(int64_t npix) {
  int64_t b = (npix + 2047) / 2048;   // >= 8 pixels per thread
  return (int)(b > kLossBlocks ? kLossBlocks : (b < 1 ? 1 : b));
}

// block reduction of K doubles held one-per-thread-per-k via LDS; thread 0 gets the result
template <int K>
__device__ __forceinline__ void block_reduce_store(double (&v)[K], double* out) {
  __shared__ double red[K][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double s = wave_sum_d(v[k]);
    if (lane == 0) red[k][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
  }
  __syncthreads();
}

// stats[k] = sum over the nblk per-workgroup partials, one workgroup per k: every thread sums a fixed strided subset,
// the 256 sub-sums are folded through LDS in index order (reproducible; was one thread per k walking all partials)
__global__ __launch_bounds__(256) void finalize_partials_kernel(double* stats, int K, int nblk) {
  __shared__ double red[256];
  const int k = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += stats[K + (size_t)b * K + k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) stats[k] = red[0];
}

// ---------------------------------------------------------------- BCE with logits -----------------
__device__ __forceinline__ float bce_term(float x, float y, float pw) {
  // aten binary_cross_entropy_with_logits: (1-y)*x + (1 + (pos_weight-1)*y) * (log1p(exp(-|x|)) + max(-x,0))
  return (1.f - y) * x + (1.f + (pw - 1.f) * y) * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f));
}
__global__ __launch_bounds__(256) void bce_partial_kernel(const float* __restrict__ logits,
                                                          const int64_t* __restrict__ labels, int64_t npix,
                                                          int64_t ignore, float eps, float pw,
                                                          double* __restrict__ stats) {
  double v[2] = {0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t t = labels[i];
    if (t != ignore) {
      const float y = (t == 0) ? eps : (float)t - eps;  // label smoothing (eps = 0: plain BCE)
      v[0] += (double)bce_term(logits[i], y, pw);
      v[1] += 1.0;
    }
  }
  block_reduce_store<2>(v, stats + 2 + (size_t)blockIdx.x * 2);
}
__global__ void mean_loss_kernel(const double* stats, float* loss) {
  if (threadIdx.x == 0) *loss = (float)(stats[0] / stats[1]);  // 0/0 -> NaN, as mean of an empty tensor
}
__global__ void sum_loss_kernel(const double* stats, float* loss) {
  if (threadIdx.x == 0) *loss = (float)stats[0];               // reduction='sum': 0 for an empty selection, as aten
}
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ logits,
                                                      const int64_t* __restrict__ labels, int64_t npix, int64_t ignore,
                                                      float eps, float pw, int sum_reduction,
                                                      const double* __restrict__ stats,
                                                      const float* __restrict__ grad_scale, float* __restrict__ dlogits,
                                                      int accumulate) {
  const float k = (grad_scale ? *grad_scale : 1.f) / (sum_reduction ? 1.f : (float)stats[1]);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t t = labels[i];
    float g = 0.f;
    if (t != ignore) {
      const float p = 1.f / (1.f + expf(-logits[i]));
      const float y = (t == 0) ? eps : (float)t - eps;
      g = ((1.f - y) - (1.f + (pw - 1.f) * y) * (1.f - p)) * k;   // pw = 1: p - y
    }
    dlogits[i] = accumulate ? dlogits[i] + g : g;
  }
}

// ---------------------------------------------------------------- dice -----------------------------
// stats finals: inter[c] (k = c), z[c] (k = C + c)
template <int CT>  // CT == 1: sigmoid path;  CT == 0: softmax path with runtime C
__global__ __launch_bounds__(256) void dice_partial_kernel(const float* __restrict__ logits,
                                                           const int64_t* __restrict__ labels, int64_t npix, int C,
                                                           int64_t ignore, double* __restrict__ stats) {
  extern __shared__ double dacc[];  // [256][2C] only for the softmax path
  if (CT == 1) {
    double v[2] = {0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
      const int64_t t = labels[i];
      if (t != ignore) {
        const float p = 1.f / (1.f + expf(-logits[i]));
        const float y = (float)t;
        v[0] += (double)(p * y);
        v[1] += (double)p + (double)y;
      }
    }
    block_reduce_store<2>(v, stats + 2 + (size_t)blockIdx.x * 2);
  } else {
    const int K = 2 * C;
    double* mine = dacc + (size_t)threadIdx.x * K;
    for (int k = 0; k < K; ++k) mine[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
      const int64_t t = labels[i];
      if (t == ignore) continue;
      const float* x = logits + i * C;
      float m = x[0];
      for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(x[c] - m);
      const float lse = m + logf(se);
      for (int c = 0; c < C; ++c) {
        const float p = expf(x[c] - lse);  // reference: log_softmax(dim=1).exp()
        const float y = (t == c) ? 1.f : 0.f;
        mine[c] += (double)(p * y);
        mine[C + c] += (double)p + (double)y;
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) {
      double s = 0.0;
      for (int t = 0; t < 256; ++t) s += dacc[(size_t)t * K + k];
      stats[K + (size_t)blockIdx.x * K + k] = s;
    }
  }
}
__global__ void dice_finish_kernel(const double* stats, int C, float smooth, int ignore_channel, float* loss) {
  if (threadIdx.x != 0) return;
  double acc = 0.0;
  int nc = 0;
  for (int c = 0; c < C; ++c) {
    if (C > 1 && c == ignore_channel) continue;
    acc += (2.0 * stats[c] + (double)smooth) / (stats[C + c] + (double)smooth);
    ++nc;
  }
  *loss = (float)(1.0 - acc / (double)nc);
}
// d loss / d p_c = a_c * y_c + b_c,  a_c = -2/(nc (z_c+s)),  b_c = (2 I_c + s)/(nc (z_c+s)^2)
template <int CT>
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ logits,
                                                       const int64_t* __restrict__ labels, int64_t npix, int C,
                                                       int64_t ignore, const double* __restrict__ stats, float smooth,
                                                       int ignore_channel, const float* __restrict__ grad_scale,
                                                       float* __restrict__ dlogits, int accumulate) {
  __shared__ float sa[kMaxClasses], sb[kMaxClasses];
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    int nc = 0;
    for (int k = 0; k < C; ++k) nc += (C > 1 && k == ignore_channel) ? 0 : 1;
    const double zs = stats[C + c] + (double)smooth;
    const double gs = grad_scale ? (double)*grad_scale : 1.0;
    const bool off = (C > 1 && c == ignore_channel);
    sa[c] = off ? 0.f : (float)(-2.0 / ((double)nc * zs) * gs);
    sb[c] = off ? 0.f : (float)((2.0 * stats[c] + (double)smooth) / ((double)nc * zs * zs) * gs);
  }
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t t = labels[i];
    if (CT == 1) {
      float g = 0.f;
      if (t != ignore) {
        const float p = 1.f / (1.f + expf(-logits[i]));
        g = (sa[0] * (float)t + sb[0]) * p * (1.f - p);
      }
      dlogits[i] = accumulate ? dlogits[i] + g : g;
    } else {
      const float* x = logits + i * C;
      float* d = dlogits + i * C;
      if (t == ignore) {
        if (!accumulate)
          for (int c = 0; c < C; ++c) d[c] = 0.f;
        continue;
      }
      float m = x[0];
      for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += expf(x[c] - m);
      const float lse = m + logf(se);
      float dotgp = 0.f;
      for (int c = 0; c < C; ++c) {
        const float p = expf(x[c] - lse);
        const float g = sa[c] * ((t == c) ? 1.f : 0.f) + sb[c];
        dotgp += g * p;
      }
      for (int c = 0; c < C; ++c) {
        const float p = expf(x[c] - lse);
        const float g = sa[c] * ((t == c) ? 1.f : 0.f) + sb[c];
        const float v = p * (g - dotgp);
        d[c] = accumulate ? d[c] + v : v;
      }
    }
  }
}

// ---------------------------------------------------------------- cross entropy -------------------
// stats finals: [0] sum nll, [1] count, [2] sum_p (-sum_c logp_c)
__global__ __launch_bounds__(256) void ce_partial_kernel(const float* __restrict__ logits,
                                                         const int64_t* __restrict__ labels, int64_t npix, int C,
                                                         int64_t ignore, double* __restrict__ stats) {
  double v[3] = {0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t t = labels[i];
    if (t == ignore) continue;
    const float* x = logits + i * C;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float se = 0.f, sx = 0.f;
    for (int c = 0; c < C; ++c) {
      se += expf(x[c] - m);
      sx += x[c];
    }
    const float lse = m + logf(se);
    const float xt = (t >= 0 && t < C) ? x[t] : 0.f;
    v[0] += (double)(lse - xt);
    v[1] += 1.0;
    v[2] += (double)((float)C * lse - sx);
  }
  block_reduce_store<3>(v, stats + 3 + (size_t)blockIdx.x * 3);
}
__global__ void ce_finish_kernel(const double* stats, int C, float eps, float* loss) {
  if (threadIdx.x != 0) return;
  const double nll = stats[0] / stats[1];
  const double sm = stats[2] / stats[1];
  *loss = (float)((1.0 - (double)eps) * nll + (eps != 0.f ? (double)eps / (double)C * sm : 0.0));
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits,
                                                     const int64_t* __restrict__ labels, int64_t npix, int C,
                                                     int64_t ignore, float eps, const double* __restrict__ stats,
                                                     const float* __restrict__ grad_scale, float* __restrict__ dlogits,
                                                     int accumulate) {
  const float k = (grad_scale ? *grad_scale : 1.f) / (float)stats[1];
  const float k1 = (1.f - eps) * k, k2 = eps / (float)C * k;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t t = labels[i];
    const float* x = logits + i * C;
    float* d = dlogits + i * C;
    if (t == ignore) {
      if (!accumulate)
        for (int c = 0; c < C; ++c) d[c] = 0.f;
      continue;
    }
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(x[c] - m);
    const float lse = m + logf(se);
    for (int c = 0; c < C; ++c) {
      const float p = expf(x[c] - lse);
      const float v = k1 * (p - ((t == c) ? 1.f : 0.f)) + k2 * ((float)C * p - 1.f);
      d[c] = accumulate ? d[c] + v : v;
    }
  }
}

// ---------------------------------------------------------------- soft cross entropy --------------
// reference loss.py:238-242: -(target * log_softmax(input, 1)).mean(dim=(0,2,3)).sum()
//   = -sum_{pixels, c} t_c * logp_c / npix.   stats final: [0] = sum_{p,c} t_c * (lse - x_c)
__global__ __launch_bounds__(256) void soft_ce_partial_kernel(const float* __restrict__ logits,
                                                              const float* __restrict__ target, int64_t npix, int C,
                                                              double* __restrict__ stats) {
  double v[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const float* x = logits + i * C;
    const float* t = target + i * C;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(x[c] - m);
    const float lse = m + logf(se);
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc += t[c] * (lse - x[c]);
    v[0] += (double)acc;
  }
  block_reduce_store<1>(v, stats + 1 + (size_t)blockIdx.x);
}
__global__ void soft_ce_finish_kernel(const double* stats, double inv_npix, float* loss) {
  if (threadIdx.x == 0) *loss = (float)(stats[0] * inv_npix);
}
__global__ __launch_bounds__(256) void soft_ce_bwd_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ target, int64_t npix, int C,
                                                          float inv_npix, const float* __restrict__ grad_scale,
                                                          float* __restrict__ dlogits) {
  const float k = (grad_scale ? *grad_scale : 1.f) * inv_npix;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const float* x = logits + i * C;
    const float* t = target + i * C;
    float* d = dlogits + i * C;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float se = 0.f, ts = 0.f;
    for (int c = 0; c < C; ++c) {
      se += expf(x[c] - m);
      ts += t[c];
    }
    const float lse = m + logf(se);
    for (int c = 0; c < C; ++c) d[c] = k * (expf(x[c] - lse) * ts - t[c]);
  }
}

}  // namespace evk

using namespace evk;

extern "C" int evk_soft_ce_fwd(const float* logits, const float* target, int64_t npix, int32_t C, float* loss,
                               double* stats, void* stream) {
  EVK_REQUIRE(logits && target && loss && stats && npix > 0 && C >= 1, EVK_E_INVALID, "soft_ce_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nb = loss_grid(npix);
  hipLaunchKernelGGL(soft_ce_partial_kernel, dim3(nb), dim3(256), 0, st, logits, target, npix, C, stats);
  hipLaunchKernelGGL(finalize_partials_kernel, dim3(1), dim3(256), 0, st, stats, 1, nb);
  hipLaunchKernelGGL(soft_ce_finish_kernel, dim3(1), dim3(64), 0, st, (const double*)stats, 1.0 / (double)npix, loss);
  return check_launch("soft_ce_fwd");
}
extern "C" int evk_soft_ce_bwd(const float* logits, const float* target, int64_t npix, int32_t C,
                               const float* grad_scale, float* dlogits, void* stream) {
  EVK_REQUIRE(logits && target && dlogits && npix > 0 && C >= 1, EVK_E_INVALID, "soft_ce_bwd: bad argument");
  const int nb = (int)((npix + 255) / 256 > 4096 ? 4096 : (npix + 255) / 256);
  hipLaunchKernelGGL(soft_ce_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, logits, target, npix, C,
                     (float)(1.0 / (double)npix), grad_scale, dlogits);
  return check_launch("soft_ce_bwd");
}

extern "C" int64_t evk_loss_stats_doubles(int32_t K) { return (int64_t)K * (1 + kLossBlocks); }

extern "C" int evk_bce_fwd_ex(const float* logits, const int64_t* labels, int64_t npix, int64_t ignore_index,
                              float label_smoothing, float pos_weight, int32_t reduction, float* loss, double* stats,
                              void* stream) {
  EVK_REQUIRE(logits && labels && loss && stats && npix > 0, EVK_E_INVALID, "bce_fwd: bad argument");
  EVK_REQUIRE(reduction == 0 || reduction == 1, EVK_E_UNSUPPORTED, "bce_fwd: reduction must be 0 (mean) or 1 (sum)");
  hipStream_t st = (hipStream_t)stream;
  const int nb = loss_grid(npix);
  hipLaunchKernelGGL(bce_partial_kernel, dim3(nb), dim3(256), 0, st, logits, labels, npix, ignore_index,
                     label_smoothing, pos_weight, stats);
  hipLaunchKernelGGL(finalize_partials_kernel, dim3(2), dim3(256), 0, st, stats, 2, nb);
  if (reduction == 0) {
    hipLaunchKernelGGL(mean_loss_kernel, dim3(1), dim3(64), 0, st, (const double*)stats, loss);
  } else {
    hipLaunchKernelGGL(sum_loss_kernel, dim3(1), dim3(64), 0, st, (const double*)stats, loss);
  }
  return check_launch("bce_fwd");
}
extern "C" int evk_bce_fwd(const float* logits, const int64_t* labels, int64_t npix, int64_t ignore_index,
                           float label_smoothing, float* loss, double* stats, void* stream) {
  return evk_bce_fwd_ex(logits, labels, npix, ignore_index, label_smoothing, 1.f, 0, loss, stats, stream);
}
extern "C" int evk_bce_bwd_ex(const float* logits, const int64_t* labels, int64_t npix, int64_t ignore_index,
                              float label_smoothing, float pos_weight, int32_t reduction, const double* stats,
                              const float* grad_scale, float* dlogits, int32_t accumulate, void* stream) {
  EVK_REQUIRE(logits && labels && stats && dlogits && npix > 0, EVK_E_INVALID, "bce_bwd: bad argument");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3((int)((npix + 255) / 256 > 4096 ? 4096 : (npix + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, logits, labels, npix, ignore_index, label_smoothing, pos_weight,
                     reduction == 1 ? 1 : 0, stats, grad_scale, dlogits, accumulate);
  return check_launch("bce_bwd");
}
extern "C" int evk_bce_bwd(const float* logits, const int64_t* labels, int64_t npix, int64_t ignore_index,
                           float label_smoothing, const double* stats, const float* grad_scale, float* dlogits,
                           int32_t accumulate, void* stream) {
  return evk_bce_bwd_ex(logits, labels, npix, ignore_index, label_smoothing, 1.f, 0, stats, grad_scale, dlogits,
                        accumulate, stream);
}

extern "C" int evk_dice_stats(const float* logits, const int64_t* labels, int64_t npix, int32_t C, int64_t ignore_index,
                              double* stats, void* stream) {
  EVK_REQUIRE(logits && labels && stats && npix > 0 && C >= 1 && C <= kMaxClasses, EVK_E_INVALID,
              "dice_stats: bad argument (1 <= C <= %d)", kMaxClasses);
  hipStream_t st = (hipStream_t)stream;
  const int nb = loss_grid(npix);
  if (C == 1) {
    hipLaunchKernelGGL(dice_partial_kernel<1>, dim3(nb), dim3(256), 0, st, logits, labels, npix, C, ignore_index, stats);
  } else {
    const size_t lds = (size_t)256 * 2 * C * sizeof(double);
    EVK_REQUIRE(lds <= 64 * 1024, EVK_E_UNSUPPORTED, "dice_stats: C=%d too large for the LDS accumulator", C);
    hipLaunchKernelGGL(dice_partial_kernel<0>, dim3(nb), dim3(256), lds, st, logits, labels, npix, C, ignore_index,
                       stats);
  }
  hipLaunchKernelGGL(finalize_partials_kernel, dim3(2 * C), dim3(256), 0, st, stats, 2 * C, nb);
  return check_launch("dice_stats");
}
extern "C" int evk_dice_finish(const double* stats, int32_t C, float smooth, int32_t ignore_channel, float* loss,
                               void* stream) {
  EVK_REQUIRE(stats && loss && C >= 1, EVK_E_INVALID, "dice_finish: bad argument");
  hipLaunchKernelGGL(dice_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats, C, smooth, ignore_channel,
                     loss);
  return check_launch("dice_finish");
}
extern "C" int evk_dice_bwd(const float* logits, const int64_t* labels, int64_t npix, int32_t C, int64_t ignore_index,
                            const double* stats, float smooth, int32_t ignore_channel, const float* grad_scale,
                            float* dlogits, int32_t accumulate, void* stream) {
  EVK_REQUIRE(logits && labels && stats && dlogits && npix > 0 && C >= 1 && C <= kMaxClasses, EVK_E_INVALID,
              "dice_bwd: bad argument");
  const int nb = (int)((npix + 255) / 256 > 4096 ? 4096 : (npix + 255) / 256);
  if (C == 1)
    hipLaunchKernelGGL(dice_bwd_kernel<1>, dim3(nb), dim3(256), 0, (hipStream_t)stream, logits, labels, npix, C,
                       ignore_index, stats, smooth, ignore_channel, grad_scale, dlogits, accumulate);
  else
    hipLaunchKernelGGL(dice_bwd_kernel<0>, dim3(nb), dim3(256), 0, (hipStream_t)stream, logits, labels, npix, C,
                       ignore_index, stats, smooth, ignore_channel, grad_scale, dlogits, accumulate);
  return check_launch("dice_bwd");
}

extern "C" int evk_ce_fwd(const float* logits, const int64_t* labels, int64_t npix, int32_t C, int64_t ignore_index,
                          float label_smoothing, float* loss, double* stats, void* stream) {
  EVK_REQUIRE(logits && labels && loss && stats && npix > 0 && C >= 1, EVK_E_INVALID, "ce_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nb = loss_grid(npix);
  hipLaunchKernelGGL(ce_partial_kernel, dim3(nb), dim3(256), 0, st, logits, labels, npix, C, ignore_index, stats);
  hipLaunchKernelGGL(finalize_partials_kernel, dim3(3), dim3(256), 0, st, stats, 3, nb);
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, st, (const double*)stats, C, label_smoothing, loss);
  return check_launch("ce_fwd");
}
extern "C" int evk_ce_bwd(const float* logits, const int64_t* labels, int64_t npix, int32_t C, int64_t ignore_index,
                          float label_smoothing, const double* stats, const float* grad_scale, float* dlogits,
                          int32_t accumulate, void* stream) {
  EVK_REQUIRE(logits && labels && stats && dlogits && npix > 0 && C >= 1, EVK_E_INVALID, "ce_bwd: bad argument");
  const int nb = (int)((npix + 255) / 256 > 4096 ? 4096 : (npix + 255) / 256);
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, logits, labels, npix, C, ignore_index,
                     label_smoothing, stats, grad_scale, dlogits, accumulate);
  return check_launch("ce_bwd");
}
