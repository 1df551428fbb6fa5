// SURVEY §8 "next" rows f2 / f3 on gfx950:
//   * class-probability statistics (tp, sum p, sum y per class) and their adjoint: the sufficient statistics
//     of tversky_loss_with_logits (reference ever/module/loss.py:78-143) and of any dice-like ratio loss;
//   * focal losses on float targets (loss.py:158-201);
//   * the confusion matrix of the evaluation loop (reference ever/metric/confusion_matrix.py:11-24, which
//     goes through the host and scipy.sparse) accumulated on the device, from predictions or straight
//     from logits (threshold 0 / argmax fused).
// HBM-bound streaming kernels, fp64 partial sums reduced in a fixed order (reproducible); the confusion
// matrix is integer (LDS-privatised histogram + integer atomics: exact in any order).
#include "common.hpp"

namespace evk {

constexpr int kNRBlocks = 256;
constexpr int kNRMaxC = 64;

static inline int nr_grid(int64_t n) {
  int64_t b = (n + 1023) / 1024;
  return (int)(b > kNRBlocks ? kNRBlocks : (b < 1 ? 1 : b));
}

__device__ __forceinline__ double block_sum(double v, double* red /*[4]*/) {
  const double s = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// stats layout: [3C finals: tp[C], sp[C], sy[C]] [nblk x 3C partials]
__global__ __launch_bounds__(256) void prob_stats_kernel(const float* __restrict__ logits,
                                                         const int64_t* __restrict__ labels, int64_t npix, int C,
                                                         int64_t ignore, double* __restrict__ stats) {
  __shared__ double red[4];
  double tp[kNRMaxC], sp[kNRMaxC], sy[kNRMaxC];
  for (int c = 0; c < C; ++c) tp[c] = sp[c] = sy[c] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t t = labels[i];
    if (t == ignore) continue;
    const float* z = logits + i * C;
    if (C == 1) {
      // p = exp(logsigmoid(z)) as the reference computes it (loss.py:121)
      const float p = expf(fminf(z[0], 0.f) - log1pf(expf(-fabsf(z[0]))));
      const double y = (double)t;
      tp[0] += (double)p * y;
      sp[0] += (double)p;
      sy[0] += y;
    } else {
      float m = z[0];
      for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += expf(z[c] - m);
      const float lse = m + logf(s);
      for (int c = 0; c < C; ++c) {
        const float p = expf(z[c] - lse);
        sp[c] += (double)p;
        if (t == c) {
          tp[c] += (double)p;
          sy[c] += 1.0;
        }
      }
    }
  }
  double* out = stats + 3 * C + (size_t)blockIdx.x * 3 * C;
  for (int c = 0; c < C; ++c) {
    const double a = block_sum(tp[c], red), b = block_sum(sp[c], red), d = block_sum(sy[c], red);
    if (threadIdx.x == 0) {
      out[c] = a;
      out[C + c] = b;
      out[2 * C + c] = d;
    }
  }
}
__global__ void nr_finalize_kernel(double* stats, int K, int nblk) {
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += stats[K + (size_t)b * K + k];
    stats[k] = s;
  }
}
// dlogits for L(tp, sp): per pixel G_c = gtp[c]*y_c + gsp[c];  sigmoid: dz = G p (1-p);
// softmax: dz_j = p_j (G_j - sum_k G_k p_k).  (sy does not depend on the logits.)
__global__ __launch_bounds__(256) void prob_stats_bwd_kernel(const float* __restrict__ logits,
                                                             const int64_t* __restrict__ labels, int64_t npix, int C,
                                                             int64_t ignore, const float* __restrict__ gtp,
                                                             const float* __restrict__ gsp, float* __restrict__ dlogits,
                                                             int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t t = labels[i];
    const float* z = logits + i * C;
    float* d = dlogits + i * C;
    if (t == ignore) {
      if (!accumulate)
        for (int c = 0; c < C; ++c) d[c] = 0.f;
      continue;
    }
    if (C == 1) {
      const float p = expf(fminf(z[0], 0.f) - log1pf(expf(-fabsf(z[0]))));
      const float g = (gtp[0] * (float)t + gsp[0]) * p * (1.f - p);
      d[0] = accumulate ? d[0] + g : g;
    } else {
      float m = z[0];
      for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += expf(z[c] - m);
      const float lse = m + logf(s);
      float dot = 0.f;
      for (int c = 0; c < C; ++c) dot += (gsp[c] + (t == c ? gtp[c] : 0.f)) * expf(z[c] - lse);
      for (int c = 0; c < C; ++c) {
        const float p = expf(z[c] - lse);
        const float g = p * ((gsp[c] + (t == c ? gtp[c] : 0.f)) - dot);
        d[c] = accumulate ? d[c] + g : g;
      }
    }
  }
}

// ---------------------------------------------------------------- focal losses on float targets
// mode 0: focal_loss(normalize=False): mean_i w_i * bce_i, w = pt^gamma detached (loss.py:158-176)
// mode 1: sigmoid_focal_loss: sum_i alpha_t * bce_i * (1-p_t)^gamma, gradient through the factor (:179-201)
// mode 2: focal_loss(normalize=True): value sum_i bce_i (the normalisation cancels identically)
__device__ __forceinline__ float softplus_neg_abs(float x) { return log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ void focal_term(float x, float y, float gamma, float alpha, int mode, float& val,
                                           float& grad) {
  const float bce = (1.f - y) * x + (softplus_neg_abs(x) + fmaxf(-x, 0.f));
  const float p = 1.f / (1.f + expf(-x));
  const float dbce = p - y;
  if (mode == 2) {
    val = bce;
    grad = dbce;
  } else if (mode == 0) {
    const float pt = (1.f - p) * y + p * (1.f - y);
    const float w = powf(pt, gamma);
    val = w * bce;
    grad = w * dbce;
  } else {
    const float p_t = p * y + (1.f - p) * (1.f - y);
    const float q = 1.f - p_t;                 // modulating base
    const float dq = -(2.f * y - 1.f) * p * (1.f - p);  // d(1 - p_t)/dx
    const float qg = powf(q, gamma);
    const float dqg = (q > 0.f) ? gamma * powf(q, gamma - 1.f) * dq : 0.f;
    const float a = alpha >= 0.f ? alpha * y + (1.f - alpha) * (1.f - y) : 1.f;
    val = a * bce * qg;
    grad = a * (dbce * qg + bce * dqg);
  }
}
__global__ __launch_bounds__(256) void focal_partial_kernel(const float* __restrict__ logits,
                                                            const float* __restrict__ target, int64_t n, float gamma,
                                                            float alpha, int mode, double* __restrict__ stats) {
  __shared__ double red[4];
  double v = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float val, g;
    focal_term(logits[i], target[i], gamma, alpha, mode, val, g);
    v += (double)val;
  }
  const double s = block_sum(v, red);
  if (threadIdx.x == 0) stats[1 + blockIdx.x] = s;
}
__global__ void scale_store_kernel(const double* stats, double scale, float* loss) {
  if (threadIdx.x == 0) *loss = (float)(stats[0] * scale);
}
__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ logits,
                                                        const float* __restrict__ target, int64_t n, float gamma,
                                                        float alpha, int mode, float scale,
                                                        const float* __restrict__ grad_scale,
                                                        float* __restrict__ dlogits) {
  const float k = (grad_scale ? *grad_scale : 1.f) * scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float val, g;
    focal_term(logits[i], target[i], gamma, alpha, mode, val, g);
    dlogits[i] = k * g;
  }
}

// ---------------------------------------------------------------- confusion matrix
constexpr int kCmLdsBins = 4096;
template <int FROM_LOGITS>
__global__ __launch_bounds__(256) void confusion_kernel(const void* __restrict__ pred_or_logits,
                                                        const int64_t* __restrict__ y_true, int64_t n, int Cl, int C,
                                                        unsigned long long* __restrict__ cm) {
  __shared__ unsigned int h[kCmLdsBins];
  const int bins = C * C;
  const bool lds = bins <= kCmLdsBins;
  if (lds) {
    for (int i = threadIdx.x; i < bins; i += 256) h[i] = 0u;
    __syncthreads();
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t t = y_true[i];
    int64_t pr;
    if (FROM_LOGITS) {
      const float* z = reinterpret_cast<const float*>(pred_or_logits) + i * Cl;
      if (Cl == 1) {
        pr = z[0] > 0.f ? 1 : 0;  // sigmoid(z) > 0.5
      } else {
        int best = 0;
        float bv = z[0];
        for (int c = 1; c < Cl; ++c)
          if (z[c] > bv) { bv = z[c]; best = c; }  // first maximum, as torch.argmax
        pr = best;
      }
    } else {
      pr = reinterpret_cast<const int64_t*>(pred_or_logits)[i];
    }
    if (t < 0 || t >= C || pr < 0 || pr >= C) continue;  // ignore labels (255, -1) fall outside [0, C)
    const int bin = (int)t * C + (int)pr;
    if (lds) atomicAdd(&h[bin], 1u);
    else atomicAdd(&cm[bin], 1ull);
  }
  if (lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256)
      if (h[i]) atomicAdd(&cm[i], (unsigned long long)h[i]);
  }
}

}  // namespace evk

using namespace evk;

extern "C" int64_t evk_prob_stats_doubles(int32_t C) { return (int64_t)3 * C * (1 + kNRBlocks); }

extern "C" int evk_prob_stats(const float* logits, const int64_t* labels, int64_t npix, int32_t C, int64_t ignore_index,
                              double* stats, void* stream) {
  EVK_REQUIRE(logits && labels && stats, EVK_E_INVALID, "prob_stats: null pointer");
  EVK_REQUIRE(C >= 1 && C <= kNRMaxC, EVK_E_UNSUPPORTED, "prob_stats: C=%d outside [1,%d]", C, kNRMaxC);
  hipStream_t st = (hipStream_t)stream;
  const int nblk = nr_grid(npix);
  hipLaunchKernelGGL(prob_stats_kernel, dim3(nblk), dim3(256), 0, st, logits, labels, npix, C, ignore_index, stats);
  hipLaunchKernelGGL(nr_finalize_kernel, dim3(1), dim3(256), 0, st, stats, 3 * C, nblk);
  return check_launch("prob_stats");
}

extern "C" int evk_prob_stats_bwd(const float* logits, const int64_t* labels, int64_t npix, int32_t C,
                                  int64_t ignore_index, const float* g_tp, const float* g_sp, float* dlogits,
                                  int32_t accumulate, void* stream) {
  EVK_REQUIRE(logits && labels && g_tp && g_sp && dlogits, EVK_E_INVALID, "prob_stats_bwd: null pointer");
  EVK_REQUIRE(C >= 1 && C <= kNRMaxC, EVK_E_UNSUPPORTED, "prob_stats_bwd: C=%d outside [1,%d]", C, kNRMaxC);
  const int64_t b = (npix + 255) / 256;
  hipLaunchKernelGGL(prob_stats_bwd_kernel, dim3((unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b))), dim3(256), 0,
                     (hipStream_t)stream, logits, labels, npix, C, ignore_index, g_tp, g_sp, dlogits, accumulate);
  return check_launch("prob_stats_bwd");
}

extern "C" int evk_focal_fwd(const float* logits, const float* target, int64_t n, float gamma, float alpha,
                             int32_t mode, int32_t mean, float* loss, double* stats, void* stream) {
  EVK_REQUIRE(logits && target && loss && stats, EVK_E_INVALID, "focal_fwd: null pointer");
  EVK_REQUIRE(mode >= 0 && mode <= 2, EVK_E_INVALID, "focal_fwd: mode %d", mode);
  hipStream_t st = (hipStream_t)stream;
  const int nblk = nr_grid(n);
  hipLaunchKernelGGL(focal_partial_kernel, dim3(nblk), dim3(256), 0, st, logits, target, n, gamma, alpha, mode, stats);
  hipLaunchKernelGGL(nr_finalize_kernel, dim3(1), dim3(64), 0, st, stats, 1, nblk);
  hipLaunchKernelGGL(scale_store_kernel, dim3(1), dim3(64), 0, st, (const double*)stats, mean ? 1.0 / (double)n : 1.0,
                     loss);
  return check_launch("focal_fwd");
}

extern "C" int evk_focal_bwd(const float* logits, const float* target, int64_t n, float gamma, float alpha,
                             int32_t mode, int32_t mean, const float* grad_scale, float* dlogits, void* stream) {
  EVK_REQUIRE(logits && target && dlogits, EVK_E_INVALID, "focal_bwd: null pointer");
  const int64_t b = (n + 255) / 256;
  hipLaunchKernelGGL(focal_bwd_kernel, dim3((unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b))), dim3(256), 0,
                     (hipStream_t)stream, logits, target, n, gamma, alpha, mode, mean ? 1.f / (float)n : 1.f,
                     grad_scale, dlogits);
  return check_launch("focal_bwd");
}

extern "C" int evk_confusion_matrix(const int64_t* y_true, const int64_t* y_pred, int64_t n, int32_t num_classes,
                                    int64_t* cm, void* stream) {
  EVK_REQUIRE(y_true && y_pred && cm, EVK_E_INVALID, "confusion_matrix: null pointer");
  EVK_REQUIRE(num_classes >= 1 && num_classes <= 4096, EVK_E_UNSUPPORTED, "confusion_matrix: %d classes", num_classes);
  hipLaunchKernelGGL(confusion_kernel<0>, dim3(nr_grid(n)), dim3(256), 0, (hipStream_t)stream, (const void*)y_pred, y_true,
                     n, 0, num_classes, reinterpret_cast<unsigned long long*>(cm));
  return check_launch("confusion_matrix");
}

extern "C" int evk_confusion_from_logits(const float* logits, const int64_t* y_true, int64_t npix, int32_t C_logits,
                                         int32_t num_classes, int64_t* cm, void* stream) {
  EVK_REQUIRE(logits && y_true && cm, EVK_E_INVALID, "confusion_from_logits: null pointer");
  EVK_REQUIRE(C_logits >= 1 && num_classes >= (C_logits == 1 ? 2 : C_logits) && num_classes <= 4096, EVK_E_INVALID,
              "confusion_from_logits: %d logit channels vs %d classes", C_logits, num_classes);
  hipLaunchKernelGGL(confusion_kernel<1>, dim3(nr_grid(npix)), dim3(256), 0, (hipStream_t)stream, (const void*)logits,
                     y_true, npix, C_logits, num_classes, reinterpret_cast<unsigned long long*>(cm));
  return check_launch("confusion_from_logits");
}

// ---------------------------------------------------------------- per-pixel cross entropy + OHEM
// F.cross_entropy(reduction='none', ignore_index): loss_i = lse(z_i) - z_i[t_i], 0 on ignored pixels; the
// backward takes a per-pixel upstream gradient.  online_hard_example_mining (reference loss.py:146-155): mean of
// the non-zero values among the k = int(keep_ratio * N) largest losses.  The k-th largest value is found exactly by
// a three-pass radix select on the order-preserving integer image of the floats (11 + 11 + 10 bits); ties at the
// threshold are admitted through a counter (which of several bit-identical losses is kept is arbitrary, as in
// torch.topk; the loss value does not depend on it).
namespace evk {

__global__ __launch_bounds__(256) void ce_pixel_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                       int64_t npix, int C, int64_t ignore, float* __restrict__ loss) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t t = labels[i];
    if (t == ignore || t < 0 || t >= C) {
      loss[i] = 0.f;
      continue;
    }
    const float* z = logits + i * C;
    float m = z[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(z[c] - m);
    loss[i] = (m + logf(s)) - z[t];
  }
}
__global__ __launch_bounds__(256) void ce_pixel_bwd_kernel(const float* __restrict__ logits,
                                                           const int64_t* __restrict__ labels, int64_t npix, int C,
                                                           int64_t ignore, const float* __restrict__ gpix,
                                                           float* __restrict__ dlogits) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t t = labels[i];
    float* d = dlogits + i * C;
    const float g = gpix[i];
    if (t == ignore || t < 0 || t >= C || g == 0.f) {
      for (int c = 0; c < C; ++c) d[c] = 0.f;
      continue;
    }
    const float* z = logits + i * C;
    float m = z[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, z[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(z[c] - m);
    const float lse = m + logf(s);
    for (int c = 0; c < C; ++c) d[c] = g * (expf(z[c] - lse) - (t == c ? 1.f : 0.f));
  }
}

__device__ __forceinline__ uint32_t ordered_bits(float v) {
  const uint32_t b = __builtin_bit_cast(uint32_t, v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // monotone: a < b  <=>  ordered(a) < ordered(b)
}
// state: [0] prefix value (ordered bits, high part fixed so far)  [1] remaining rank (k-th largest inside the prefix)
//        [2] count of elements strictly above the threshold       [3] ties admitted so far (backward)
//        [4..4+2048) histogram
__global__ __launch_bounds__(256) void ohem_hist_kernel(const float* __restrict__ v, int64_t n, int shift, int bits,
                                                        uint32_t prefix_mask, unsigned long long* __restrict__ state) {
  __shared__ unsigned int h[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0u;
  __syncthreads();
  const uint32_t prefix = (uint32_t)state[0];
  const uint32_t bmask = (1u << bits) - 1u;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t u = ordered_bits(v[i]);
    if ((u & prefix_mask) == (prefix & prefix_mask)) atomicAdd(&h[(u >> shift) & bmask], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (1 << bits); i += 256)
    if (h[i]) atomicAdd(&state[4 + i], (unsigned long long)h[i]);
}
// walk the histogram from the top: find the bin holding the remaining-rank-th largest element
__global__ void ohem_pick_kernel(unsigned long long* __restrict__ state, int shift, int bits) {
  if (threadIdx.x != 0) return;
  unsigned long long rank = state[1];  // 1-based rank inside the current prefix
  unsigned long long above = state[2];
  int b = (1 << bits) - 1;
  for (; b > 0; --b) {
    const unsigned long long c = state[4 + b];
    if (rank <= c) break;
    rank -= c;
    above += c;
  }
  state[0] = (state[0] | ((unsigned long long)b << shift)) & 0xffffffffull;
  state[1] = rank;
  state[2] = above;
  for (int i = 0; i < (1 << bits); ++i) state[4 + i] = 0ull;
}
// sums: [0] sum of kept values, [1] number of kept non-zero values
__global__ __launch_bounds__(256) void ohem_sum_kernel(const float* __restrict__ v, int64_t n,
                                                       const unsigned long long* __restrict__ state,
                                                       double* __restrict__ partial) {
  __shared__ double red[4];
  const uint32_t thr = (uint32_t)state[0];
  double s = 0.0, c = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float x = v[i];
    if (ordered_bits(x) > thr && x != 0.f) {
      s += (double)x;
      c += 1.0;
    }
  }
  const double ts = block_sum(s, red), tc = block_sum(c, red);
  if (threadIdx.x == 0) {
    partial[2 + 2 * blockIdx.x] = ts;
    partial[3 + 2 * blockIdx.x] = tc;
  }
}
__global__ void ohem_finish_kernel(double* __restrict__ partial, int nblk, const unsigned long long* __restrict__ state,
                                   float* __restrict__ loss) {
  if (threadIdx.x != 0) return;
  double s = 0.0, c = 0.0;
  for (int b = 0; b < nblk; ++b) {
    s += partial[2 + 2 * b];
    c += partial[3 + 2 * b];
  }
  const uint32_t thr = (uint32_t)state[0];
  const uint32_t raw = (thr & 0x80000000u) ? (thr & 0x7fffffffu) : ~thr;
  const float tv = __builtin_bit_cast(float, raw);
  const double ties = (double)state[1];  // elements equal to the threshold that belong to the top k
  if (tv != 0.f) {
    s += ties * (double)tv;
    c += ties;
  }
  partial[0] = s;
  partial[1] = c;
  *loss = (float)(s / c);  // no non-zero value kept: 0/0 = NaN, as the mean of an empty selection
}
__global__ __launch_bounds__(256) void ohem_bwd_kernel(const float* __restrict__ v, int64_t n,
                                                       unsigned long long* __restrict__ state,
                                                       const double* __restrict__ partial,
                                                       const float* __restrict__ grad_scale, float* __restrict__ dv) {
  const uint32_t thr = (uint32_t)state[0];
  const unsigned long long ties = state[1];
  const float g = (grad_scale ? *grad_scale : 1.f) / (float)partial[1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float x = v[i];
    const uint32_t u = ordered_bits(x);
    float o = 0.f;
    if (x != 0.f) {
      if (u > thr) o = g;
      else if (u == thr && atomicAdd(&state[3], 1ull) < ties) o = g;
    }
    dv[i] = o;
  }
}

}  // namespace evk

extern "C" int evk_ce_pixel_fwd(const float* logits, const int64_t* labels, int64_t npix, int32_t C, int64_t ignore_index,
                                float* loss_pix, void* stream) {
  EVK_REQUIRE(logits && labels && loss_pix && C >= 2, EVK_E_INVALID, "ce_pixel_fwd: bad argument");
  const int64_t b = (npix + 255) / 256;
  hipLaunchKernelGGL(ce_pixel_kernel, dim3((unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b))), dim3(256), 0,
                     (hipStream_t)stream, logits, labels, npix, C, ignore_index, loss_pix);
  return check_launch("ce_pixel_fwd");
}
extern "C" int evk_ce_pixel_bwd(const float* logits, const int64_t* labels, int64_t npix, int32_t C, int64_t ignore_index,
                                const float* grad_pix, float* dlogits, void* stream) {
  EVK_REQUIRE(logits && labels && grad_pix && dlogits && C >= 2, EVK_E_INVALID, "ce_pixel_bwd: bad argument");
  const int64_t b = (npix + 255) / 256;
  hipLaunchKernelGGL(ce_pixel_bwd_kernel, dim3((unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b))), dim3(256), 0,
                     (hipStream_t)stream, logits, labels, npix, C, ignore_index, grad_pix, dlogits);
  return check_launch("ce_pixel_bwd");
}

extern "C" int64_t evk_ohem_state_bytes(void) { return (int64_t)((4 + 2048) * 8 + (2 + 2 * kNRBlocks) * 8); }

extern "C" int evk_ohem_fwd(const float* losses, int64_t n, int64_t keep, float* loss, void* state, void* stream) {
  EVK_REQUIRE(losses && loss && state && n > 0 && keep > 0 && keep <= n, EVK_E_INVALID, "ohem_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* s = reinterpret_cast<unsigned long long*>(state);
  double* partial = reinterpret_cast<double*>(s + 4 + 2048);
  hipError_t e = hipMemsetAsync(state, 0, (size_t)evk_ohem_state_bytes(), st);
  if (e != hipSuccess) { set_error("ohem_fwd: memset: %s", hipGetErrorString(e)); return EVK_E_LAUNCH; }
  e = hipMemcpyAsync(s + 1, &keep, sizeof(keep), hipMemcpyHostToDevice, st);  // rank = k (k-th largest)
  if (e != hipSuccess) { set_error("ohem_fwd: memcpy: %s", hipGetErrorString(e)); return EVK_E_LAUNCH; }
  const int nblk = nr_grid(n);
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  const uint32_t masks[3] = {0u, 0xffe00000u, 0xfffffc00u};
  for (int p = 0; p < 3; ++p) {
    hipLaunchKernelGGL(ohem_hist_kernel, dim3(nblk), dim3(256), 0, st, losses, n, shifts[p], bits[p], masks[p], s);
    hipLaunchKernelGGL(ohem_pick_kernel, dim3(1), dim3(64), 0, st, s, shifts[p], bits[p]);
  }
  hipLaunchKernelGGL(ohem_sum_kernel, dim3(nblk), dim3(256), 0, st, losses, n, (const unsigned long long*)s, partial);
  hipLaunchKernelGGL(ohem_finish_kernel, dim3(1), dim3(64), 0, st, partial, nblk, (const unsigned long long*)s, loss);
  return check_launch("ohem_fwd");
}

extern "C" int evk_ohem_bwd(const float* losses, int64_t n, void* state, const float* grad_scale, float* dlosses,
                            void* stream) {
  EVK_REQUIRE(losses && state && dlosses && n > 0, EVK_E_INVALID, "ohem_bwd: bad argument");
  unsigned long long* s = reinterpret_cast<unsigned long long*>(state);
  const double* partial = reinterpret_cast<const double*>(s + 4 + 2048);
  const int64_t b = (n + 255) / 256;
  hipLaunchKernelGGL(ohem_bwd_kernel, dim3((unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b))), dim3(256), 0,
                     (hipStream_t)stream, losses, n, s, partial, grad_scale, dlosses);
  return check_launch("ohem_bwd");
}
