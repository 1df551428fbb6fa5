// BatchNorm2d (+ residual add, + ReLU) forward / backward on NHWC fp32 for gfx950.
// Replaces aten::batch_norm / native_batch_norm_backward, aten::add_, aten::relu_ at reference
// ever/module/_resnets.py:95-112 (bn1..bn3, `out += identity`, relu), fs_relation.py:39-53,
// fpn.py:163-167.  HBM-bound: every kernel streams [rows][C] with 16-byte accesses, a workgroup's
// row range is one contiguous span.  Statistics are reduced in two stages (fp32 partials per
// workgroup, fp64 finalisation) so results do not depend on the launch grid.
#include "common.hpp"
#include <stdlib.h>

namespace evk {

constexpr int kMaxStatBlocks = 2048;

struct BnPlan {
  int nblk;
  int64_t rows_per_blk;
  int tpc, rl;
};
static BnPlan bn_plan(int64_t rows, int C) {
  BnPlan p;
  const int c4 = C / 4;
  p.tpc = c4 < 256 ? c4 : 256;
  p.rl = 256 / p.tpc;
  int64_t nb = (rows * (int64_t)C + 65535) / 65536;  // ~64K elements per workgroup
  if (nb > kMaxStatBlocks) nb = kMaxStatBlocks;
  if (nb < 1) nb = 1;
  int64_t rpb = (rows + nb - 1) / nb;
  rpb = ((rpb + p.rl - 1) / p.rl) * p.rl;
  p.rows_per_blk = rpb;
  p.nblk = (int)((rows + rpb - 1) / rpb);
  return p;
}

// partial[blk][0][C] = sum (x - pivot), partial[blk][1][C] = sum (x - pivot)^2 with pivot[c] = x[0][c].
// Shifting by a sample of the same channel keeps var = E[d^2] - E[d]^2 free of the catastrophic
// cancellation the raw moments suffer when |mean| >> std (8-sample statistics at 2x2 maps).
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                               int64_t rows, int C, int64_t rows_per_blk, int tpc,
                                                               int rl) {
  __shared__ f32x4 red[2][256];
  const int c4 = C >> 2;
  const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r1 = min(rows, r0 + rows_per_blk);
  for (int cb = tc; cb < c4; cb += tpc) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    const f32x4 pv = *reinterpret_cast<const f32x4*>(x + cb * 4);
    if (tr < rl) {
      // 4 independent 16-byte loads in flight per lane: the reduction is latency bound otherwise
      // (measured 2.7 TB/s with one load per lane per trip)
      int64_t r = r0 + tr;
      const int64_t st = rl;
      for (; r + 3 * st < r1; r += 4 * st) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + r * C + cb * 4) - pv;
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + (r + st) * C + cb * 4) - pv;
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(x + (r + 2 * st) * C + cb * 4) - pv;
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(x + (r + 3 * st) * C + cb * 4) - pv;
        s += (v0 + v1) + (v2 + v3);
        q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      }
      for (; r < r1; r += st) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + cb * 4) - pv;
        s += v;
        q += v * v;
      }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (tr == 0) {
      for (int k = 1; k < rl; ++k) {
        s += red[0][k * tpc + tc];
        q += red[1][k * tpc + tc];
      }
      float* o = partial + (size_t)blockIdx.x * 2 * C;
      *reinterpret_cast<f32x4*>(o + cb * 4) = s;
      *reinterpret_cast<f32x4*>(o + C + cb * 4) = q;
    }
    __syncthreads();
  }
}

// Sum the per-workgroup partials of 8 channels with 32 lanes per channel in fp64 (four independent loads
// in flight per lane: the chain of up to 2048 partials per channel is latency bound, and a grid of C/8
// workgroups instead of C/32 spreads it over more CUs), then fold the 32 lanes through LDS in a fixed
// order.  Returns true on the lane that holds the totals.
constexpr int kFinCh = 8, kFinLanes = 32;
__device__ __forceinline__ bool reduce_partials(const float* __restrict__ partial, int nblk, int C, int& c, double& s,
                                                double& q) {
  __shared__ double red[2][kFinLanes][kFinCh];
  const int tc = threadIdx.x % kFinCh, tl = threadIdx.x / kFinCh;
  c = blockIdx.x * kFinCh + tc;
  s = 0.0;
  q = 0.0;
  if (c < C) {
    const float* ps = partial + c;
    const size_t st = (size_t)2 * C;
    int b = tl;
    for (; b + 3 * kFinLanes < nblk; b += 4 * kFinLanes) {
      const float s0 = ps[b * st], s1 = ps[(b + kFinLanes) * st], s2 = ps[(b + 2 * kFinLanes) * st],
                  s3 = ps[(b + 3 * kFinLanes) * st];
      const float q0 = ps[b * st + C], q1 = ps[(b + kFinLanes) * st + C], q2 = ps[(b + 2 * kFinLanes) * st + C],
                  q3 = ps[(b + 3 * kFinLanes) * st + C];
      s += ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
      q += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
    }
    for (; b < nblk; b += kFinLanes) {
      s += (double)ps[b * st];
      q += (double)ps[b * st + C];
    }
  }
  red[0][tl][tc] = s;
  red[1][tl][tc] = q;
  __syncthreads();
  if (tl != 0 || c >= C) return false;
  for (int k = 1; k < kFinLanes; ++k) {
    s += red[0][k][tc];
    q += red[1][k][tc];
  }
  return true;
}

// mean / biased var -> save_mean, save_invstd, scale/shift for the apply pass; running stats update
// follows torch: running = (1-m)*running + m*stat, with the UNBIASED variance (n/(n-1)).
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ partial, int nblk, int C,
                                                             double inv_rows, double unbias,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float momentum,
                                                             float eps, float* __restrict__ save_mean,
                                                             float* __restrict__ save_invstd,
                                                             float* __restrict__ scale_shift) {
  int c;
  double s, q;
  if (!reduce_partials(partial, nblk, C, c, s, q)) return;
  const double dm = s * inv_rows;  // mean of (x - pivot)
  const double mean = (double)x[c] + dm;
  double var = q * inv_rows - dm * dm;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  save_mean[c] = meanf;
  save_invstd[c] = invstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = g * invstd;
  scale_shift[c] = sc;
  scale_shift[C + c] = b - meanf * sc;
}

__global__ void bn_eval_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                                    float* __restrict__ scale_shift, float* __restrict__ save_mean,
                                    float* __restrict__ save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = (float)(1.0 / sqrt((double)rv[c] + (double)eps));
  if (save_mean) save_mean[c] = rm[c];
  if (save_invstd) save_invstd[c] = invstd;
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = g * invstd;
  scale_shift[c] = sc;
  scale_shift[C + c] = b - rm[c] * sc;
}

// y = act(x*scale + shift [+ residual]);  scale/shift staged in LDS
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ residual,
                                                       const float* __restrict__ scale_shift, float* __restrict__ y,
                                                       size_t n4, int C, int relu, int unroll) {
  extern __shared__ __attribute__((aligned(16))) float ss[];  // [2][C]
  for (int i = threadIdx.x; i < 2 * C; i += 256) ss[i] = scale_shift[i];
  __syncthreads();
  const int c4 = C >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const f32x4* r4 = reinterpret_cast<const f32x4*>(residual);
  f32x4* y4 = reinterpret_cast<f32x4*>(y);
  const f32x4* sc4 = reinterpret_cast<const f32x4*>(ss);
  const f32x4* sh4 = reinterpret_cast<const f32x4*>(ss + C);
  // Four independent 16-byte loads per lane and trip: with one, 8 waves/SIMD x 1 KB keeps only ~8 MB in
  // flight chip-wide, below latency x bandwidth (measured 3.4 TB/s on the 268 MB maps).  The channel
  // chunk index advances by a constant per trip (no 64-bit modulo in the loop).
  const size_t S = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  int cb = (int)(i % c4);
  const int dcb = (int)(S % c4);
  auto fin = [&](f32x4 v, const f32x4* rp) {
    if (residual) v += *rp;
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    return v;
  };
  for (; unroll && i + 3 * S < n4; i += 4 * S) {
    int c1 = cb + dcb; c1 = c1 >= c4 ? c1 - c4 : c1;
    int c2 = c1 + dcb; c2 = c2 >= c4 ? c2 - c4 : c2;
    int c3 = c2 + dcb; c3 = c3 >= c4 ? c3 - c4 : c3;
    const f32x4 a0 = x4[i], a1 = x4[i + S], a2 = x4[i + 2 * S], a3 = x4[i + 3 * S];
    f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0, q3 = q0;
    if (residual) { q0 = r4[i]; q1 = r4[i + S]; q2 = r4[i + 2 * S]; q3 = r4[i + 3 * S]; }
    y4[i] = fin(a0 * sc4[cb] + sh4[cb], &q0);
    y4[i + S] = fin(a1 * sc4[c1] + sh4[c1], &q1);
    y4[i + 2 * S] = fin(a2 * sc4[c2] + sh4[c2], &q2);
    y4[i + 3 * S] = fin(a3 * sc4[c3] + sh4[c3], &q3);
    cb = c3 + dcb; cb = cb >= c4 ? cb - c4 : cb;
  }
  for (; i < n4; i += S) {
    f32x4 q0 = {0.f, 0.f, 0.f, 0.f};
    if (residual) q0 = r4[i];
    y4[i] = fin(x4[i] * sc4[cb] + sh4[cb], &q0);
    cb += dcb; cb = cb >= c4 ? cb - c4 : cb;
  }
}

// Backward stage 1: g = dy * (y > 0) ; partial[blk][0][C] = sum g, [1][C] = sum g * xhat.
// Optionally writes g to d_residual.
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float* __restrict__ d_residual,
                                                             float* __restrict__ partial, int64_t rows, int C,
                                                             int64_t rows_per_blk, int tpc, int rl, int relu) {
  // relu: 0 none, 1 mask from the saved output y, 2 mask recomputed from x (no residual: the
  // forward's pre-activation is x*sc+sh with the same sc/sh arithmetic as bn_stats_final_kernel)
  __shared__ f32x4 red[2][256];
  const int c4 = C >> 2;
  const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r1 = min(rows, r0 + rows_per_blk);
  for (int cb = tc; cb < c4; cb += tpc) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + cb * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + cb * 4);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (relu == 2) {
      const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
      const f32x4 g4 = gamma ? *reinterpret_cast<const f32x4*>(gamma + cb * 4) : one;
      const f32x4 b4 = beta ? *reinterpret_cast<const f32x4*>(beta + cb * 4) : zero;
      sc = g4 * is;
      sh = b4 - mu * sc;
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (tr < rl) {
      auto one = [&](const f32x4 gin, const f32x4 xv, const f32x4 yin, size_t off) {
        f32x4 g = gin;
        if (relu) {
          const f32x4 yy = (relu == 1) ? yin : xv * sc + sh;
          g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
          g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        }
        if (d_residual) *reinterpret_cast<f32x4*>(d_residual + off) = g;
        s += g;
        q += g * ((xv - mu) * is);
      };
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      int64_t r = r0 + tr;
      const int64_t st = rl;
      // four rows per trip: 8-12 independent 16-byte loads in flight per lane
      for (; r + 3 * st < r1; r += 4 * st) {
        const size_t o0 = (size_t)r * C + cb * 4, o1 = (size_t)(r + st) * C + cb * 4;
        const size_t o2 = (size_t)(r + 2 * st) * C + cb * 4, o3 = (size_t)(r + 3 * st) * C + cb * 4;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(dy + o0), g1 = *reinterpret_cast<const f32x4*>(dy + o1);
        const f32x4 g2 = *reinterpret_cast<const f32x4*>(dy + o2), g3 = *reinterpret_cast<const f32x4*>(dy + o3);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + o0), x1 = *reinterpret_cast<const f32x4*>(x + o1);
        const f32x4 x2 = *reinterpret_cast<const f32x4*>(x + o2), x3 = *reinterpret_cast<const f32x4*>(x + o3);
        f32x4 y0 = zero4, y1 = zero4, y2 = zero4, y3 = zero4;
        if (relu == 1) {
          y0 = *reinterpret_cast<const f32x4*>(y + o0);
          y1 = *reinterpret_cast<const f32x4*>(y + o1);
          y2 = *reinterpret_cast<const f32x4*>(y + o2);
          y3 = *reinterpret_cast<const f32x4*>(y + o3);
        }
        one(g0, x0, y0, o0);
        one(g1, x1, y1, o1);
        one(g2, x2, y2, o2);
        one(g3, x3, y3, o3);
      }
      for (; r < r1; r += st) {
        const size_t o0 = (size_t)r * C + cb * 4;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(dy + o0);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + o0);
        const f32x4 y0 = (relu == 1) ? *reinterpret_cast<const f32x4*>(y + o0) : zero4;
        one(g0, x0, y0, o0);
      }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (tr == 0) {
      for (int k = 1; k < rl; ++k) {
        s += red[0][k * tpc + tc];
        q += red[1][k * tpc + tc];
      }
      float* o = partial + (size_t)blockIdx.x * 2 * C;
      *reinterpret_cast<f32x4*>(o + cb * 4) = s;
      *reinterpret_cast<f32x4*>(o + C + cb * 4) = q;
    }
    __syncthreads();
  }
}

// coef[0][C] = gamma*invstd ; coef[1][C] = mean(g) ; coef[2][C] = mean(g*xhat)  (0 when !train)
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ partial, int nblk, int C,
                                                           double inv_rows, const float* __restrict__ gamma,
                                                           const float* __restrict__ invstd,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ coef, int train) {
  int c;
  double s, q;
  if (!reduce_partials(partial, nblk, C, c, s, q)) return;
  if (dbeta) dbeta[c] = (float)s;
  if (dgamma) dgamma[c] = (float)q;
  const float g = gamma ? gamma[c] : 1.f;
  coef[c] = g * invstd[c];
  coef[C + c] = train ? (float)(s * inv_rows) : 0.f;
  coef[2 * C + c] = train ? (float)(q * inv_rows) : 0.f;
}

// dx = coef0 * (g - coef1 - xhat*coef2)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ coef,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ dx,
                                                           size_t n4, int C, int relu, int unroll) {
  extern __shared__ __attribute__((aligned(16))) float ss[];  // [7][C]: coef0..2, mean, invstd, sc, sh
  for (int i = threadIdx.x; i < 3 * C; i += 256) ss[i] = coef[i];
  for (int i = threadIdx.x; i < C; i += 256) {
    const float m = mean[i], is_ = invstd[i];
    ss[3 * C + i] = m;
    ss[4 * C + i] = is_;
    const float sc_ = (gamma ? gamma[i] : 1.f) * is_;
    ss[5 * C + i] = sc_;
    ss[6 * C + i] = (beta ? beta[i] : 0.f) - m * sc_;
  }
  __syncthreads();
  const f32x4* sc4 = reinterpret_cast<const f32x4*>(ss + 5 * C);
  const f32x4* sh4 = reinterpret_cast<const f32x4*>(ss + 6 * C);
  const int c4 = C >> 2;
  const f32x4* k0 = reinterpret_cast<const f32x4*>(ss);
  const f32x4* k1 = reinterpret_cast<const f32x4*>(ss + C);
  const f32x4* k2 = reinterpret_cast<const f32x4*>(ss + 2 * C);
  const f32x4* mu = reinterpret_cast<const f32x4*>(ss + 3 * C);
  const f32x4* is = reinterpret_cast<const f32x4*>(ss + 4 * C);
  const f32x4* dy4 = reinterpret_cast<const f32x4*>(dy);
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const f32x4* y4 = reinterpret_cast<const f32x4*>(y);
  f32x4* dx4 = reinterpret_cast<f32x4*>(dx);
  const size_t S = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  int cb = (int)(i % c4);
  const int dcb = (int)(S % c4);
  auto one = [&](f32x4 g, const f32x4 xv, const f32x4 yv, int c) {
    if (relu) {
      const f32x4 yy = (relu == 1) ? yv : xv * sc4[c] + sh4[c];
      g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
      g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
    }
    const f32x4 xh = (xv - mu[c]) * is[c];
    return k0[c] * (g - k1[c] - xh * k2[c]);
  };
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  // two elements per trip = 4-6 independent 16-byte loads per lane in flight (see bn_apply_kernel)
  for (; unroll && i + S < n4; i += 2 * S) {
    int c1 = cb + dcb; c1 = c1 >= c4 ? c1 - c4 : c1;
    const f32x4 g0 = dy4[i], g1 = dy4[i + S];
    const f32x4 x0 = x4[i], x1 = x4[i + S];
    f32x4 y0 = z4, y1 = z4;
    if (relu == 1) { y0 = y4[i]; y1 = y4[i + S]; }
    dx4[i] = one(g0, x0, y0, cb);
    dx4[i + S] = one(g1, x1, y1, c1);
    cb = c1 + dcb; cb = cb >= c4 ? cb - c4 : cb;
  }
  for (; i < n4; i += S) {
    const f32x4 y0 = (relu == 1) ? y4[i] : z4;
    dx4[i] = one(dy4[i], x4[i], y0, cb);
    cb += dcb; cb = cb >= c4 ? cb - c4 : cb;
  }
}

static int bn_unroll() {
  static const int u = getenv("EVK_BN_UNROLL") ? atoi(getenv("EVK_BN_UNROLL")) : 1;
  return u;
}
static int stream_grid(size_t n4) {
  size_t b = (n4 + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace evk

using namespace evk;

extern "C" size_t evk_bn_workspace_bytes(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return 0;
  return ((size_t)kMaxStatBlocks * 2 * C + 8 * (size_t)C) * sizeof(float);
}

extern "C" int evk_bn_fwd_train(const float* x, const float* residual, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* y,
                                float* save_mean, float* save_invstd, int64_t rows, int32_t C, uint32_t flags,
                                void* workspace, size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(x && y && save_mean && save_invstd, EVK_E_INVALID, "bn_fwd_train: null pointer");
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, EVK_E_UNSUPPORTED, "bn_fwd_train: rows=%lld C=%d",
              (long long)rows, C);
  EVK_REQUIRE(workspace && workspace_bytes >= evk_bn_workspace_bytes(rows, C), EVK_E_WORKSPACE,
              "bn_fwd_train: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const BnPlan pl = bn_plan(rows, C);
  float* partial = (float*)workspace;
  float* scale_shift = partial + (size_t)kMaxStatBlocks * 2 * C;
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(pl.nblk), dim3(256), 0, st, x, partial, rows, C, pl.rows_per_blk,
                     pl.tpc, pl.rl);
  int rc = check_launch("bn_stats_partial");
  if (rc) return rc;
  const double unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, st, x, partial, pl.nblk, C,
                     1.0 / (double)rows, unbias, gamma, beta, running_mean, running_var, momentum, eps, save_mean,
                     save_invstd, scale_shift);
  rc = check_launch("bn_stats_final");
  if (rc) return rc;
  const size_t n4 = (size_t)rows * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_grid(n4)), dim3(256), 2 * C * sizeof(float), st, x, residual,
                     scale_shift, y, n4, C, (flags & EVK_BN_RELU) ? 1 : 0, bn_unroll());
  return check_launch("bn_apply");
}

extern "C" int evk_bn_fwd_eval(const float* x, const float* residual, const float* gamma, const float* beta,
                               const float* running_mean, const float* running_var, float eps, float* y,
                               float* save_mean, float* save_invstd, int64_t rows, int32_t C, uint32_t flags,
                               void* workspace, size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(x && y && running_mean && running_var, EVK_E_INVALID, "bn_fwd_eval: null pointer");
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, EVK_E_UNSUPPORTED, "bn_fwd_eval: rows=%lld C=%d",
              (long long)rows, C);
  EVK_REQUIRE(workspace && workspace_bytes >= 2 * (size_t)C * sizeof(float), EVK_E_WORKSPACE,
              "bn_fwd_eval: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* scale_shift = (float*)workspace;
  hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, running_mean,
                     running_var, eps, C, scale_shift, save_mean, save_invstd);
  int rc = check_launch("bn_eval_coef");
  if (rc) return rc;
  const size_t n4 = (size_t)rows * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_grid(n4)), dim3(256), 2 * C * sizeof(float), st, x, residual,
                     scale_shift, y, n4, C, (flags & EVK_BN_RELU) ? 1 : 0, bn_unroll());
  return check_launch("bn_apply");
}

extern "C" int evk_bn_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
                          const float* save_mean, const float* save_invstd, float* dx, float* d_residual,
                          float* dgamma, float* dbeta, int64_t rows, int32_t C, uint32_t flags, int32_t train,
                          void* workspace, size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(dy && x && save_mean && save_invstd && dx, EVK_E_INVALID, "bn_bwd: null pointer");
  // ReLU mask: from the saved output y when given (needed with a residual), else recomputed from x
  const int relu = (flags & EVK_BN_RELU) ? (y ? 1 : 2) : 0;
  EVK_REQUIRE(relu != 2 || !d_residual, EVK_E_INVALID,
              "bn_bwd: a residual branch needs the forward output y for the ReLU mask");
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, EVK_E_UNSUPPORTED, "bn_bwd: rows=%lld C=%d",
              (long long)rows, C);
  EVK_REQUIRE(workspace && workspace_bytes >= evk_bn_workspace_bytes(rows, C), EVK_E_WORKSPACE,
              "bn_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const BnPlan pl = bn_plan(rows, C);
  float* partial = (float*)workspace;
  float* coef = partial + (size_t)kMaxStatBlocks * 2 * C;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(pl.nblk), dim3(256), 0, st, dy, x, y, save_mean, save_invstd,
                     gamma, beta, d_residual, partial, rows, C, pl.rows_per_blk, pl.tpc, pl.rl, relu);
  int rc = check_launch("bn_bwd_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, st, partial, pl.nblk, C,
                     1.0 / (double)rows, gamma, save_invstd, dgamma, dbeta, coef, train ? 1 : 0);
  rc = check_launch("bn_bwd_final");
  if (rc) return rc;
  const size_t n4 = (size_t)rows * C / 4;
  // when d_residual holds g already, stage 3 can read it instead of re-masking dy
  const float* gsrc = d_residual ? d_residual : dy;
  const int relu3 = d_residual ? 0 : relu;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4)), dim3(256), 7 * C * sizeof(float), st, gsrc, x, y,
                     save_mean, save_invstd, coef, gamma, beta, dx, n4, C, relu3, bn_unroll());
  return check_launch("bn_bwd_apply");
}

// ------------------------------------------------------------------------------------------------
// Staged entry points for synchronized BatchNorm (torch.nn.SyncBatchNorm under
// `sync_bn=True`, reference ever/trainer/th_ddp_trainer.py): the same kernels as above with the
// cross-rank exchange left to the caller between the stages (one small all-gather forward, one small
// all-reduce backward, on torch.distributed / RCCL).
namespace evk {

// stats[c] = local mean, stats[C + c] = local sum of squared deviations from it (fp64)
__global__ __launch_bounds__(256) void bn_local_final_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                             int nblk, int C, double rows, double* __restrict__ stats) {
  int c;
  double s, q;
  if (!reduce_partials(partial, nblk, C, c, s, q)) return;
  const double dm = s / rows;  // mean of (x - pivot)
  stats[c] = (double)x[c] + dm;
  double m2 = q - s * dm;
  stats[C + c] = m2 > 0.0 ? m2 : 0.0;
}
__global__ void bn_coef_from_stats_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                          const float* __restrict__ mean, const float* __restrict__ invstd, int C,
                                          float* __restrict__ scale_shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = (gamma ? gamma[c] : 1.f) * invstd[c];
  scale_shift[c] = sc;
  scale_shift[C + c] = (beta ? beta[c] : 0.f) - mean[c] * sc;
}
__global__ __launch_bounds__(256) void bn_bwd_sums_final_kernel(const float* __restrict__ partial, int nblk, int C,
                                                                double* __restrict__ sums) {
  int c;
  double s, q;
  if (!reduce_partials(partial, nblk, C, c, s, q)) return;
  sums[c] = s;
  sums[C + c] = q;
}
__global__ void bn_bwd_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ invstd,
                                   const float* __restrict__ mean_g, const float* __restrict__ mean_gx, int C,
                                   float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  coef[c] = (gamma ? gamma[c] : 1.f) * invstd[c];
  coef[C + c] = mean_g[c];
  coef[2 * C + c] = mean_gx[c];
}

}  // namespace evk

static int bn_stage_check(const char* what, int64_t rows, int32_t C, const void* ws, size_t ws_bytes) {
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, EVK_E_UNSUPPORTED, "%s: rows=%lld C=%d", what, (long long)rows,
              C);
  EVK_REQUIRE(ws && ws_bytes >= evk_bn_workspace_bytes(rows, C), EVK_E_WORKSPACE, "%s: workspace too small", what);
  return EVK_OK;
}

extern "C" int evk_bn_local_stats(const float* x, double* stats, int64_t rows, int32_t C, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(x && stats, EVK_E_INVALID, "bn_local_stats: null pointer");
  int rc = bn_stage_check("bn_local_stats", rows, C, workspace, workspace_bytes);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const BnPlan pl = bn_plan(rows, C);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(pl.nblk), dim3(256), 0, st, x, partial, rows, C, pl.rows_per_blk,
                     pl.tpc, pl.rl);
  hipLaunchKernelGGL(bn_local_final_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, st, x, (const float*)partial,
                     pl.nblk, C, (double)rows, stats);
  return check_launch("bn_local_stats");
}

extern "C" int evk_bn_apply_stats(const float* x, const float* residual, const float* gamma, const float* beta,
                                  const float* mean, const float* invstd, float* y, int64_t rows, int32_t C,
                                  uint32_t flags, void* workspace, size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(x && y && mean && invstd, EVK_E_INVALID, "bn_apply_stats: null pointer");
  int rc = bn_stage_check("bn_apply_stats", rows, C, workspace, workspace_bytes);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  float* scale_shift = (float*)workspace;
  hipLaunchKernelGGL(bn_coef_from_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, mean, invstd, C,
                     scale_shift);
  const size_t n4 = (size_t)rows * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_grid(n4)), dim3(256), 2 * C * sizeof(float), st, x, residual,
                     (const float*)scale_shift, y, n4, C, (flags & EVK_BN_RELU) ? 1 : 0, bn_unroll());
  return check_launch("bn_apply_stats");
}

extern "C" int evk_bn_bwd_local_sums(const float* dy, const float* x, const float* y, const float* gamma,
                                     const float* beta, const float* mean, const float* invstd, float* d_residual,
                                     double* sums, int64_t rows, int32_t C, uint32_t flags, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(dy && x && mean && invstd && sums, EVK_E_INVALID, "bn_bwd_local_sums: null pointer");
  const int relu = (flags & EVK_BN_RELU) ? (y ? 1 : 2) : 0;
  EVK_REQUIRE(relu != 2 || !d_residual, EVK_E_INVALID, "bn_bwd_local_sums: a residual branch needs y for the ReLU mask");
  int rc = bn_stage_check("bn_bwd_local_sums", rows, C, workspace, workspace_bytes);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const BnPlan pl = bn_plan(rows, C);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(pl.nblk), dim3(256), 0, st, dy, x, y, mean, invstd, gamma, beta,
                     d_residual, partial, rows, C, pl.rows_per_blk, pl.tpc, pl.rl, relu);
  hipLaunchKernelGGL(bn_bwd_sums_final_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, st, (const float*)partial,
                     pl.nblk, C, sums);
  return check_launch("bn_bwd_local_sums");
}

extern "C" int evk_bn_bwd_apply_sums(const float* dy, const float* x, const float* y, const float* gamma,
                                     const float* beta, const float* mean, const float* invstd, const float* mean_g,
                                     const float* mean_gx, float* dx, int64_t rows, int32_t C, uint32_t flags,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(dy && x && mean && invstd && mean_g && mean_gx && dx, EVK_E_INVALID, "bn_bwd_apply_sums: null pointer");
  const int relu = (flags & EVK_BN_RELU) ? (y ? 1 : 2) : 0;
  int rc = bn_stage_check("bn_bwd_apply_sums", rows, C, workspace, workspace_bytes);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  float* coef = (float*)workspace;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, invstd, mean_g, mean_gx, C, coef);
  const size_t n4 = (size_t)rows * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4)), dim3(256), 7 * C * sizeof(float), st, dy, x, y, mean,
                     invstd, (const float*)coef, gamma, beta, dx, n4, C, relu, bn_unroll());
  return check_launch("bn_bwd_apply_sums");
}
