// GroupNorm (+ReLU), channel concat / split and channel-wise dropout scaling on NHWC fp32 for gfx950 — the
// operators FSRelationV2 adds to the FarSeg path (reference ever/module/fs_relation.py:76-163:
// nn.GroupNorm(32, C) in the scene encoders, torch.cat([r * p, o], dim=1), nn.Dropout2d(p=0.1)); GroupNorm on
// full maps is also what the FreeNet-style models of SURVEY §8 f4 need.
//
// GroupNorm statistics are per (sample, group): stage 1 reduces per (sample, row chunk, channel) with the
// BatchNorm-style pivot shift, stage 2 folds the channels of a group in fp64.  Backward mirrors it:
// per-(sample, chunk, channel) sums of dy and dy*xhat, folded per group (for dx) and per channel (dgamma,
// dbeta) in fp64, then one apply pass.  HBM-bound: 3|x| forward, 5|x| backward, 16-byte accesses.
#include "common.hpp"

namespace evk {

constexpr int kGnMaxChunks = 256;

struct GnPlan {
  int nchunk;
  int64_t rows_per_chunk;
  int tpc, rl;
};
static GnPlan gn_plan(int64_t HW, int C) {
  GnPlan p;
  const int c4 = C / 4;
  p.tpc = c4 < 256 ? c4 : 256;
  p.rl = 256 / p.tpc;
  int64_t nb = (HW * (int64_t)C + 65535) / 65536;
  if (nb > kGnMaxChunks) nb = kGnMaxChunks;
  if (nb < 1) nb = 1;
  int64_t rpc = (HW + nb - 1) / nb;
  rpc = ((rpc + p.rl - 1) / p.rl) * p.rl;
  p.rows_per_chunk = rpc;
  p.nchunk = (int)((HW + rpc - 1) / rpc);
  return p;
}

// MODE 0: forward statistics   s = sum (x - pivot), q = sum (x - pivot)^2, pivot = x[n][0][c]
// MODE 1: backward statistics  s = sum g,           q = sum g * xhat      (g = dy masked by y > 0 if relu)
template <int MODE>
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ partial,
                                                         int64_t HW, int C, int G, int64_t rows_per_chunk, int nchunk,
                                                         int tpc, int rl, int relu) {
  __shared__ f32x4 red[2][256];
  const int n = blockIdx.y;
  const int c4 = C >> 2, cg = C / G;
  const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
  const int64_t r1 = min(HW, r0 + rows_per_chunk);
  const float* xn = x + (size_t)n * HW * C;
  const float* dn = MODE ? dy + (size_t)n * HW * C : nullptr;
  const float* yn = (MODE && relu) ? y + (size_t)n * HW * C : nullptr;
  for (int cb = tc; cb < c4; cb += tpc) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    f32x4 pv = {0.f, 0.f, 0.f, 0.f}, mu = pv, is = pv;
    if (MODE == 0) {
      pv = *reinterpret_cast<const f32x4*>(xn + cb * 4);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int g = (cb * 4 + e) / cg;
        mu[e] = mean[n * G + g];
        is[e] = rstd[n * G + g];
      }
    }
    if (tr < rl)
      for (int64_t r = r0 + tr; r < r1; r += rl) {
        const size_t o = (size_t)r * C + cb * 4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xn + o);
        if (MODE == 0) {
          const f32x4 v = xv - pv;
          s += v;
          q += v * v;
        } else {
          f32x4 g = *reinterpret_cast<const f32x4*>(dn + o);
          if (relu) {
            const f32x4 yy = *reinterpret_cast<const f32x4*>(yn + o);
            g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
            g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
          }
          s += g;
          q += g * ((xv - mu) * is);
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
      float* o = partial + ((size_t)n * nchunk + blockIdx.x) * 2 * C;
      *reinterpret_cast<f32x4*>(o + cb * 4) = s;
      *reinterpret_cast<f32x4*>(o + C + cb * 4) = q;
    }
    __syncthreads();
  }
}

// one workgroup per (n, g): fold chunks and the group's channels in fp64
__global__ __launch_bounds__(64) void gn_stats_final_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                            int64_t HW, int C, int G, int nchunk, float eps,
                                                            float* __restrict__ mean, float* __restrict__ rstd) {
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int cg = C / G;
  double sx = 0.0, sxx = 0.0;
  for (int e = threadIdx.x; e < cg; e += 64) {
    const int c = g * cg + e;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < nchunk; ++b) {
      s += (double)partial[((size_t)n * nchunk + b) * 2 * C + c];
      q += (double)partial[((size_t)n * nchunk + b) * 2 * C + C + c];
    }
    const double p = (double)x[(size_t)n * HW * C + c];
    sx += s + (double)HW * p;
    sxx += q + 2.0 * p * s + (double)HW * p * p;
  }
  sx = wave_sum_d(sx);
  sxx = wave_sum_d(sxx);
  if (threadIdx.x == 0) {
    const double cnt = (double)HW * cg;
    const double m = sx / cnt;
    double var = sxx / cnt - m * m;
    if (var < 0.0) var = 0.0;
    mean[blockIdx.x] = (float)m;
    rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y,
                                                       int64_t HW, int C, int G, size_t n4, int relu) {
  const int c4 = C >> 2, cg = C / G;
  const size_t per_n = (size_t)HW * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const int n = (int)(i / per_n);
    const int cb = (int)(i % c4);
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = cb * 4 + e, g = c / cg;
      const float sc = (gamma ? gamma[c] : 1.f) * rstd[n * G + g];
      float r = (v[e] - mean[n * G + g]) * sc + (beta ? beta[c] : 0.f);
      v[e] = relu ? fmaxf(r, 0.f) : r;
    }
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}

// group sums for dx and channel sums for dgamma / dbeta.  coef[n][g] = (m1, m2) = group means of gamma*g and
// gamma*g*xhat.  One workgroup per (n, g) for the coefficients; dgamma/dbeta by a second tiny kernel.
__global__ __launch_bounds__(64) void gn_bwd_group_kernel(const float* __restrict__ partial, const float* __restrict__ gamma,
                                                          int64_t HW, int C, int G, int nchunk,
                                                          float* __restrict__ coef) {
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int cg = C / G;
  double a = 0.0, b = 0.0;
  for (int e = threadIdx.x; e < cg; e += 64) {
    const int c = g * cg + e;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < nchunk; ++k) {
      s += (double)partial[((size_t)n * nchunk + k) * 2 * C + c];
      q += (double)partial[((size_t)n * nchunk + k) * 2 * C + C + c];
    }
    const double gm = gamma ? (double)gamma[c] : 1.0;
    a += gm * s;
    b += gm * q;
  }
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  if (threadIdx.x == 0) {
    const double cnt = (double)HW * cg;
    coef[2 * blockIdx.x] = (float)(a / cnt);
    coef[2 * blockIdx.x + 1] = (float)(b / cnt);
  }
}
__global__ __launch_bounds__(256) void gn_bwd_param_kernel(const float* __restrict__ partial, int N, int C, int nchunk,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < nchunk; ++k) {
      s += (double)partial[((size_t)n * nchunk + k) * 2 * C + c];
      q += (double)partial[((size_t)n * nchunk + k) * 2 * C + C + c];
    }
  if (dbeta) dbeta[c] = (float)s;
  if (dgamma) dgamma[c] = (float)q;
}
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ coef, float* __restrict__ dx,
                                                           int64_t HW, int C, int G, size_t n4, int relu) {
  const int c4 = C >> 2, cg = C / G;
  const size_t per_n = (size_t)HW * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const int n = (int)(i / per_n);
    const int cb = (int)(i % c4);
    f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    if (relu) {
      const f32x4 yy = reinterpret_cast<const f32x4*>(y)[i];
      g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
      g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = cb * 4 + e, gi = n * G + c / cg;
      const float is = rstd[gi];
      const float xh = (xv[e] - mean[gi]) * is;
      o[e] = is * ((gamma ? gamma[c] : 1.f) * g[e] - coef[2 * gi] - xh * coef[2 * gi + 1]);
    }
    reinterpret_cast<f32x4*>(dx)[i] = o;
  }
}

// ---------------------------------------------------------------- concat / split along channels
// out[r][0..Ca) = a[r][:], out[r][Ca..Ca+Cb) = b[r][:]   (Ca, Cb multiples of 4)
__global__ __launch_bounds__(256) void concat2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, size_t rows, int Ca, int Cb) {
  const int ca4 = Ca >> 2, c4 = (Ca + Cb) >> 2;
  const size_t n4 = rows * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const size_t r = i / c4;
    const int cb = (int)(i - r * c4);
    reinterpret_cast<f32x4*>(out)[i] = cb < ca4 ? reinterpret_cast<const f32x4*>(a)[r * ca4 + cb]
                                                : reinterpret_cast<const f32x4*>(b)[r * (c4 - ca4) + (cb - ca4)];
  }
}
__global__ __launch_bounds__(256) void split2_kernel(const float* __restrict__ src, float* __restrict__ a,
                                                     float* __restrict__ b, size_t rows, int Ca, int Cb) {
  const int ca4 = Ca >> 2, c4 = (Ca + Cb) >> 2;
  const size_t n4 = rows * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const size_t r = i / c4;
    const int cb = (int)(i - r * c4);
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    if (cb < ca4) {
      if (a) reinterpret_cast<f32x4*>(a)[r * ca4 + cb] = v;
    } else if (b) {
      reinterpret_cast<f32x4*>(b)[r * (c4 - ca4) + (cb - ca4)] = v;
    }
  }
}
// y[n][hw][c] = x * scale[n][c]   (Dropout2d with a precomputed keep-mask / (1-p); its own adjoint)
__global__ __launch_bounds__(256) void channel_scale_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                            float* __restrict__ y, int64_t HW, int C, size_t n4) {
  const int c4 = C >> 2;
  const size_t per_n = (size_t)HW * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const int n = (int)(i / per_n);
    const int cb = (int)(i % c4);
    reinterpret_cast<f32x4*>(y)[i] =
        reinterpret_cast<const f32x4*>(x)[i] * *reinterpret_cast<const f32x4*>(scale + (size_t)n * C + cb * 4);
  }
}

static int stream_blocks(size_t n4) {
  size_t b = (n4 + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace evk

using namespace evk;

extern "C" size_t evk_gn_workspace_bytes(int32_t N, int64_t HW, int32_t C, int32_t G) {
  const GnPlan pl = gn_plan(HW, C);
  return ((size_t)N * pl.nchunk * 2 * C + (size_t)2 * N * G) * sizeof(float);
}

static int gn_check(const char* what, int32_t N, int64_t HW, int32_t C, int32_t G) {
  EVK_REQUIRE(N > 0 && HW > 0 && C > 0 && G > 0, EVK_E_INVALID, "%s: non-positive dimension", what);
  EVK_REQUIRE(C % 4 == 0 && C % G == 0, EVK_E_UNSUPPORTED, "%s: C=%d must be a multiple of 4 and of G=%d", what, C, G);
  EVK_REQUIRE(N <= 65535, EVK_E_UNSUPPORTED, "%s: batch %d > 65535", what, N);
  return EVK_OK;
}

extern "C" int evk_gn_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y, float* save_mean,
                          float* save_rstd, int32_t N, int64_t HW, int32_t C, int32_t G, uint32_t flags,
                          void* workspace, size_t workspace_bytes, void* stream) {
  int rc = gn_check("gn_fwd", N, HW, C, G);
  if (rc) return rc;
  EVK_REQUIRE(x && y && save_mean && save_rstd, EVK_E_INVALID, "gn_fwd: null pointer");
  EVK_REQUIRE(workspace && workspace_bytes >= evk_gn_workspace_bytes(N, HW, C, G), EVK_E_WORKSPACE, "gn_fwd: workspace");
  hipStream_t st = (hipStream_t)stream;
  const GnPlan pl = gn_plan(HW, C);
  float* partial = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(gn_partial_kernel<0>, dim3(pl.nchunk, N), dim3(256), 0, st, x, nullptr, nullptr, nullptr, nullptr,
                     partial, HW, C, G, pl.rows_per_chunk, pl.nchunk, pl.tpc, pl.rl, 0);
  hipLaunchKernelGGL(gn_stats_final_kernel, dim3(N * G), dim3(64), 0, st, x, partial, HW, C, G, pl.nchunk, eps, save_mean,
                     save_rstd);
  const size_t n4 = (size_t)N * HW * C / 4;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(stream_blocks(n4)), dim3(256), 0, st, x, save_mean, save_rstd, gamma, beta, y,
                     HW, C, G, n4, (flags & 1u) ? 1 : 0);
  return check_launch("gn_fwd");
}

extern "C" int evk_gn_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* save_mean,
                          const float* save_rstd, float* dx, float* dgamma, float* dbeta, int32_t N, int64_t HW,
                          int32_t C, int32_t G, uint32_t flags, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = gn_check("gn_bwd", N, HW, C, G);
  if (rc) return rc;
  const int relu = (flags & 1u) ? 1 : 0;
  EVK_REQUIRE(dy && x && save_mean && save_rstd && dx && (!relu || y), EVK_E_INVALID, "gn_bwd: null pointer");
  EVK_REQUIRE(workspace && workspace_bytes >= evk_gn_workspace_bytes(N, HW, C, G), EVK_E_WORKSPACE, "gn_bwd: workspace");
  hipStream_t st = (hipStream_t)stream;
  const GnPlan pl = gn_plan(HW, C);
  float* partial = reinterpret_cast<float*>(workspace);
  float* coef = partial + (size_t)N * pl.nchunk * 2 * C;
  hipLaunchKernelGGL(gn_partial_kernel<1>, dim3(pl.nchunk, N), dim3(256), 0, st, x, dy, y, save_mean, save_rstd, partial,
                     HW, C, G, pl.rows_per_chunk, pl.nchunk, pl.tpc, pl.rl, relu);
  hipLaunchKernelGGL(gn_bwd_group_kernel, dim3(N * G), dim3(64), 0, st, (const float*)partial, gamma, HW, C, G, pl.nchunk,
                     coef);
  if (dgamma || dbeta)
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const float*)partial, N, C,
                       pl.nchunk, dgamma, dbeta);
  const size_t n4 = (size_t)N * HW * C / 4;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(stream_blocks(n4)), dim3(256), 0, st, dy, x, y, save_mean, save_rstd, gamma,
                     (const float*)coef, dx, HW, C, G, n4, relu);
  return check_launch("gn_bwd");
}

extern "C" int evk_concat_channels(const float* a, const float* b, float* out, int64_t rows, int32_t Ca, int32_t Cb,
                                   void* stream) {
  EVK_REQUIRE(a && b && out, EVK_E_INVALID, "concat_channels: null pointer");
  EVK_REQUIRE(Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0, EVK_E_UNSUPPORTED,
              "concat_channels: channel counts (%d, %d) must be positive multiples of 4", Ca, Cb);
  const size_t n4 = (size_t)rows * (Ca + Cb) / 4;
  hipLaunchKernelGGL(concat2_kernel, dim3(stream_blocks(n4)), dim3(256), 0, (hipStream_t)stream, a, b, out, (size_t)rows,
                     Ca, Cb);
  return check_launch("concat_channels");
}

extern "C" int evk_split_channels(const float* src, float* a, float* b, int64_t rows, int32_t Ca, int32_t Cb,
                                  void* stream) {
  EVK_REQUIRE(src && (a || b), EVK_E_INVALID, "split_channels: null pointer");
  EVK_REQUIRE(Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0, EVK_E_UNSUPPORTED,
              "split_channels: channel counts (%d, %d) must be positive multiples of 4", Ca, Cb);
  const size_t n4 = (size_t)rows * (Ca + Cb) / 4;
  hipLaunchKernelGGL(split2_kernel, dim3(stream_blocks(n4)), dim3(256), 0, (hipStream_t)stream, src, a, b, (size_t)rows,
                     Ca, Cb);
  return check_launch("split_channels");
}

extern "C" int evk_channel_scale(const float* x, const float* scale, float* y, int32_t N, int64_t HW, int32_t C,
                                 void* stream) {
  EVK_REQUIRE(x && scale && y, EVK_E_INVALID, "channel_scale: null pointer");
  EVK_REQUIRE(C > 0 && C % 4 == 0, EVK_E_UNSUPPORTED, "channel_scale: C=%d must be a multiple of 4", C);
  const size_t n4 = (size_t)N * HW * C / 4;
  hipLaunchKernelGGL(channel_scale_kernel, dim3(stream_blocks(n4)), dim3(256), 0, (hipStream_t)stream, x, scale, y, HW, C,
                     n4);
  return check_launch("channel_scale");
}
