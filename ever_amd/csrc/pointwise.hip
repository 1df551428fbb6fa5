// HBM-bound wavefront kernels of the EVer hot path on NHWC fp32 (gfx950): ReLU / add / scale,
// channel padding and NCHW<->NHWC at the model boundary, MaxPool 3x3 s2, nearest x2 + lateral add,
// bilinear align_corners=True resampling and its adjoint, global average pool, FS-Relation, 4-way mean.
// Reference call sites are cited per entry point in include/ever_hip.h.
// All kernels use 16-byte accesses along the channel axis (C % 4 == 0) with a scalar fallback where
// the path has C == 1 (classifier logits).
#include <stdlib.h>
#include "common.hpp"

namespace evk {

// max|v| of what a workgroup wrote into slot (workgroup & 63) of the output's operand-scale buffer (64 slots 32 words
// apart, zero on entry; include/ever_hip.h: evk_absmax) — one atomic per wave, as the BatchNorm and convolution epilogues
__device__ __forceinline__ uint32_t abs4_bits(uint32_t m, const f32x4 v) {
  // by value into floats first: __builtin_bit_cast on a vector-element lvalue (v.y) reads element 0 with this compiler
  const float x = v.x, y = v.y, z = v.z, w = v.w;
  m = max(m, __float_as_uint(x) & 0x7fffffffu); m = max(m, __float_as_uint(y) & 0x7fffffffu);
  m = max(m, __float_as_uint(z) & 0x7fffffffu); m = max(m, __float_as_uint(w) & 0x7fffffffu);
  return m;
}
__device__ __forceinline__ void commit_absmax(uint32_t* __restrict__ slots, uint32_t m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(&slots[(blockIdx.x & 63) * 32], m);
}

// One element per thread (the grid-stride loops below then run once): resident workgroups sweep one contiguous
// window of HBM in dispatch order, 6.1 TB/s for 1R+1W on 268 MB against 5.1 for 4096 grid-striding workgroups
// (tools/probes/copy_patterns.hip).
static inline int grid_for(size_t n, int per_block = 256, int cap = 1 << 24) {
  size_t b = (n + per_block - 1) / per_block;
  return (int)(b > (size_t)cap ? cap : (b < 1 ? 1 : b));
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_d(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * expf(-0.5f * x * x);
}

template <int OP>  // 0 relu, 1 relu_bwd, 2 add, 3 scale, 4 a*b*alpha, 5 gelu, 6 gelu_bwd (a = dy, b = x)
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ o, size_t n, float alpha) {
  const size_t n4 = n >> 2;
  const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
  const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
  f32x4* o4 = reinterpret_cast<f32x4*>(o);
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 v = a4[i];
    if (OP == 0) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    } else if (OP == 1) {  // a = dy, b = y
      const f32x4 y = b4[i];
      v.x = y.x > 0.f ? v.x : 0.f; v.y = y.y > 0.f ? v.y : 0.f;
      v.z = y.z > 0.f ? v.z : 0.f; v.w = y.w > 0.f ? v.w : 0.f;
    } else if (OP == 2) {
      v += b4[i];
    } else if (OP == 3) {
      v *= alpha;
    } else if (OP == 4) {
      v = v * b4[i] * alpha;
    } else if (OP == 5) {
      v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w);
    } else {
      const f32x4 x = b4[i];
      v.x *= gelu_d(x.x); v.y *= gelu_d(x.y); v.z *= gelu_d(x.z); v.w *= gelu_d(x.w);
    }
    o4[i] = v;
  }
  // tail
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float v = a[i];
    if (OP == 0) v = fmaxf(v, 0.f);
    else if (OP == 1) v = b[i] > 0.f ? v : 0.f;
    else if (OP == 2) v += b[i];
    else if (OP == 3) v *= alpha;
    else if (OP == 4) v = v * b[i] * alpha;
    else if (OP == 5) v = gelu_f(v);
    else v *= gelu_d(b[i]);
    o[i] = v;
  }
}

__global__ __launch_bounds__(256) void mean4_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    const float* __restrict__ c, const float* __restrict__ d,
                                                    float* __restrict__ o, size_t n4, uint32_t* __restrict__ amax) {
  const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
  const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
  const f32x4* c4 = reinterpret_cast<const f32x4*>(c);
  const f32x4* d4 = reinterpret_cast<const f32x4*>(d);
  f32x4* o4 = reinterpret_cast<f32x4*>(o);
  uint32_t m = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = (((a4[i] + b4[i]) + c4[i]) + d4[i]) * 0.25f;  // same association as python sum(list)/4
    o4[i] = v;
    m = abs4_bits(m, v);
  }
  if (amax) commit_absmax(amax, m);
}

// ------------------------------------------------------------------------------------------------
__global__ void pad_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int C,
                                    int Cp) {
  const size_t total = rows * (size_t)Cp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / Cp;
    const int c = (int)(i - r * Cp);
    dst[i] = c < C ? src[r * C + c] : 0.f;
  }
}
__global__ void unpad_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int Cp,
                                      int C) {
  const size_t total = rows * (size_t)C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / C;
    const int c = (int)(i - r * C);
    dst[i] = src[r * Cp + c];
  }
}
// one thread per pixel: plane reads are coalesced across lanes, the Cp-wide pixel is written whole
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, size_t HW,
                                    int Cp) {
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / HW, p = i - n * HW;
    const float* s = src + n * C * HW + p;
    float* d = dst + i * Cp;
    for (int c = 0; c < Cp; ++c) d[c] = c < C ? s[(size_t)c * HW] : 0.f;
  }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, size_t HW,
                                    int Cp) {
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / HW, p = i - n * HW;
    const float* s = src + i * Cp;
    float* d = dst + n * C * HW + p;
    for (int c = 0; c < C; ++c) d[(size_t)c * HW] = s[c];
  }
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(3, 2, 1).  code[..] = winning tap ky*3+kx (first maximum in scan order, NaN propagates,
// matching aten's max_pool2d_with_indices).
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ code, int N, int H, int W, int C,
                                                          int Ho, int Wo) {
  const int c4 = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cb = (int)(i % c4);
    size_t pix = i / c4;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bx = 0, by = 0, bz = 0, bw = 0;
    bool first = true;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + iy) * W + ix) * C + cb * 4);
        const int t = ky * 3 + kx;
        if (first) {
          best = v; bx = by = bz = bw = t; first = false;
        } else {
          if (v.x > best.x || v.x != v.x) { best.x = v.x; bx = t; }
          if (v.y > best.y || v.y != v.y) { best.y = v.y; by = t; }
          if (v.z > best.z || v.z != v.z) { best.z = v.z; bz = t; }
          if (v.w > best.w || v.w != v.w) { best.w = v.w; bw = t; }
        }
      }
    }
    const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + cb * 4;
    *reinterpret_cast<f32x4*>(y + o) = best;
    *reinterpret_cast<uint32_t*>(code + o) = (uint32_t)bx | ((uint32_t)by << 8) | ((uint32_t)bz << 16) | ((uint32_t)bw << 24);
  }
}
// gather-form adjoint: input pixel (iy,ix) is tap (iy-2oy+1, ix-2ox+1) of at most 2x2 windows
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy,
                                                          const uint8_t* __restrict__ code, float* __restrict__ dx,
                                                          int N, int H, int W, int C, int Ho, int Wo) {
  const int c4 = C >> 2;
  const size_t total = (size_t)N * H * W * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cb = (int)(i % c4);
    size_t pix = i / c4;
    const int ix = (int)(pix % W);
    pix /= W;
    const int iy = (int)(pix % H);
    const int n = (int)(pix / H);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    const int oy_lo = max(0, iy >> 1), oy_hi = min(Ho - 1, (iy + 1) >> 1);
    const int ox_lo = max(0, ix >> 1), ox_hi = min(Wo - 1, (ix + 1) >> 1);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const int ky = iy - 2 * oy + 1;
      if ((unsigned)ky > 2u) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const int kx = ix - 2 * ox + 1;
        if ((unsigned)kx > 2u) continue;
        const uint32_t t = (uint32_t)(ky * 3 + kx);
        const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + cb * 4;
        const uint32_t cd = *reinterpret_cast<const uint32_t*>(code + o);
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
        if ((cd & 0xff) == t) g.x += d.x;
        if (((cd >> 8) & 0xff) == t) g.y += d.y;
        if (((cd >> 16) & 0xff) == t) g.z += d.z;
        if ((cd >> 24) == t) g.w += d.w;
      }
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = g;
  }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nearest2x_add_kernel(const float* __restrict__ top,
                                                            const float* __restrict__ lateral,
                                                            float* __restrict__ out, int N, int H, int W, int C,
                                                            uint32_t* __restrict__ amax) {
  const int c4 = C >> 2;
  const int Ht = H >> 1, Wt = W >> 1;
  const size_t total = (size_t)N * H * W * c4;
  uint32_t m = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cb = (int)(i % c4);
    size_t pix = i / c4;
    const int x = (int)(pix % W);
    pix /= W;
    const int y = (int)(pix % H);
    const int n = (int)(pix / H);
    const f32x4 t = *reinterpret_cast<const f32x4*>(top + (((size_t)n * Ht + (y >> 1)) * Wt + (x >> 1)) * C + cb * 4);
    const f32x4 l = *reinterpret_cast<const f32x4*>(lateral + i * 4);
    const f32x4 v = l + t;
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
    m = abs4_bits(m, v);
  }
  if (amax) commit_absmax(amax, m);
}
__global__ __launch_bounds__(256) void nearest2x_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dtop,
                                                            int N, int H, int W, int C) {
  // H, W are the fine (dout) dims
  const int c4 = C >> 2;
  const int Ht = H >> 1, Wt = W >> 1;
  const size_t total = (size_t)N * Ht * Wt * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cb = (int)(i % c4);
    size_t pix = i / c4;
    const int x = (int)(pix % Wt);
    pix /= Wt;
    const int y = (int)(pix % Ht);
    const int n = (int)(pix / Ht);
    const float* b = dout + (((size_t)n * H + 2 * y) * W + 2 * x) * C + cb * 4;
    const f32x4 s = (*reinterpret_cast<const f32x4*>(b) + *reinterpret_cast<const f32x4*>(b + C)) +
                    (*reinterpret_cast<const f32x4*>(b + (size_t)W * C) +
                     *reinterpret_cast<const f32x4*>(b + (size_t)W * C + C));
    *reinterpret_cast<f32x4*>(dtop + i * 4) = s;
  }
}

// F.max_pool2d(x, 1, 2, 0) — a window of ONE pixel, stride 2: y[n, yo, xo, :] = x[n, 2 yo, 2 xo, :] (reference fpn.py:118-120,
// LastLevelMaxPool).  ADJ = the adjoint: dx is dy at the even pixels and zero elsewhere (one thread per 16 bytes of dx).
template <bool ADJ>
__global__ __launch_bounds__(256) void subsample2_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H,
                                                         int W, int C, int Ho, int Wo) {
  const int c4 = C >> 2;
  const size_t total = ADJ ? (size_t)N * H * W * c4 : (size_t)N * Ho * Wo * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cb = (int)(i % c4);
    size_t pix = i / c4;
    const int wd = ADJ ? W : Wo, hd = ADJ ? H : Ho;
    const int x = (int)(pix % wd);
    pix /= wd;
    const int y = (int)(pix % hd);
    const int n = (int)(pix / hd);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!ADJ)
      v = *reinterpret_cast<const f32x4*>(src + (((size_t)n * H + 2 * y) * W + 2 * x) * C + cb * 4);
    else if (!(x & 1) && !(y & 1))
      v = *reinterpret_cast<const f32x4*>(src + (((size_t)n * Ho + (y >> 1)) * Wo + (x >> 1)) * C + cb * 4);
    *reinterpret_cast<f32x4*>(dst + i * 4) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Bilinear, align_corners=True (nn.UpsamplingBilinear2d).  Index math follows aten's
// upsample_bilinear2d: scale = (in-1)/(out-1) in float, src = scale*dst, i0 = (int)src,
// i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1.
__device__ __forceinline__ void bl_coord(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
  const float s = scale * (float)o;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

template <int VEC>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                           int Hi, int Wi, int Ho, int Wo, int C, float sy, float sx) {
  const int cv = C / VEC;
  const size_t total = (size_t)N * Ho * Wo * cv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cb = (int)(i % cv);
    size_t pix = i / cv;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    bl_coord(oy, sy, Hi, y0, y1, hy0, hy1);
    bl_coord(ox, sx, Wi, x0, x1, wx0, wx1);
    const float* b = x + (size_t)n * Hi * Wi * C + cb * VEC;
    const size_t o00 = ((size_t)y0 * Wi + x0) * C, o01 = ((size_t)y0 * Wi + x1) * C;
    const size_t o10 = ((size_t)y1 * Wi + x0) * C, o11 = ((size_t)y1 * Wi + x1) * C;
    if (VEC == 4) {
      const f32x4 v00 = *reinterpret_cast<const f32x4*>(b + o00), v01 = *reinterpret_cast<const f32x4*>(b + o01);
      const f32x4 v10 = *reinterpret_cast<const f32x4*>(b + o10), v11 = *reinterpret_cast<const f32x4*>(b + o11);
      *reinterpret_cast<f32x4*>(y + i * 4) = hy0 * (wx0 * v00 + wx1 * v01) + hy1 * (wx0 * v10 + wx1 * v11);
    } else {
      y[i] = hy0 * (wx0 * b[o00] + wx1 * b[o01]) + hy1 * (wx0 * b[o10] + wx1 * b[o11]);
    }
  }
}

// adjoint in gather form: every input pixel sums the (few) output pixels whose 2x2 footprint
// contains it, with exactly the forward's weights.  Candidate range per axis:
// src in (i-1, i+1)  <=>  o in ((i-1)/scale, (i+1)/scale), widened by one for float rounding.
template <int VEC>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N,
                                                           int Hi, int Wi, int Ho, int Wo, int C, float sy, float sx,
                                                           float isy, float isx) {
  const int cv = C / VEC;
  const size_t total = (size_t)N * Hi * Wi * cv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cb = (int)(i % cv);
    size_t pix = i / cv;
    const int ix = (int)(pix % Wi);
    pix /= Wi;
    const int iy = (int)(pix % Hi);
    const int n = (int)(pix / Hi);
    int oy_lo = (int)floorf((float)(iy - 1) * isy) - 1, oy_hi = (int)ceilf((float)(iy + 1) * isy) + 1;
    int ox_lo = (int)floorf((float)(ix - 1) * isx) - 1, ox_hi = (int)ceilf((float)(ix + 1) * isx) + 1;
    oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, Ho - 1);
    ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, Wo - 1);
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    float acc1 = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float hy0, hy1;
      bl_coord(oy, sy, Hi, y0, y1, hy0, hy1);
      float wy = 0.f;
      if (y0 == iy) wy += hy0;
      if (y1 == iy) wy += hy1;
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float wx0, wx1;
        bl_coord(ox, sx, Wi, x0, x1, wx0, wx1);
        float wx = 0.f;
        if (x0 == ix) wx += wx0;
        if (x1 == ix) wx += wx1;
        if (wx == 0.f) continue;
        const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + cb * VEC;
        if (VEC == 4) acc4 += (wy * wx) * *reinterpret_cast<const f32x4*>(dy + o);
        else acc1 += (wy * wx) * dy[o];
      }
    }
    if (VEC == 4) *reinterpret_cast<f32x4*>(dx + i * 4) = acc4;
    else dx[i] = acc1;
  }
}

// Wide-channel forms (C >= 128): ONE WAVE PER PIXEL, lanes across the 16-byte channel chunks, so that the pixel's
// index arithmetic is wave-uniform.  The element-per-thread kernels above spend ~100 (forward) to ~1000 (backward)
// VALU instructions per 16 bytes on it and are VALU-bound on the 256-channel decoder maps (2.9 TB/s).  Uniform
// arithmetic lands on the scalar unit, ONE per CU: three integer divisions per pixel there were just as slow
// (124 us for 335 MB), hence the multiply-shift divisions and, in the backward, the candidate weights computed
// across lanes (one VALU pass for all 12 candidates of an axis) and fetched by v_readlane.
// Forward: a workgroup owns kTileR x kTileC output pixels, stages the input patch they read in LDS (one global load
// per input pixel and workgroup) and interpolates out of LDS.  Four global loads per output pixel — even wave-uniform
// and L2-resident — made the kernel L2->L1 bound: 90 us for the 64^2 -> 128^2 x 256 maps against 38 us for its
// 268 MB of stores alone (tools/probes/upsample_probe.hip).
constexpr int kTileR = 4, kTileC = 16;
// LPP lanes per pixel: 64 (C > 128), or 32 — the decoder's 128-channel maps fill only half a wave per pixel, so a wave
// takes two pixels side by side there (the per-pixel arithmetic is then per half-wave, on the vector unit)
template <int LPP>
__global__ __launch_bounds__(256) void bilinear_fwd_tile_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                int Hi, int Wi, int Ho, int Wo, int C, float sy,
                                                                float sx, int patch_cols) {
  extern __shared__ __attribute__((aligned(16))) float patch[];  // [rows][patch_cols][C]
  constexpr int PPW = 64 / LPP;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int sub = lane / LPP, cl = (lane % LPP) * 4;
  const int n = blockIdx.z, oy0 = blockIdx.y * kTileR, ox0 = blockIdx.x * kTileC;
  const int oy_last = min(oy0 + kTileR, Ho) - 1, ox_last = min(ox0 + kTileC, Wo) - 1;
  int ylo, yhi, xlo, xhi, t0;
  float f0, f1;
  bl_coord(oy0, sy, Hi, ylo, t0, f0, f1);
  bl_coord(oy_last, sy, Hi, t0, yhi, f0, f1);
  bl_coord(ox0, sx, Wi, xlo, t0, f0, f1);
  bl_coord(ox_last, sx, Wi, t0, xhi, f0, f1);
  const int pr = yhi - ylo + 1, pc = xhi - xlo + 1;  // pc <= patch_cols by the host's bound
  const float* b = x + (size_t)n * Hi * Wi * C;
  for (int q = wave * PPW + sub; q < pr * pc; q += 4 * PPW) {
    const int r = q / pc, c = q - r * pc;
    const float* src = b + ((size_t)(ylo + r) * Wi + (xlo + c)) * C;
    float* dst = patch + (size_t)(r * patch_cols + c) * C;
    for (int k = cl; k < C; k += LPP * 4) *reinterpret_cast<f32x4*>(dst + k) = *reinterpret_cast<const f32x4*>(src + k);
  }
  __syncthreads();
  // wave w: output row oy0 + w (kTileR == 4 waves), all kTileC columns
  const int oy = oy0 + wave;
  if (oy > oy_last) return;
  int y0, y1;
  float hy0, hy1;
  bl_coord(oy, sy, Hi, y0, y1, hy0, hy1);
  const float* r0 = patch + (size_t)(y0 - ylo) * patch_cols * C;
  const float* r1 = patch + (size_t)(y1 - ylo) * patch_cols * C;
  float* o = y + (((size_t)n * Ho + oy) * Wo + ox0) * C;
  for (int j = sub; j <= ox_last - ox0; j += PPW) {
    int x0, x1;
    float wx0, wx1;
    bl_coord(ox0 + j, sx, Wi, x0, x1, wx0, wx1);
    const int c0 = (x0 - xlo) * C, c1 = (x1 - xlo) * C;
    for (int k = cl; k < C; k += LPP * 4) {
      const f32x4 v00 = *reinterpret_cast<const f32x4*>(r0 + c0 + k), v01 = *reinterpret_cast<const f32x4*>(r0 + c1 + k);
      const f32x4 v10 = *reinterpret_cast<const f32x4*>(r1 + c0 + k), v11 = *reinterpret_cast<const f32x4*>(r1 + c1 + k);
      *reinterpret_cast<f32x4*>(o + (size_t)j * C + k) = hy0 * (wx0 * v00 + wx1 * v01) + hy1 * (wx0 * v10 + wx1 * v11);
    }
  }
}

// weight with which output coordinate o (one axis) reads input coordinate i; 0 when it does not, or o is out of range
__device__ __forceinline__ float bl_adjoint_weight(int o, int o_hi, float scale, int in, int i) {
  int i0, i1;
  float l0, l1;
  bl_coord(o, scale, in, i0, i1, l0, l1);
  float w = 0.f;
  if (i0 == i) w += l0;
  if (i1 == i) w += l1;
  return o <= o_hi ? w : 0.f;
}

__global__ __launch_bounds__(256) void bilinear_bwd_wave_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                int npix, FastDiv fdWi, FastDiv fdHi, int Ho, int Wo,
                                                                int C, float sy, float sx, float isy, float isx) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int pix = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;   // see bilinear_fwd_wave_kernel
  if (pix >= npix) return;
  const int Hi = (int)fdHi.div, Wi = (int)fdWi.div;
  const uint32_t t = fdiv((uint32_t)pix, fdWi), ix = (uint32_t)pix - t * fdWi.div;
  const uint32_t n = fdiv(t, fdHi), iy = t - n * fdHi.div;
  int oy_lo = (int)floorf((float)((int)iy - 1) * isy) - 1, oy_hi = (int)ceilf((float)((int)iy + 1) * isy) + 1;
  int ox_lo = (int)floorf((float)((int)ix - 1) * isx) - 1, ox_hi = (int)ceilf((float)((int)ix + 1) * isx) + 1;
  oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, Ho - 1);
  ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, Wo - 1);
  // lane j (mod 16) holds the weight of candidate lo + j on each axis; the candidate range of this kernel's callers
  // (scale >= 2 up-sampling) is at most 12 wide, wider ranges take the element-per-thread kernel
  const int j = lane & 15;
  const float wyv = bl_adjoint_weight(oy_lo + j, oy_hi, sy, Hi, (int)iy);
  const float wxv = bl_adjoint_weight(ox_lo + j, ox_hi, sx, Wi, (int)ix);
  uint32_t my = (uint32_t)(__ballot(wyv != 0.f) & 0xffffull);
  const uint32_t mx = (uint32_t)(__ballot(wxv != 0.f) & 0xffffull);
  const float* b = dy + (size_t)n * Ho * Wo * C;
  float* o = dx + (size_t)pix * C;
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // same candidate order (oy ascending, then ox ascending) and weight arithmetic as bilinear_bwd_kernel
  while (my) {
    const int jy = __builtin_ctz(my);
    my &= my - 1;
    const float wy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wyv), jy));
    const float* row = b + (size_t)(oy_lo + jy) * Wo * C;
    uint32_t m = mx;
    while (m) {
      const int jx = __builtin_ctz(m);
      m &= m - 1;
      const float wx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wxv), jx));
      const float w = wy * wx;
      const float* src = row + (size_t)(ox_lo + jx) * C + lane * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (lane * 4 + 256 * k < C) acc[k] += w * *reinterpret_cast<const f32x4*>(src + 256 * k);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (lane * 4 + 256 * k < C) *reinterpret_cast<f32x4*>(o + lane * 4 + 256 * k) = acc[k];
}

// ------------------------------------------------------------------------------------------------
// Global average pool: x [N][HW][C] -> y [N][C].  grid (ceil(c4/tpc), N)
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C,
                                                      int tpc, int rl, float inv) {
  __shared__ f32x4 red[256];
  const int c4 = C >> 2;
  const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
  const int cb = blockIdx.x * tpc + tc;
  const int n = blockIdx.y;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (cb < c4 && tr < rl)
    for (int p = tr; p < HW; p += rl) s += *reinterpret_cast<const f32x4*>(x + ((size_t)n * HW + p) * C + cb * 4);
  red[threadIdx.x] = s;
  __syncthreads();
  if (tr == 0 && cb < c4) {
    for (int k = 1; k < rl; ++k) s += red[k * tpc + tc];
    *reinterpret_cast<f32x4*>(y + (size_t)n * C + cb * 4) = s * inv;
  }
}
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N,
                                                      int HW, int C, float inv) {
  const int c4 = C >> 2;
  const size_t total = (size_t)N * HW * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cb = (int)(i % c4);
    const size_t n = i / ((size_t)HW * c4);
    *reinterpret_cast<f32x4*>(dx + i * 4) = *reinterpret_cast<const f32x4*>(dy + n * C + cb * 4) * inv;
  }
}

// ------------------------------------------------------------------------------------------------
// FS-Relation: one wave per pixel; lanes stride the channel axis in 16-byte chunks.
__global__ __launch_bounds__(256) void relation_fwd_kernel(const float* __restrict__ scene,
                                                           const float* __restrict__ content,
                                                           const float* __restrict__ feat, float* __restrict__ out,
                                                           float* __restrict__ r, int N, int HW, int C) {
  const int c4 = C >> 2;
  const int lane = threadIdx.x & 63;
  const size_t npix = (size_t)N * HW;
  const size_t wstride = (size_t)gridDim.x * 4;
  for (size_t pix = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += wstride) {
    const size_t n = pix / HW;
    const float* sc = scene + n * C;
    const float* ct = content + pix * C;
    float dot = 0.f;
    for (int cb = lane; cb < c4; cb += 64) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(sc + cb * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(ct + cb * 4);
      dot += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    dot = wave_sum(dot);
    const float rv = 1.f / (1.f + expf(-dot));
    if (lane == 0) r[pix] = rv;
    const float* ft = feat + pix * C;
    float* o = out + pix * C;
    for (int cb = lane; cb < c4; cb += 64)
      *reinterpret_cast<f32x4*>(o + cb * 4) = *reinterpret_cast<const f32x4*>(ft + cb * 4) * rv;
  }
}
// backward: dfeat = r*dout ; dz = (sum_c dout*feat) * r*(1-r) ; dcontent = dz*scene ;
// partial dscene[blk][n?]: each workgroup owns a pixel range inside ONE image n and accumulates
// sum_p dz*content into partial[n][blk][C].
__global__ __launch_bounds__(256) void relation_bwd_kernel(const float* __restrict__ dout,
                                                           const float* __restrict__ scene,
                                                           const float* __restrict__ content,
                                                           const float* __restrict__ feat, const float* __restrict__ r,
                                                           float* __restrict__ dcontent, float* __restrict__ dfeat,
                                                           float* __restrict__ partial, int HW, int C, int pix_per_blk,
                                                           int nblk) {
  extern __shared__ __attribute__((aligned(16))) float sacc[];  // [4 waves][C]
  const int c4 = C >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * pix_per_blk, p1 = min(HW, p0 + pix_per_blk);
  const float* sc = scene + (size_t)n * C;
  float* myacc = sacc + wave * C;
  for (int c = lane; c < C; c += 64) myacc[c] = 0.f;
  for (int p = p0 + wave; p < p1; p += 4) {
    const size_t pix = (size_t)n * HW + p;
    const float rv = r[pix];
    const float* d_o = dout + pix * C;
    const float* ft = feat + pix * C;
    const float* ct = content + pix * C;
    float dot = 0.f;
    for (int cb = lane; cb < c4; cb += 64) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(d_o + cb * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(ft + cb * 4);
      dot += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
      *reinterpret_cast<f32x4*>(dfeat + pix * C + cb * 4) = a * rv;
    }
    dot = wave_sum(dot);
    const float dz = dot * rv * (1.f - rv);
    for (int cb = lane; cb < c4; cb += 64) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc + cb * 4);
      const f32x4 c4v = *reinterpret_cast<const f32x4*>(ct + cb * 4);
      *reinterpret_cast<f32x4*>(dcontent + pix * C + cb * 4) = s4 * dz;
      f32x4* a = reinterpret_cast<f32x4*>(myacc + cb * 4);
      *a = *a + c4v * dz;
    }
  }
  __syncthreads();
  float* o = partial + ((size_t)n * nblk + blockIdx.x) * C;
  for (int c = threadIdx.x; c < C; c += 256) o[c] = (sacc[c] + sacc[C + c]) + (sacc[2 * C + c] + sacc[3 * C + c]);
}
// 8 channels x 32 partial-lanes per workgroup (fixed fold order), as the other finalisations: a serial walk over up to
// 256 partials per thread was 32 us of latency per call
__global__ __launch_bounds__(256) void relation_dscene_final_kernel(const float* __restrict__ partial,
                                                                    float* __restrict__ dscene, int nblk, int C) {
  __shared__ double red[32][8];
  const int tc = threadIdx.x & 7, tl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + tc;
  const int n = blockIdx.y;
  double s = 0.0;
  if (c < C)
    for (int b = tl; b < nblk; b += 32) s += (double)partial[((size_t)n * nblk + b) * C + c];
  red[tl][tc] = s;
  __syncthreads();
  if (tl == 0 && c < C) {
    for (int k = 1; k < 32; ++k) s += red[k][tc];
    dscene[(size_t)n * C + c] = (float)s;
  }
}

// ---- FS-Relation with the two BatchNorm + ReLU passes inside (reference fs_relation.py:39-53,61-71: content / re-encoding =
// ReLU(BN(conv1x1(p))), out = sigmoid(<scene, content>) * re-encoded).  The separate passes wrote the two normalised
// 256-channel maps only for this kernel to read them back, and their backward re-read both pairs (g, z) just to form the
// per-channel sums: here the forward reads the convolution outputs z and applies scale / shift / ReLU on the fly, the
// backward rebuilds both activations from z, writes the MASKED gradients g and leaves the BatchNorm partial sums
// (sum g, sum g * xhat, max|g|, max|xhat| per workgroup and channel) for evk_bn_bwd_from_partials.
__device__ __forceinline__ f32x4 bn_relu4(const f32x4 z, const f32x4 sc, const f32x4 sh) {
  f32x4 y = z * sc + sh;
  y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
  return y;
}
__global__ __launch_bounds__(256) void relation_bn_fwd_kernel(const float* __restrict__ scene, const float* __restrict__ zc,
                                                              const float* __restrict__ ssc, const float* __restrict__ zf,
                                                              const float* __restrict__ ssf, float* __restrict__ out,
                                                              float* __restrict__ r, int N, int HW, int C,
                                                              uint32_t* __restrict__ amax) {
  const int c4 = C >> 2;
  const int lane = threadIdx.x & 63;
  const size_t npix = (size_t)N * HW;
  const size_t wstride = (size_t)gridDim.x * 4;
  uint32_t m = 0;
  for (size_t pix = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += wstride) {
    const size_t n = pix / HW;
    const float* sc = scene + n * C;
    const float* ct = zc + pix * C;
    float dot = 0.f;
    for (int cb = lane; cb < c4; cb += 64) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(sc + cb * 4);
      const f32x4 b = bn_relu4(*reinterpret_cast<const f32x4*>(ct + cb * 4), *reinterpret_cast<const f32x4*>(ssc + cb * 4),
                               *reinterpret_cast<const f32x4*>(ssc + C + cb * 4));
      dot += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    dot = wave_sum(dot);
    const float rv = 1.f / (1.f + expf(-dot));
    if (lane == 0) r[pix] = rv;
    const float* ft = zf + pix * C;
    float* o = out + pix * C;
    for (int cb = lane; cb < c4; cb += 64) {
      const f32x4 v = bn_relu4(*reinterpret_cast<const f32x4*>(ft + cb * 4), *reinterpret_cast<const f32x4*>(ssf + cb * 4),
                               *reinterpret_cast<const f32x4*>(ssf + C + cb * 4)) * rv;
      *reinterpret_cast<f32x4*>(o + cb * 4) = v;
      m = abs4_bits(m, v);
    }
  }
  if (amax) commit_absmax(amax, m);
}
// Per-lane REGISTER accumulators over a workgroup's pixels (a lane owns NCH 16-byte channel chunks): the scene gradient,
// then per BatchNorm (content, re-encoding) sum g, sum g*xhat, max|g|, max|xhat|; the four waves are folded through LDS once,
// at the end ([4 waves][9][C] floats).  (Accumulating in LDS per pixel — nine dependent read-modify-writes — made this
// kernel slower than the plain relation backward it replaces.)
template <int NCH>
__global__ __launch_bounds__(256) void relation_bn_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ scene, const float* __restrict__ zc,
    const float* __restrict__ ssc, const float* __restrict__ mic, const float* __restrict__ zf, const float* __restrict__ ssf,
    const float* __restrict__ mif, const float* __restrict__ r, float* __restrict__ gc, float* __restrict__ gf,
    float* __restrict__ partial, float* __restrict__ bnp_c, float* __restrict__ bnm_c, float* __restrict__ bnp_f,
    float* __restrict__ bnm_f, int HW, int C, int pix_per_blk, int nblk) {
  extern __shared__ __attribute__((aligned(16))) float sacc[];
  const int c4 = C >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * pix_per_blk, p1 = min(HW, p0 + pix_per_blk);
  const float* sc = scene + (size_t)n * C;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[NCH][9];
  f32x4 s4[NCH], scc[NCH], shc[NCH], muc[NCH], isc[NCH], scf[NCH], shf[NCH], muf[NCH], isf[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int cb = lane + 64 * j;
    const bool ok = cb < c4;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[j][k] = zero4;
    s4[j] = ok ? *reinterpret_cast<const f32x4*>(sc + cb * 4) : zero4;
    scc[j] = ok ? *reinterpret_cast<const f32x4*>(ssc + cb * 4) : zero4;
    shc[j] = ok ? *reinterpret_cast<const f32x4*>(ssc + C + cb * 4) : zero4;
    muc[j] = ok ? *reinterpret_cast<const f32x4*>(mic + cb * 4) : zero4;
    isc[j] = ok ? *reinterpret_cast<const f32x4*>(mic + C + cb * 4) : zero4;
    scf[j] = ok ? *reinterpret_cast<const f32x4*>(ssf + cb * 4) : zero4;
    shf[j] = ok ? *reinterpret_cast<const f32x4*>(ssf + C + cb * 4) : zero4;
    muf[j] = ok ? *reinterpret_cast<const f32x4*>(mif + cb * 4) : zero4;
    isf[j] = ok ? *reinterpret_cast<const f32x4*>(mif + C + cb * 4) : zero4;
  }
  auto amax4 = [](f32x4& m, const f32x4 v) {
    m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y)); m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
  };
  for (int p = p0 + wave; p < p1; p += 4) {
    const size_t pix = (size_t)n * HW + p;
    const float rv = r[pix];
    f32x4 a[NCH], zfv[NCH], zcv[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {     // all three rows of the pixel in flight before the first use
      const int cb = lane + 64 * j;
      const bool ok = cb < c4;
      a[j] = ok ? *reinterpret_cast<const f32x4*>(dout + pix * C + cb * 4) : zero4;
      zfv[j] = ok ? *reinterpret_cast<const f32x4*>(zf + pix * C + cb * 4) : zero4;
      zcv[j] = ok ? *reinterpret_cast<const f32x4*>(zc + pix * C + cb * 4) : zero4;
    }
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {     // re-encoding branch: g_f = dout * r where the activation is positive
      const int cb = lane + 64 * j;
      const f32x4 y = bn_relu4(zfv[j], scf[j], shf[j]);
      dot += a[j].x * y.x + a[j].y * y.y + a[j].z * y.z + a[j].w * y.w;
      f32x4 g = a[j] * rv;
      g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f; g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
      if (cb < c4) *reinterpret_cast<f32x4*>(gf + pix * C + cb * 4) = g;
      const f32x4 xh = (zfv[j] - muf[j]) * isf[j];
      acc[j][5] += g;
      acc[j][6] += g * xh;
      amax4(acc[j][7], g);
      amax4(acc[j][8], xh);
    }
    dot = wave_sum(dot);
    const float dz = dot * rv * (1.f - rv);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {     // content branch: g_c = dz * scene where the activation is positive
      const int cb = lane + 64 * j;
      const f32x4 y = bn_relu4(zcv[j], scc[j], shc[j]);
      f32x4 g = s4[j] * dz;
      g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f; g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
      if (cb < c4) *reinterpret_cast<f32x4*>(gc + pix * C + cb * 4) = g;
      acc[j][0] += y * dz;
      const f32x4 xh = (zcv[j] - muc[j]) * isc[j];
      acc[j][1] += g;
      acc[j][2] += g * xh;
      amax4(acc[j][3], g);
      amax4(acc[j][4], xh);
    }
  }
  float* my = sacc + (size_t)wave * 9 * C;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int cb = lane + 64 * j;
    if (cb < c4)
#pragma unroll
      for (int k = 0; k < 9; ++k) *reinterpret_cast<f32x4*>(my + (size_t)k * C + cb * 4) = acc[j][k];
  }
  __syncthreads();
  const size_t blk = (size_t)n * nblk + blockIdx.x;
  const size_t W9 = (size_t)9 * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    auto fold = [&](int k) { return (sacc[k * C + c] + sacc[W9 + k * C + c]) + (sacc[2 * W9 + k * C + c] + sacc[3 * W9 + k * C + c]); };
    auto fmx = [&](int k) {
      return fmaxf(fmaxf(sacc[k * C + c], sacc[W9 + k * C + c]), fmaxf(sacc[2 * W9 + k * C + c], sacc[3 * W9 + k * C + c]));
    };
    partial[blk * C + c] = fold(0);
    bnp_c[blk * 2 * C + c] = fold(1);
    bnp_c[blk * 2 * C + C + c] = fold(2);
    bnm_c[blk * 2 * C + c] = fmx(3);
    bnm_c[blk * 2 * C + C + c] = fmx(4);
    bnp_f[blk * 2 * C + c] = fold(5);
    bnp_f[blk * 2 * C + C + c] = fold(6);
    bnm_f[blk * 2 * C + c] = fmx(7);
    bnm_f[blk * 2 * C + C + c] = fmx(8);
  }
}

static int relation_blocks(int HW) {
  // pixels per workgroup and the cap on workgroups per image (swept in the step in round 5: level)
  constexpr int ppb = 64, cap = 256;
  int b = (HW + ppb - 1) / ppb;
  return b > cap ? cap : (b < 1 ? 1 : b);
}

}  // namespace evk

using namespace evk;

#define EW_LAUNCH(OP, a, b, o, n, alpha, name)                                                                  \
  do {                                                                                                          \
    EVK_REQUIRE((a) && (o) && (n) >= 0, EVK_E_INVALID, name ": bad argument");                                  \
    if ((n) == 0) return EVK_OK;                                                                                \
    hipLaunchKernelGGL(ew_kernel<OP>, dim3(grid_for(((size_t)(n) + 3) / 4)), dim3(256), 0, (hipStream_t)stream, \
                       a, b, o, (size_t)(n), alpha);                                                            \
    return check_launch(name);                                                                                  \
  } while (0)

extern "C" int evk_relu_fwd(const float* x, float* y, int64_t n, void* stream) {
  EW_LAUNCH(0, x, (const float*)nullptr, y, n, 0.f, "relu_fwd");
}
extern "C" int evk_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
  EVK_REQUIRE(y, EVK_E_INVALID, "relu_bwd: null y");
  EW_LAUNCH(1, dy, y, dx, n, 0.f, "relu_bwd");
}
extern "C" int evk_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
  EVK_REQUIRE(b, EVK_E_INVALID, "add: null b");
  EW_LAUNCH(2, a, b, out, n, 0.f, "add");
}
extern "C" int evk_scale(const float* a, float alpha, float* out, int64_t n, void* stream) {
  EW_LAUNCH(3, a, (const float*)nullptr, out, n, alpha, "scale");
}
// Grouped convolution as a dense one (reference _resnets.py:21-24 `groups=groups`, the ResNeXt bodies :291-324): the
// weight [Cout][taps][Cin / groups] becomes the block-diagonal dense [Cout][taps][Cin], zeros outside an output channel's
// group — exact (the zeros contribute exact zeros in every arithmetic), and at 32 groups x 4..8 channels the dense form is
// what fills an MFMA tile anyway.  The adjoint gathers the diagonal blocks of the dense weight gradient.
__global__ __launch_bounds__(256) void group_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout,
                                                           int taps, int Cin, int groups, int gather) {
  const int cpg = Cin / groups, opg = Cout / groups;
  const size_t n = (size_t)Cout * taps * (gather ? cpg : Cin);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (gather) {   // dst = grouped [Cout][taps][cpg], src = dense
      const int c = (int)(i % cpg);
      const size_t r = i / cpg;           // co * taps + tap
      const int co = (int)(r / taps);
      dst[i] = src[r * Cin + (size_t)(co / opg) * cpg + c];
    } else {        // dst = dense [Cout][taps][Cin], src = grouped
      const int ci = (int)(i % Cin);
      const size_t r = i / Cin;
      const int co = (int)(r / taps), g = co / opg;
      const int c = ci - g * cpg;
      dst[i] = (c >= 0 && c < cpg) ? src[r * cpg + c] : 0.f;
    }
  }
}
static int group_weight(const float* src, float* dst, int Cout, int taps, int Cin, int groups, int gather, void* stream) {
  EVK_REQUIRE(src && dst, EVK_E_INVALID, "group_weight: null pointer");
  EVK_REQUIRE(Cout > 0 && taps > 0 && Cin > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0, EVK_E_INVALID,
              "group_weight: Cout=%d Cin=%d groups=%d", Cout, Cin, groups);
  const size_t n = (size_t)Cout * taps * (gather ? Cin / groups : Cin);
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(group_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, Cout, taps, Cin, groups, gather);
  return check_launch("group_weight");
}
extern "C" int evk_group_weight_expand(const float* w, float* dense, int32_t Cout, int32_t taps, int32_t Cin, int32_t groups,
                                       void* stream) {
  return group_weight(w, dense, Cout, taps, Cin, groups, 0, stream);
}
extern "C" int evk_group_weight_gather(const float* ddense, float* dw, int32_t Cout, int32_t taps, int32_t Cin, int32_t groups,
                                       void* stream) {
  return group_weight(ddense, dw, Cout, taps, Cin, groups, 1, stream);
}
extern "C" int evk_mul_scale(const float* a, const float* b, float alpha, float* out, int64_t n, void* stream) {
  EVK_REQUIRE(b, EVK_E_INVALID, "mul_scale: null b");
  EW_LAUNCH(4, a, b, out, n, alpha, "mul_scale");
}
extern "C" int evk_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
  EW_LAUNCH(5, x, (const float*)nullptr, y, n, 0.f, "gelu_fwd");
}
extern "C" int evk_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
  EVK_REQUIRE(x, EVK_E_INVALID, "gelu_bwd: null x");
  EW_LAUNCH(6, dy, x, dx, n, 0.f, "gelu_bwd");
}
extern "C" int evk_mean4_fwd(const float* a, const float* b, const float* c, const float* d, float* out, int64_t n,
                             uint32_t* out_absmax, void* stream) {
  EVK_REQUIRE(a && b && c && d && out && n > 0 && n % 4 == 0, EVK_E_INVALID, "mean4: bad argument");
  hipLaunchKernelGGL(mean4_kernel, dim3(grid_for((size_t)n / 4)), dim3(256), 0, (hipStream_t)stream, a, b, c, d, out,
                     (size_t)n / 4, out_absmax);
  return check_launch("mean4");
}

extern "C" int evk_pad_channels(const float* src, float* dst, int64_t rows, int32_t C, int32_t Cp, void* stream) {
  EVK_REQUIRE(src && dst && rows > 0 && C > 0 && Cp >= C, EVK_E_INVALID, "pad_channels: bad argument");
  hipLaunchKernelGGL(pad_channels_kernel, dim3(grid_for((size_t)rows * Cp)), dim3(256), 0, (hipStream_t)stream, src,
                     dst, (size_t)rows, C, Cp);
  return check_launch("pad_channels");
}
extern "C" int evk_unpad_channels(const float* src, float* dst, int64_t rows, int32_t Cp, int32_t C, void* stream) {
  EVK_REQUIRE(src && dst && rows > 0 && C > 0 && Cp >= C, EVK_E_INVALID, "unpad_channels: bad argument");
  hipLaunchKernelGGL(unpad_channels_kernel, dim3(grid_for((size_t)rows * C)), dim3(256), 0, (hipStream_t)stream, src,
                     dst, (size_t)rows, Cp, C);
  return check_launch("unpad_channels");
}
extern "C" int evk_nchw_to_nhwc(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cp,
                                void* stream) {
  EVK_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, EVK_E_INVALID, "nchw_to_nhwc: bad argument");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((size_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, src,
                     dst, N, C, (size_t)H * W, Cp);
  return check_launch("nchw_to_nhwc");
}
extern "C" int evk_nhwc_to_nchw(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cp,
                                void* stream) {
  EVK_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, EVK_E_INVALID, "nhwc_to_nchw: bad argument");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((size_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, src,
                     dst, N, C, (size_t)H * W, Cp);
  return check_launch("nhwc_to_nchw");
}

extern "C" int evk_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* code, int32_t N, int32_t H, int32_t W, int32_t C,
                                    void* stream) {
  EVK_REQUIRE(x && y && code && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, EVK_E_INVALID,
              "maxpool_fwd: bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((size_t)N * Ho * Wo * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, y, code, N, H, W, C, Ho, Wo);
  return check_launch("maxpool_fwd");
}
extern "C" int evk_maxpool3x3s2_bwd(const float* dy, const uint8_t* code, float* dx, int32_t N, int32_t H, int32_t W,
                                    int32_t C, void* stream) {
  EVK_REQUIRE(dy && dx && code && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, EVK_E_INVALID,
              "maxpool_bwd: bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((size_t)N * H * W * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, dy, code, dx, N, H, W, C, Ho, Wo);
  return check_launch("maxpool_bwd");
}

extern "C" int evk_upsample_nearest2x_add_fwd(const float* top, const float* lateral, float* out, int32_t N, int32_t H,
                                              int32_t W, int32_t C, uint32_t* out_absmax, void* stream) {
  EVK_REQUIRE(top && lateral && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && H % 2 == 0 && W % 2 == 0,
              EVK_E_INVALID, "nearest2x_add: bad argument (H,W must be even, C %% 4 == 0)");
  hipLaunchKernelGGL(nearest2x_add_kernel, dim3(grid_for((size_t)N * H * W * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, top, lateral, out, N, H, W, C, out_absmax);
  return check_launch("nearest2x_add");
}
extern "C" int evk_upsample_nearest2x_bwd(const float* dout, float* dtop, int32_t N, int32_t H, int32_t W, int32_t C,
                                          void* stream) {
  EVK_REQUIRE(dout && dtop && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && H % 2 == 0 && W % 2 == 0,
              EVK_E_INVALID, "nearest2x_bwd: bad argument");
  hipLaunchKernelGGL(nearest2x_bwd_kernel, dim3(grid_for((size_t)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, dout, dtop, N, H, W, C);
  return check_launch("nearest2x_bwd");
}

extern "C" int evk_subsample2_fwd(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  EVK_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, EVK_E_INVALID, "subsample2_fwd: bad argument (C %% 4 == 0)");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(subsample2_kernel<false>, dim3(grid_for((size_t)N * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y,
                     N, H, W, C, Ho, Wo);
  return check_launch("subsample2_fwd");
}
extern "C" int evk_subsample2_bwd(const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  EVK_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, EVK_E_INVALID, "subsample2_bwd: bad argument (C %% 4 == 0)");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(subsample2_kernel<true>, dim3(grid_for((size_t)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, dx,
                     N, H, W, C, Ho, Wo);
  return check_launch("subsample2_bwd");
}

static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

extern "C" int evk_upsample_bilinear_fwd(const float* x, float* y, int32_t N, int32_t Hi, int32_t Wi, int32_t Ho,
                                         int32_t Wo, int32_t C, void* stream) {
  EVK_REQUIRE(x && y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, EVK_E_INVALID,
              "bilinear_fwd: bad argument");
  const float sy = ac_scale(Hi, Ho), sx = ac_scale(Wi, Wo);
  // patch of one kTileR x kTileC output tile: rows/cols <= floor((tile - 1) * scale) + 3 (first i0 .. last i1)
  const int prow = (int)((kTileR - 1) * sy) + 3, pcol = (int)((kTileC - 1) * sx) + 3;
  const size_t patch_bytes = (size_t)prow * pcol * C * sizeof(float);
  if (C % 4 == 0 && C >= 128 && patch_bytes <= 64 * 1024 && N <= 65535 && (Ho + kTileR - 1) / kTileR <= 65535) {
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bilinear_fwd_tile_kernel<64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bilinear_fwd_tile_kernel<32>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    }
    const dim3 grid((Wo + kTileC - 1) / kTileC, (Ho + kTileR - 1) / kTileR, N);
    if (C <= 128)
      hipLaunchKernelGGL(bilinear_fwd_tile_kernel<32>, grid, dim3(256), patch_bytes, (hipStream_t)stream, x, y, Hi, Wi, Ho, Wo,
                         C, sy, sx, pcol);
    else
      hipLaunchKernelGGL(bilinear_fwd_tile_kernel<64>, grid, dim3(256), patch_bytes, (hipStream_t)stream, x, y, Hi, Wi, Ho, Wo,
                         C, sy, sx, pcol);
  }
  else if (C % 4 == 0)
    hipLaunchKernelGGL(bilinear_fwd_kernel<4>, dim3(grid_for((size_t)N * Ho * Wo * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, x, y, N, Hi, Wi, Ho, Wo, C, sy, sx);
  else
    hipLaunchKernelGGL(bilinear_fwd_kernel<1>, dim3(grid_for((size_t)N * Ho * Wo * C)), dim3(256), 0,
                       (hipStream_t)stream, x, y, N, Hi, Wi, Ho, Wo, C, sy, sx);
  return check_launch("bilinear_fwd");
}
extern "C" int evk_upsample_bilinear_bwd(const float* dy, float* dx, int32_t N, int32_t Hi, int32_t Wi, int32_t Ho,
                                         int32_t Wo, int32_t C, void* stream) {
  EVK_REQUIRE(dy && dx && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, EVK_E_INVALID,
              "bilinear_bwd: bad argument");
  const float sy = ac_scale(Hi, Ho), sx = ac_scale(Wi, Wo);
  // inverse scales for the candidate range; in == 1 => every output maps to input 0
  const float isy = sy > 0.f ? 1.f / sy : (float)Ho, isx = sx > 0.f ? 1.f / sx : (float)Wo;
  const long long npix_i = (long long)N * Hi * Wi;
  // candidate range per axis: ceil((i+1)/s)+1 - (floor((i-1)/s)-1) + 1 <= 2/s + 5; the wave kernel holds 16 per axis
  const bool narrow = 2.f * isy + 5.f <= 16.f && 2.f * isx + 5.f <= 16.f && Hi > 1 && Wi > 1;
  if (C % 4 == 0 && C >= 128 && C <= 1024 && narrow && npix_i < 0x7fffffffLL)
    hipLaunchKernelGGL(bilinear_bwd_wave_kernel, dim3((unsigned)((npix_i + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       dy, dx, (int)npix_i, make_fastdiv((uint32_t)Wi), make_fastdiv((uint32_t)Hi), Ho, Wo, C, sy, sx,
                       isy, isx);
  else if (C % 4 == 0)
    hipLaunchKernelGGL(bilinear_bwd_kernel<4>, dim3(grid_for((size_t)N * Hi * Wi * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, dy, dx, N, Hi, Wi, Ho, Wo, C, sy, sx, isy, isx);
  else
    hipLaunchKernelGGL(bilinear_bwd_kernel<1>, dim3(grid_for((size_t)N * Hi * Wi * C)), dim3(256), 0,
                       (hipStream_t)stream, dy, dx, N, Hi, Wi, Ho, Wo, C, sy, sx, isy, isx);
  return check_launch("bilinear_bwd");
}

extern "C" int evk_gap_fwd(const float* x, float* y, int32_t N, int32_t HW, int32_t C, void* stream) {
  EVK_REQUIRE(x && y && N > 0 && HW > 0 && C > 0 && C % 4 == 0, EVK_E_INVALID, "gap_fwd: bad argument");
  const int c4 = C / 4;
  const int tpc = c4 < 64 ? c4 : 64;
  const int rl = 256 / tpc;
  hipLaunchKernelGGL(gap_fwd_kernel, dim3((c4 + tpc - 1) / tpc, N), dim3(256), 0, (hipStream_t)stream, x, y, HW, C,
                     tpc, rl, 1.f / (float)HW);
  return check_launch("gap_fwd");
}
extern "C" int evk_gap_bwd(const float* dy, float* dx, int32_t N, int32_t HW, int32_t C, void* stream) {
  EVK_REQUIRE(dy && dx && N > 0 && HW > 0 && C > 0 && C % 4 == 0, EVK_E_INVALID, "gap_bwd: bad argument");
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for((size_t)N * HW * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy,
                     dx, N, HW, C, 1.f / (float)HW);
  return check_launch("gap_bwd");
}

extern "C" int evk_relation_fwd(const float* scene, const float* content, const float* feat, float* out, float* r,
                                int32_t N, int32_t HW, int32_t C, void* stream) {
  EVK_REQUIRE(scene && content && feat && out && r && N > 0 && HW > 0 && C > 0 && C % 4 == 0, EVK_E_INVALID,
              "relation_fwd: bad argument");
  hipLaunchKernelGGL(relation_fwd_kernel, dim3(grid_for((size_t)N * HW, 4, 8192)), dim3(256), 0, (hipStream_t)stream,
                     scene, content, feat, out, r, N, HW, C);
  return check_launch("relation_fwd");
}
extern "C" int evk_relation_bn_fwd(const float* scene, const float* zc, const float* scale_shift_c, const float* zf,
                                   const float* scale_shift_f, float* out, float* r, int32_t N, int32_t HW, int32_t C,
                                   uint32_t* out_absmax, void* stream) {
  EVK_REQUIRE(scene && zc && scale_shift_c && zf && scale_shift_f && out && r && N > 0 && HW > 0 && C > 0 && C % 4 == 0,
              EVK_E_INVALID, "relation_bn_fwd: bad argument");
  hipLaunchKernelGGL(relation_bn_fwd_kernel, dim3(grid_for((size_t)N * HW, 4, 8192)), dim3(256), 0, (hipStream_t)stream,
                     scene, zc, scale_shift_c, zf, scale_shift_f, out, r, N, HW, C, out_absmax);
  return check_launch("relation_bn_fwd");
}
// partial records per BatchNorm: evk_relation_bn_parts(N, HW); workspace: evk_relation_bn_workspace_bytes
extern "C" int32_t evk_relation_bn_parts(int32_t N, int32_t HW) { return N > 0 && HW > 0 ? N * relation_blocks(HW) : 0; }
extern "C" size_t evk_relation_bn_workspace_bytes(int32_t N, int32_t HW, int32_t C) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  return (size_t)N * relation_blocks(HW) * C * 9 * sizeof(float);   // dscene partials + 2 x (sums [2][C] + maxima [2][C])
}
extern "C" int evk_relation_bn_bwd(const float* dout, const float* scene, const float* zc, const float* scale_shift_c,
                                   const float* mean_invstd_c, const float* zf, const float* scale_shift_f,
                                   const float* mean_invstd_f, const float* r, float* dscene, float* gc, float* gf,
                                   int32_t N, int32_t HW, int32_t C, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  EVK_REQUIRE(dout && scene && zc && scale_shift_c && mean_invstd_c && zf && scale_shift_f && mean_invstd_f && r && dscene &&
                  gc && gf,
              EVK_E_INVALID, "relation_bn_bwd: null pointer");
  EVK_REQUIRE(N > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 448, EVK_E_UNSUPPORTED, "relation_bn_bwd: C=%d", C);
  EVK_REQUIRE(workspace && workspace_bytes >= evk_relation_bn_workspace_bytes(N, HW, C), EVK_E_WORKSPACE,
              "relation_bn_bwd: workspace too small");
  const int nblk = relation_blocks(HW);
  const int ppb = (HW + nblk - 1) / nblk;
  const size_t nb = (size_t)N * nblk;
  // workspace, in floats (nb = evk_relation_bn_parts records): [nb C] scene-gradient partials | content BatchNorm sums
  // [nb][2][C] | its maxima [nb][2][C] | re-encoding BatchNorm sums | its maxima  (evk_bn_bwd_from_partials takes the pairs)
  float* w = (float*)workspace;
  float* partial = w;
  float* bnp_c = w + nb * C;
  float* bnm_c = w + nb * 3 * C;
  float* bnp_f = w + nb * 5 * C;
  float* bnm_f = w + nb * 7 * C;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)4 * 9 * C * sizeof(float);
  EVK_REQUIRE(lds <= 64 * 1024, EVK_E_UNSUPPORTED, "relation_bn_bwd: C=%d does not fit the LDS", C);
  const int nch = (C / 4 + 63) / 64;
#define EVK_REL_BWD(NCH)                                                                                                   \
  hipLaunchKernelGGL(relation_bn_bwd_kernel<NCH>, dim3(nblk, N), dim3(256), lds, st, dout, scene, zc, scale_shift_c,        \
                     mean_invstd_c, zf, scale_shift_f, mean_invstd_f, r, gc, gf, partial, bnp_c, bnm_c, bnp_f, bnm_f, HW, C, \
                     ppb, nblk)
  if (nch == 1) EVK_REL_BWD(1);
  else if (nch == 2) EVK_REL_BWD(2);
  else EVK_REL_BWD(4);
#undef EVK_REL_BWD
  int rc = check_launch("relation_bn_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(relation_dscene_final_kernel, dim3((C + 7) / 8, N), dim3(256), 0, st, (const float*)partial, dscene, nblk,
                     C);
  return check_launch("relation_dscene_final");
}
extern "C" size_t evk_relation_workspace_bytes(int32_t N, int32_t HW, int32_t C) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  return (size_t)N * relation_blocks(HW) * C * sizeof(float);
}
extern "C" int evk_relation_bwd(const float* dout, const float* scene, const float* content, const float* feat,
                                const float* r, float* dscene, float* dcontent, float* dfeat, int32_t N, int32_t HW,
                                int32_t C, void* workspace, size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(dout && scene && content && feat && r && dscene && dcontent && dfeat, EVK_E_INVALID,
              "relation_bwd: null pointer");
  EVK_REQUIRE(N > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 4096, EVK_E_UNSUPPORTED, "relation_bwd: C=%d", C);
  EVK_REQUIRE(workspace && workspace_bytes >= evk_relation_workspace_bytes(N, HW, C), EVK_E_WORKSPACE,
              "relation_bwd: workspace too small");
  const int nblk = relation_blocks(HW);
  const int ppb = (HW + nblk - 1) / nblk;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(relation_bwd_kernel, dim3(nblk, N), dim3(256), 4 * C * sizeof(float), st, dout, scene, content,
                     feat, r, dcontent, dfeat, (float*)workspace, HW, C, ppb, nblk);
  int rc = check_launch("relation_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(relation_dscene_final_kernel, dim3((C + 7) / 8, N), dim3(256), 0, st,
                     (const float*)workspace, dscene, nblk, C);
  return check_launch("relation_dscene_final");
}
