// The ResNet stem convolution (7x7, stride 2, padding 3; reference _resnets.py:149, resnet.py:100-117 for 4-band input)
// as a SPACE-TO-DEPTH 4x4 stride-1 convolution, so that it runs on the split-MFMA kernels instead of the exact-fp32
// fallback (Cin = 3 / 4 is not a multiple of 8):
//   s2d[n][a][b][(py*2+px)*4 + c] = x[n][2a+py][2b+px][c]            (2x2 pixel blocks -> 16 channels, c padded to 4)
//   y[oy] = sum_ky w7[ky] x[2oy + ky - 3]  with  2oy + ky - 3 = 2(oy + da) + py,  da = floor((ky-3)/2) in {-2..1}
// i.e. a 4x4 kernel over s2d rows oy-2 .. oy+1.  The s2d tensor is written with 2 zero rows/columns in front and 1
// behind, so the 4x4 convolution needs no padding of its own (asymmetric padding is not expressible in evk_conv_desc):
//   dst [N][H/2 + 3][W/2 + 3][16],   y = conv4x4(dst, w4, stride 1, pad 0)  ->  [N][H/2][W/2][Cout]
//   w4[co][ta][tb][(py*2+px)*4 + c] = w7[co][2ta+py-1][2tb+px-1][c]   (zero where the index leaves 0..6 or c >= C)
// K = 256 of which 147 (3-band) are real taps; the arithmetic is the same sum of the same products.
#include "common.hpp"

namespace evk {

__global__ __launch_bounds__(256) void stem_s2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C,
                                                       int H, int W, int nchw) {
  const int Ha = H / 2 + 3, Wa = W / 2 + 3;
  const size_t total = (size_t)N * Ha * Wa;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int bp = (int)(i % Wa);
  const size_t r = i / Wa;
  const int ap = (int)(r % Ha);
  const int n = (int)(r / Ha);
  const int a = ap - 2, b = bp - 2;
  f32x4 v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a >= 0 && a < H / 2 && b >= 0 && b < W / 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = 2 * a + (q >> 1), x = 2 * b + (q & 1);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < C)
          v[q][c] = nchw ? src[(((size_t)n * C + c) * H + y) * W + x] : src[(((size_t)n * H + y) * W + x) * C + c];
    }
  }
  f32x4* o = reinterpret_cast<f32x4*>(dst + i * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = v[q];
}

// dir 0: w7 [Cout][7][7][C] -> w4 [Cout][4][4][16];  dir 1: dw4 -> dw7 (every w7 element has exactly one image in w4)
__global__ void stem_s2d_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int C, int dir) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (dir == 0) {
    if (i >= Cout * 256) return;
    const int ch = i & 15, tb = (i >> 4) & 3, ta = (i >> 6) & 3, co = i >> 8;
    const int c = ch & 3, px = (ch >> 2) & 1, py = ch >> 3;
    const int ky = 2 * ta + py - 1, kx = 2 * tb + px - 1;
    dst[i] = (c < C && ky >= 0 && ky < 7 && kx >= 0 && kx < 7) ? src[((co * 7 + ky) * 7 + kx) * C + c] : 0.f;
  } else {
    if (i >= Cout * 49 * C) return;
    const int c = i % C;
    int r = i / C;
    const int kx = r % 7;
    r /= 7;
    const int ky = r % 7, co = r / 7;
    const int ta = (ky + 1) >> 1, py = (ky + 1) & 1, tb = (kx + 1) >> 1, px = (kx + 1) & 1;
    dst[i] = src[((co * 4 + ta) * 4 + tb) * 16 + (py * 2 + px) * 4 + c];
  }
}

}  // namespace evk

using namespace evk;

extern "C" int evk_stem_s2d(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t src_is_nchw,
                            void* stream) {
  EVK_REQUIRE(src && dst && N > 0 && C > 0 && C <= 4 && H > 0 && W > 0, EVK_E_INVALID, "stem_s2d: bad argument");
  EVK_REQUIRE((H & 1) == 0 && (W & 1) == 0, EVK_E_UNSUPPORTED, "stem_s2d: H and W must be even (got %d x %d)", H, W);
  const size_t total = (size_t)N * (H / 2 + 3) * (W / 2 + 3);
  hipLaunchKernelGGL(stem_s2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, N,
                     C, H, W, src_is_nchw);
  return check_launch("stem_s2d");
}

extern "C" int evk_stem_s2d_weight(const float* w7, float* w4, int32_t Cout, int32_t C, void* stream) {
  EVK_REQUIRE(w7 && w4 && Cout > 0 && C > 0 && C <= 4, EVK_E_INVALID, "stem_s2d_weight: bad argument");
  hipLaunchKernelGGL(stem_s2d_weight_kernel, dim3((Cout * 256 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w7, w4,
                     Cout, C, 0);
  return check_launch("stem_s2d_weight");
}

extern "C" int evk_stem_s2d_weight_bwd(const float* dw4, float* dw7, int32_t Cout, int32_t C, void* stream) {
  EVK_REQUIRE(dw4 && dw7 && Cout > 0 && C > 0 && C <= 4, EVK_E_INVALID, "stem_s2d_weight_bwd: bad argument");
  hipLaunchKernelGGL(stem_s2d_weight_kernel, dim3((Cout * 49 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, dw4,
                     dw7, Cout, C, 1);
  return check_launch("stem_s2d_weight_bwd");
}
