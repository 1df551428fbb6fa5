// Weight-plane producers of the split ("x3") kernels as device functions over a flat thread range
// [t0, t0 + nthreads, ...): shared by the one-tensor kernels (evk_conv2d_split_weight) and the one-launch-per-step
// multi-tensor kernel (evk_conv2d_split_multi, split_weight_multi.hip).
#pragma once
#include "x3_common.hpp"
#include "../../include/ever_hip.h"

namespace evk {

constexpr int kSplitFwd = 0, kSplitDgrad = 1, kSplitHalo = 2;

// one pair of weights -> its planes at o (uint32 index; `plane` = bf16/fp16 elements per plane).  wscale == nullptr: the
// exact 3-term bf16 split; else the 2-term fp16 split of w / s, s from the weight's max |w| (x3_common.hpp), planes h, l.
__device__ __forceinline__ void split_put(float x0, float x1, uint32_t* o, size_t plane, const uint32_t* wscale) {
  if (wscale) {
    const float inv = op_scale(*wscale).inv;
    uint32_t h, l;
    split2h(x0 * inv, x1 * inv, h, l);
    o[0] = h;
    o[plane >> 1] = l;
  } else {
    uint32_t h, m, l;
    split2(x0, x1, h, m, l);
    o[0] = h;
    o[plane >> 1] = m;
    o[plane] = l;
  }
}
constexpr int kHaloCh = 16;  // channels per chunk of the halo kernel's planes (= its kCh)

// Forward planes: row co, k = (ky, kx, ci) as in the OHWI parameter.  out[pt][row][Kpad] bf16, zero padded along K.
__device__ __forceinline__ void split_fwd_body(const float* __restrict__ w, uint16_t* __restrict__ out, int rows, int K,
                                               int Kpad, size_t t0, size_t nthreads, const uint32_t* wscale = nullptr) {
  const size_t total = (size_t)rows * (Kpad >> 1);
  const size_t plane = (size_t)rows * Kpad;
  for (size_t i = t0; i < total; i += nthreads) {
    const int row = (int)(i / (Kpad >> 1));
    const int k = (int)(i - (size_t)row * (Kpad >> 1)) * 2;
    const float x0 = k < K ? w[(size_t)row * K + k] : 0.f;
    const float x1 = k + 1 < K ? w[(size_t)row * K + k + 1] : 0.f;
    split_put(x0, x1, reinterpret_cast<uint32_t*>(out + (size_t)row * Kpad + k), plane, wscale);
  }
}

// Data-gradient planes of one residue class: row ci, k = (jy, jx, co) with ky = ky0 + jy*ksy, kx = kx0 + jx*ksx
// (the class-ordered layout of pack_dgrad_weight_kernel, produced straight from the OHWI parameter).
__device__ __forceinline__ void split_dgrad_body(const float* __restrict__ w, uint16_t* __restrict__ out, int Cout,
                                                 int kh, int kw, int Cin, int ky0, int ksy, int nty, int kx0, int ksx,
                                                 int ntx, int Kpad, size_t t0, size_t nthreads, const uint32_t* wscale = nullptr) {
  const int K = nty * ntx * Cout;
  const size_t total = (size_t)Cin * (Kpad >> 1);
  const size_t plane = (size_t)Cin * Kpad;
  for (size_t i = t0; i < total; i += nthreads) {
    const int ci = (int)(i / (Kpad >> 1));
    const int k = (int)(i - (size_t)ci * (Kpad >> 1)) * 2;
    float x[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int kk = k + e;
      if (kk < K) {
        const int co = kk % Cout;
        const int t = kk / Cout;
        const int jx = t % ntx, jy = t / ntx;
        x[e] = w[(((size_t)co * kh + (ky0 + jy * ksy)) * kw + (kx0 + jx * ksx)) * Cin + ci];
      } else {
        x[e] = 0.f;
      }
    }
    split_put(x[0], x[1], reinterpret_cast<uint32_t*>(out + (size_t)ci * Kpad + k), plane, wscale);
  }
}

// Planes for the LDS-halo 3x3 kernel: out[pt][tap][chunk][row][16] bf16; tap = jy*3 + jx in the kernel's (affine)
// tap order, which is (ky, kx) for the forward and for the stride-1 data gradient alike (the sign of oys flips the
// direction, not the index).  rows = Cout (forward: w[row][ky][kx][ci]) or Cin (data gradient: w[co][ky][kx][row]).
__device__ __forceinline__ void split_halo_body(const float* __restrict__ w, uint16_t* __restrict__ out, int Cout,
                                                int Cin, int for_dgrad, size_t t0, size_t nthreads, const uint32_t* wscale = nullptr) {
  const int rows = for_dgrad ? Cin : Cout, K = for_dgrad ? Cout : Cin;  // K = reduction channels
  const int nchunk = (K + kHaloCh - 1) / kHaloCh;   // (a partial last chunk is zero-filled: K % 16 != 0, K even)
  const size_t total = (size_t)9 * nchunk * rows * (kHaloCh / 2);
  const size_t plane = (size_t)9 * nchunk * rows * kHaloCh;
  for (size_t i = t0; i < total; i += nthreads) {
    const int k2 = (int)(i % (kHaloCh / 2));
    size_t r = i / (kHaloCh / 2);
    const int row = (int)(r % rows);
    r /= rows;
    const int ch = (int)(r % nchunk);
    const int tap = (int)(r / nchunk);
    const int kc = ch * kHaloCh + 2 * k2;
    float x0 = 0.f, x1 = 0.f;
    if (kc + 1 < K) {
      if (for_dgrad) {
        x0 = w[((size_t)kc * 9 + tap) * Cin + row];
        x1 = w[((size_t)(kc + 1) * 9 + tap) * Cin + row];
      } else {
        x0 = w[((size_t)row * 9 + tap) * Cin + kc];
        x1 = w[((size_t)row * 9 + tap) * Cin + kc + 1];
      }
    }
    split_put(x0, x1, reinterpret_cast<uint32_t*>(out + (((size_t)tap * nchunk + ch) * rows + row) * kHaloCh + 2 * k2), plane, wscale);
  }
}

// Planes for the Winograd F(2,3) 3x3 kernel (conv3x3_wino_x3.hip; f16x2 arithmetic only): out[pt][ky*4 + xi][chunk][row][16]
// fp16 of U / s, U = (g0, (g0 + g1 + g2)/2, (g0 - g1 + g2)/2, g2) of kernel row ky (formed in fp64, rounded once).  That
// kernel always correlates, so the data gradient's planes hold the transposed and flipped filter:
// g[kx] = w[row][ky][kx][k] (forward: rows = Cout) or w[k][2 - ky][2 - kx][row] (data gradient: rows = Cin).
__device__ __forceinline__ void split_wino_body(const float* __restrict__ w, uint16_t* __restrict__ out, int Cout,
                                                int Cin, int for_dgrad, size_t t0, size_t nthreads, const uint32_t* wscale) {
  const int rows = for_dgrad ? Cin : Cout, K = for_dgrad ? Cout : Cin;
  const int nchunk = (K + kHaloCh - 1) / kHaloCh;
  const size_t total = (size_t)3 * nchunk * rows * (kHaloCh / 2);
  const size_t tap = (size_t)nchunk * rows * kHaloCh, plane = 12 * tap;
  for (size_t i = t0; i < total; i += nthreads) {
    const int k2 = (int)(i % (kHaloCh / 2));
    size_t r = i / (kHaloCh / 2);
    const int row = (int)(r % rows);
    r /= rows;
    const int ch = (int)(r % nchunk);
    const int ky = (int)(r / nchunk);
    const int kc = ch * kHaloCh + 2 * k2;
    float u[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    if (kc + 1 < K) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        double g[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          g[kx] = for_dgrad ? (double)w[(((size_t)(kc + e) * 3 + (2 - ky)) * 3 + (2 - kx)) * Cin + row]
                            : (double)w[(((size_t)row * 3 + ky) * 3 + kx) * Cin + kc + e];
        u[0][e] = (float)g[0];
        u[1][e] = (float)(0.5 * (g[0] + g[1] + g[2]));
        u[2][e] = (float)(0.5 * (g[0] - g[1] + g[2]));
        u[3][e] = (float)g[2];
      }
    }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
      split_put(u[xi][0], u[xi][1],
                reinterpret_cast<uint32_t*>(out + ((size_t)(ky * 4 + xi) * nchunk + ch) * rows * kHaloCh + (size_t)row * kHaloCh + 2 * k2),
                plane, wscale);
  }
}

// pairs (= threads' worth of work) of one job, for grid sizing
static inline size_t split_job_pairs(const evk_split_job& j) {
  switch (j.kind) {
    case kSplitFwd: return (size_t)j.arg[0] * (j.arg[2] >> 1);
    case kSplitDgrad: return (size_t)j.arg[3] * (j.arg[10] >> 1);
    default: {
      const int rows = j.arg[2] ? j.arg[1] : j.arg[0], K = j.arg[2] ? j.arg[0] : j.arg[1];
      return (size_t)9 * ((K + kHaloCh - 1) / kHaloCh) * rows * (kHaloCh / 2);
    }
  }
}

}  // namespace evk
