// Shared pieces of the implicit-GEMM convolution kernels (fp32 MFMA and 3-way-bf16-split MFMA).
#pragma once
#include "common.hpp"

namespace evk {

struct IGemmArgs {
  const float* src;
  const float* wgt;
  const float* bias;
  const float* accum;       // optional tensor with dst's shape, added in the epilogue (dst = conv + accum)
  float* dst;
  int N, Hs, Ws, Cs;        // gathered tensor
  int Hm, Wm;               // GEMM-row grid
  int Cd;                   // GEMM N
  int kh, kw, cpt;          // taps of this launch; cpt = Cs/4 (16-byte chunks per tap)
  int ash, asw;             // row-grid -> source scale
  int oy0, oys, ox0, oxs;   // tap -> source offset (affine)
  int M, Ktot;
  int Hd, Wd, dsh, dsw, doy, dox;  // destination pixel = (gy*dsh + doy, gx*dsw + dox) in [N,Hd,Wd,Cd]
  int dense_dst;            // 1 => dst row offset = m*Cd
  int relu;
  int tiles_m, tiles_n;
  const uint16_t* wgt3;     // split kernel: weights as 3 bf16 planes [3][Cd][Kpad] (Kpad % 32 == 0, zero padded)
  int Kpad;
};

// Store one 32-row block of accumulators whose lane's GEMM row lives at element offset `roff` of dst: bias,
// residual / accumulate, ReLU, 16-byte stores.  The MFMAs are issued as D = W_tile * X_tile^T, so a lane holds
// ONE pixel (column lane&31) and, per accumulator quad r4, FOUR consecutive output channels
// co = 8*r4 + 4*(lane>>5) + {0..3}: one 16-byte store per quad (4x fewer store instructions than the
// row-per-register layout; the small-K 1x1 convolutions are store-issue bound).
template <int NB, int WN>
__device__ __forceinline__ void igemm_store_rows(const IGemmArgs& p, f32x16 (&accrow)[NB], size_t roff, int n0, int wn,
                                                 int lh) {
  const bool vec = (p.Cd & 3) == 0 && n0 + wn * WN + NB * 32 <= p.Cd;   // whole block inside the tensor, 16-byte rows
  if (vec) {
    // gfx9 has ONE in-order vmcnt for loads and stores: a bias / accumulate load issued after a store is only
    // satisfied once that store has completed, so "load, add, store" per quad serialises the epilogue on the HBM
    // write latency.  All loads of the row block first (4 x NB quads, 16 x NB registers), one wait, then the stores.
    f32x4 add[NB][4];
    const bool has_b = p.bias != nullptr, has_a = p.accum != nullptr;
    if (has_b || has_a) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int col = n0 + wn * WN + b * 32 + 8 * r4 + 4 * lh;
          f32x4 t = {0.f, 0.f, 0.f, 0.f};
          if (has_b) t = *reinterpret_cast<const f32x4*>(p.bias + col);
          if (has_a) t += *reinterpret_cast<const f32x4*>(p.accum + roff + col);
          add[b][r4] = t;
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int col = n0 + wn * WN + b * 32 + 8 * r4 + 4 * lh;
        f32x4 v = {accrow[b][4 * r4], accrow[b][4 * r4 + 1], accrow[b][4 * r4 + 2], accrow[b][4 * r4 + 3]};
        if (has_b || has_a) v += add[b][r4];
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<f32x4*>(p.dst + roff + col) = v;
      }
    return;
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int col = n0 + wn * WN + b * 32 + 8 * r4 + 4 * lh;
      f32x4 v = {accrow[b][4 * r4], accrow[b][4 * r4 + 1], accrow[b][4 * r4 + 2], accrow[b][4 * r4 + 3]};
      if (col + 3 < p.Cd && (p.Cd & 3) == 0) {
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + col);
        if (p.accum) v += *reinterpret_cast<const f32x4*>(p.accum + roff + col);
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<f32x4*>(p.dst + roff + col) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < p.Cd) {
            float s = v[e];
            if (p.bias) s += p.bias[col + e];
            if (p.accum) s += p.accum[roff + col + e];
            if (p.relu) s = fmaxf(s, 0.f);
            p.dst[roff + col + e] = s;
          }
      }
    }
  }
}

// Epilogue shared by the row-linear kernels (the C/D fragment layout does not depend on the input dtype).
template <int MB, int NB, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const IGemmArgs& p, f32x16 (&acc)[MB][NB], int m0, int n0, int wm,
                                               int wn, int li, int lh) {
#pragma unroll
  for (int a = 0; a < MB; ++a) {
    const int row = m0 + wm * WM + a * 32 + li;
    if (row >= p.M) continue;
    size_t roff;
    if (p.dense_dst) {
      roff = (size_t)row * p.Cd;
    } else {
      const int hw = p.Hm * p.Wm;
      const int n = row / hw;
      const int rem = row - n * hw;
      const int gy = rem / p.Wm;
      const int gx = rem - gy * p.Wm;
      roff = (((size_t)n * p.Hd + (size_t)(gy * p.dsh + p.doy)) * p.Wd + (size_t)(gx * p.dsw + p.dox)) * p.Cd;
    }
    igemm_store_rows<NB, WN>(p, acc[a], roff, n0, wn, lh);
  }
}

int launch_igemm(IGemmArgs& a, hipStream_t stream);
int launch_igemm_x3(IGemmArgs& a, hipStream_t stream);
int launch_igemm_x3ws(IGemmArgs& a, hipStream_t stream);  // 1 = not applicable
int launch_igemm_x3dma(IGemmArgs& a, hipStream_t stream);  // 1 = not applicable
int launch_conv3x3_halo(IGemmArgs& a, hipStream_t stream);  // 1 = not applicable
bool conv_desc_uses_halo(const evk_conv_desc* d, int for_dgrad);
int launch_split_weight_halo(const float* w, uint16_t* out, int Cout, int Cin, int for_dgrad, hipStream_t st);

// One axis of the strided data gradient, for input pixels congruent to c (mod stride):
// taps k = k0 + j*kstep (j < nt) reach them, from source row  g + o0 + j*ostep.
struct AxisPlan {
  int k0, kstep, nt, o0, ostep;
};
AxisPlan plan_axis(int c, int pad, int dil, int stride, int ksize);

}  // namespace evk
