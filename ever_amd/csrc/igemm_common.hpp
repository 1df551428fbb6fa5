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
  int planes;               // 3 (0 means 3): exact split, six products; 1: plain bf16 operands, one product (*_bf16 entry points);
                            // 2: 2-term fp16 split of scaled operands, three products (*_f16x2 entry points)
  const uint32_t* a_scale;  // planes == 2: bit image of max|src| (evk_absmax) and of max|weight| (the planes' producer)
  const uint32_t* w_scale;
  int a_packed;             // planes == 2: src holds packed (h | l << 16) words of src / s instead of fp32 (x3_common.hpp)
  uint32_t* out_amax;       // optional: the output's operand-scale buffer (64 slots, x3_common.hpp act_absmax), raised with
                            // one atomic max per wave from the epilogue — the output is a later convolution's operand
  // BatchNorm statistics of the OUTPUT from the epilogue (forward convolutions followed by a training-mode BatchNorm):
  // bn_part[part][3][Cd] = (count, mean, M2 = sum (y - mean)^2) of the rows of row-part `part`; nullptr = off.
  // Written by igemm_store_rows_stats / bn_part_write; merged (Chan) by evk_bn_fwd_train_parts.
  float* bn_part;
  int bn_scratch_off;       // byte offset of the statistics epilogue's LDS scratch (0: it reuses the operand ring)
  int bn_want;              // host side: statistics requested for this launch (the launcher sets bn_part / bn_parts)
  int bn_parts;             // host side, out: row-parts written (0 = this launch produced no statistics)
  float* bn_buf;            // host side: the caller's partial buffer (bn_part is set from it when the kernel supports it)
  int bn_cap;               // host side: capacity of bn_buf in parts
  // ReLU bits of `accum` (common.hpp: relu_bits_*): accum is the UNMASKED gradient that arrived at a residual block's
  // BatchNorm + add + ReLU, the identity branch's gradient is accum where the block's output was positive — masked here, in
  // the epilogue that adds it, instead of being written and re-read as a tensor of its own (evk_bn_bwd: EVK_BN_LAZY_RES)
  const uint32_t* accum_bits;
};

// f16x2 arithmetic: the accumulators hold (x / s_a) * (w / s_w) sums; multiply by s_a * s_w before the epilogue
template <int MB, int NB>
__device__ __forceinline__ void igemm_scale_acc(f32x16 (&acc)[MB][NB], float s) {
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] *= s;
}

// Store one 32-row block of accumulators whose lane's GEMM row lives at element offset `roff` of dst: bias,
// residual / accumulate, ReLU, 16-byte stores.  The MFMAs are issued as D = W_tile * X_tile^T, so a lane holds
// ONE pixel (column lane&31) and, per accumulator quad r4, FOUR consecutive output channels
// co = 8*r4 + 4*(lane>>5) + {0..3}: one 16-byte store per quad (4x fewer store instructions than the
// row-per-register layout; the small-K 1x1 convolutions are store-issue bound).
// running max of |v| as a bit image (integer order = float order for non-negative floats; NaNs stay visible as > inf)
// (the accumulator is passed by reference with a separate on/off flag: a pointer that may be null pins it to scratch)
struct AmaxAcc {
  uint32_t m;
  bool on;
};
__device__ __forceinline__ void amax_one(AmaxAcc& am, float s) {
  if (am.on) am.m = max(am.m, __builtin_bit_cast(uint32_t, s) & 0x7fffffffu);
}
__device__ __forceinline__ void amax_quad(AmaxAcc& am, const f32x4 v) {
  amax_one(am, v.x); amax_one(am, v.y); amax_one(am, v.z); amax_one(am, v.w);
}
// one atomic max per wave into slot (workgroup & 63) of the output's operand-scale buffer (64 slots, 32 words apart)
__device__ __forceinline__ void amax_commit(uint32_t* __restrict__ slots, uint32_t m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(&slots[(blockIdx.x & 63) * 32], m);
}

template <int NB, int WN>
__device__ __forceinline__ void igemm_store_rows(const IGemmArgs& p, f32x16 (&accrow)[NB], size_t roff, int n0, int wn,
                                                 int lh, AmaxAcc& am) {
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
          if (has_a) {
            f32x4 av = *reinterpret_cast<const f32x4*>(p.accum + roff + col);
            if (p.accum_bits) av = relu_bits_mask(av, p.accum_bits, (roff + col) >> 2);
            t += av;
          }
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
        amax_quad(am, v);
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
        if (p.accum) {
          f32x4 av = *reinterpret_cast<const f32x4*>(p.accum + roff + col);
          if (p.accum_bits) av = relu_bits_mask(av, p.accum_bits, (roff + col) >> 2);
          v += av;
        }
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<f32x4*>(p.dst + roff + col) = v;
        amax_quad(am, v);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < p.Cd) {
            float s = v[e];
            if (p.bias) s += p.bias[col + e];
            if (p.accum) s += p.accum[roff + col + e];
            if (p.relu) s = fmaxf(s, 0.f);
            p.dst[roff + col + e] = s;
            amax_one(am, s);
          }
      }
    }
  }
}

template <int NB, int WN>
__device__ __forceinline__ void igemm_store_rows(const IGemmArgs& p, f32x16 (&accrow)[NB], size_t roff, int n0, int wn,
                                                 int lh) {
  AmaxAcc off{0u, false};
  igemm_store_rows<NB, WN>(p, accrow, roff, n0, wn, lh, off);
}

// Epilogue shared by the row-linear kernels (the C/D fragment layout does not depend on the input dtype).
template <int MB, int NB, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const IGemmArgs& p, f32x16 (&acc)[MB][NB], int m0, int n0, int wm,
                                               int wn, int li, int lh, AmaxAcc& am) {
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
    igemm_store_rows<NB, WN>(p, acc[a], roff, n0, wn, lh, am);
  }
}


template <int MB, int NB, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const IGemmArgs& p, f32x16 (&acc)[MB][NB], int m0, int n0, int wm,
                                               int wn, int li, int lh) {
  AmaxAcc off{0u, false};
  igemm_epilogue<MB, NB, WM, WN>(p, acc, m0, n0, wm, wn, li, lh, off);
}

// ------------------------------------------------------------------------------------------------------------------
// Epilogue THROUGH LDS with BatchNorm partial statistics.  A 32-row accumulator block is parked in a wave-private LDS
// scratch ([32][WN + 4] floats) and read back ROW-contiguously: lane (r, c4) then holds 4 consecutive channels of one
// pixel, stores them as part of a full output row (2-4 whole rows per store instruction), and — the point — keeps
// per-lane running sums over the rows it sees for ITS 4 channels: the column statistics need no cross-lane reduction
// until the very end (one or two Chan merges), where the register-layout epilogue would need a 32-lane reduction per
// accumulator register (~1300 VALU per wave and tile, more than a short-K tile's MFMA work).
// Shifted by the lane's first value (pivot): var from sums of (y - pivot) does not cancel when |mean| >> std.
struct BnLaneStat {
  float n;
  f32x4 piv, s, q;
};
__device__ __forceinline__ void bn_stat_init(BnLaneStat& st) {
  st.n = 0.f;
  st.piv = st.s = st.q = f32x4{0.f, 0.f, 0.f, 0.f};
}

// roff: element offset of THIS lane's row (row li of the block) in dst, or ~0 when the row is outside the tensor
template <int NB, int WN>
__device__ __forceinline__ void igemm_store_rows_stats(const IGemmArgs& p, f32x16 (&accrow)[NB], size_t roff, int n0, int wn,
                                                       int li, int lh, float* scratch, BnLaneStat& st) {
  constexpr int P = WN + 4, LPR = WN / 4, RPI = 64 / LPR;
  static_assert(WN == NB * 32 && (64 % LPR) == 0, "column block layout");
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const f32x4 v = {accrow[b][4 * r4], accrow[b][4 * r4 + 1], accrow[b][4 * r4 + 2], accrow[b][4 * r4 + 3]};
      *reinterpret_cast<f32x4*>(scratch + li * P + b * 32 + 8 * r4 + 4 * lh) = v;
    }
  const int lane = lh * 32 + li;
  const int rsub = lane / LPR, c4 = (lane % LPR) * 4;
  const int col = n0 + wn * WN + c4;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + col);
  const uint32_t ro_lo = (uint32_t)roff, ro_hi = (uint32_t)(roff >> 32);
#pragma unroll
  for (int it = 0; it < 32 / RPI; ++it) {
    const int r = it * RPI + rsub;
    const size_t ro = ((size_t)(uint32_t)__shfl((int)ro_hi, r, 64) << 32) | (uint32_t)__shfl((int)ro_lo, r, 64);
    f32x4 v = *reinterpret_cast<const f32x4*>(scratch + r * P + c4) + bias;
    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (ro != ~(size_t)0) {
      *reinterpret_cast<f32x4*>(p.dst + ro + col) = v;
      if (st.n == 0.f) st.piv = v;
      const f32x4 d = v - st.piv;
      st.s += d;
      st.q += d * d;
      st.n += 1.f;
    }
  }
}

// merge the RPI lanes that share a channel group, then the row waves of the tile (through `xch`, an LDS area of
// WAVES_N x 3 x WN floats outside every wave's scratch; ONE workgroup barrier — waves that have already ended are not
// waited for), and write ONE record per tile: bn_part[tile][3][Cd]
template <int WN, int WAVES_M, int WAVES_N = 2>
__device__ __forceinline__ void bn_part_write(const IGemmArgs& p, const BnLaneStat& st, int tile, int n0, int wm, int wn,
                                              int lane, float* xch) {
  constexpr int LPR = WN / 4;
  static_assert(WAVES_M == 1 || WAVES_M == 2 || WAVES_M == 4, "row waves per tile");
  float n = st.n;
  const float inv = n > 0.f ? 1.f / n : 0.f;
  f32x4 mean = st.piv + st.s * inv;
  f32x4 m2 = st.q - st.s * st.s * inv;
  auto merge = [&](float n2, const f32x4& mean2, const f32x4& m22) {
    const float nt = n + n2;
    const float w2 = nt > 0.f ? n2 / nt : 0.f;
    const f32x4 dlt = mean2 - mean;
    mean += dlt * w2;
    m2 += m22 + dlt * dlt * (n * w2);
    n = nt;
  };
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    const float n2 = __shfl_xor(n, off, 64);
    f32x4 mean2, m22;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mean2[e] = __shfl_xor(mean[e], off, 64);
      m22[e] = __shfl_xor(m2[e], off, 64);
    }
    merge(n2, mean2, m22);
  }
  if (WAVES_M == 2) {
    float* x = xch + wn * 3 * WN + lane * 4;
    if (wm == 1 && lane < LPR) {
      *reinterpret_cast<f32x4*>(x) = f32x4{n, n, n, n};
      *reinterpret_cast<f32x4*>(x + WN) = mean;
      *reinterpret_cast<f32x4*>(x + 2 * WN) = m2;
    }
    __syncthreads();
    if (wm == 1) return;
    if (lane < LPR) merge(x[0], *reinterpret_cast<const f32x4*>(x + WN), *reinterpret_cast<const f32x4*>(x + 2 * WN));
  }
  if (WAVES_M == 4) {   // (xch: 3 x WAVES_N x 3 x WN floats) row waves 1..3 park their record, row wave 0 merges them in order
    float* x = xch + ((wm > 0 ? wm - 1 : 0) * WAVES_N + wn) * 3 * WN + lane * 4;
    if (wm >= 1 && lane < LPR) {
      *reinterpret_cast<f32x4*>(x) = f32x4{n, n, n, n};
      *reinterpret_cast<f32x4*>(x + WN) = mean;
      *reinterpret_cast<f32x4*>(x + 2 * WN) = m2;
    }
    __syncthreads();
    if (wm >= 1) return;
    if (lane < LPR) {
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        const float* y = xch + (o * WAVES_N + wn) * 3 * WN + lane * 4;
        merge(y[0], *reinterpret_cast<const f32x4*>(y + WN), *reinterpret_cast<const f32x4*>(y + 2 * WN));
      }
    }
  }
  if (lane < LPR) {
    const int col = n0 + wn * WN + lane * 4;
    float* rec = p.bn_part + (size_t)tile * 3 * p.Cd + col;
    *reinterpret_cast<f32x4*>(rec) = f32x4{n, n, n, n};
    *reinterpret_cast<f32x4*>(rec + p.Cd) = mean;
    *reinterpret_cast<f32x4*>(rec + 2 * p.Cd) = m2;
  }
}

// the row-linear kernels' epilogue in statistics mode (dense destination, Cd % 4 == 0, whole column blocks).
// scratch_base: LDS area of (waves x 32 x (WN + 4) + WAVES_N x 3 x WN) floats, free of other use
template <int MB, int NB, int WM, int WN, int WAVES_M, int WAVES_N>
__device__ __forceinline__ void igemm_epilogue_stats(const IGemmArgs& p, f32x16 (&acc)[MB][NB], int m0, int n0, int wm, int wn,
                                                     int li, int lh, float* scratch_base) {
  BnLaneStat st;
  bn_stat_init(st);
  float* scratch = scratch_base + (wm * WAVES_N + wn) * 32 * (WN + 4);
#pragma unroll
  for (int a = 0; a < MB; ++a) {
    const int row = m0 + wm * WM + a * 32 + li;
    const size_t roff = row < p.M ? (size_t)row * p.Cd : ~(size_t)0;
    igemm_store_rows_stats<NB, WN>(p, acc[a], roff, n0, wn, li, lh, scratch, st);
  }
  bn_part_write<WN, WAVES_M, WAVES_N>(p, st, m0 / (WM * WAVES_M), n0, wm, wn, lh * 32 + li,
                             scratch_base + WAVES_M * WAVES_N * 32 * (WN + 4));
}

// host: may this launch (tile BM x BN, WAVES_M row waves) write statistics?  sets a.bn_part / a.bn_parts
inline void bn_stats_setup(IGemmArgs& a, int BM, int BN, int WAVES_M, long long row_tiles) {
  a.bn_part = nullptr;
  a.bn_parts = 0;
  if (!a.bn_want || !a.bn_buf) return;
  (void)WAVES_M;
  const long long parts = row_tiles;   // one record per tile (the row waves are merged in the epilogue)
  if (!a.dense_dst || a.accum || (a.Cd % BN) != 0 || parts > a.bn_cap) return;
  a.bn_part = a.bn_buf;
  a.bn_parts = (int)parts;
}

int launch_igemm(IGemmArgs& a, hipStream_t stream);
int launch_igemm_x3(IGemmArgs& a, hipStream_t stream);
int launch_igemm_x3ws(IGemmArgs& a, hipStream_t stream);  // 1 = not applicable
int launch_igemm_x3ws_forced(IGemmArgs& a, int bn, hipStream_t stream);
int launch_conv1x1_dma(IGemmArgs& a, hipStream_t stream);  // 1 = not applicable (conv1x1_dma.hip)
int launch_conv1x1_dma_forced(IGemmArgs& a, int bn, hipStream_t stream);
bool conv1x1_dma_applicable(const IGemmArgs& a);
int launch_conv1x1_sp_forced(IGemmArgs& a, int bn, hipStream_t stream);   // software-pipelined, loader waves (conv1x1_sp.hip)
bool conv1x1_sp_applicable(const IGemmArgs& a);
int launch_conv1x1_ps2(IGemmArgs& a, hipStream_t stream);  // persistent, loader + compute + store waves (conv1x1_ps2.hip)
bool conv1x1_ps2_applicable(const IGemmArgs& a);
int launch_conv3x3_halo(IGemmArgs& a, hipStream_t stream);  // 1 = not applicable
bool conv_desc_uses_halo(const evk_conv_desc* d, int for_dgrad);
bool conv3x3_halo_applies(const IGemmArgs& a);
int launch_conv3x3_wino(IGemmArgs& a, hipStream_t stream);  // 1 = not applicable (conv3x3_wino_x3.hip; f16x2 only)
bool conv_desc_uses_wino(const evk_conv_desc* d, int for_dgrad);   // geometry only: the caller knows the arithmetic
int launch_split_weight_wino(const float* w, uint16_t* out, int Cout, int Cin, int for_dgrad, hipStream_t st,
                             const uint32_t* wscale);
int launch_split_weight_halo(const float* w, uint16_t* out, int Cout, int Cin, int for_dgrad, hipStream_t st,
                             const uint32_t* wscale = nullptr);

// One axis of the strided data gradient, for input pixels congruent to c (mod stride):
// taps k = k0 + j*kstep (j < nt) reach them, from source row  g + o0 + j*ostep.
struct AxisPlan {
  int k0, kstep, nt, o0, ostep;
};
AxisPlan plan_axis(int c, int pad, int dil, int stride, int ksize);

}  // namespace evk
