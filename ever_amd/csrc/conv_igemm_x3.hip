// Implicit-GEMM convolution (forward and data gradient) on the bf16 matrix pipe with fp32-grade
// arithmetic: every fp32 operand is split EXACTLY into three bf16 terms  x = h + m + l  (round-to-
// nearest-even at each level: |m| <= 2^-9 |x|, |l| <= 2^-18 |x|) and the product is assembled from the six
// partial products that are not below fp32 resolution,
//     x*w  ~=  l*wh + h*wl + m*wm + m*wh + h*wm + h*wh        (dropped: m*wl, l*wm, l*wl  <=  2^-26 |x*w|)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs
// (6 x 32 cycles for 32x32x16) replace eight v_mfma_f32_32x32x2_f32 (8 x 64 cycles): 2.67x the
// matrix-pipe rate of the exact-fp32 kernel in conv_igemm.hip, with an error per product below one
// fp32 rounding (measured at the logits against fp64: same as the fp32 CPU reference).
//
// Same gather / destination algebra as conv_igemm.hip (IGemmArgs); requires Cs % 8 == 0.
//   A (im2col rows, fp32 in HBM): global -> VGPR (two 16-byte loads = 8 floats per row slot) -> split in
//     registers (v_cvt_pk_bf16_f32 + exact residuals) -> three bf16 planes in LDS.
//   B (weights): split ONCE per step by split_weight_kernel into three bf16 planes [3][Cd][Kpad] in HBM
//     (they are reused by every tile), streamed global -> VGPR -> LDS with no VALU work.
// LDS stage: 3 x [BM][32] + 3 x [BN][32] bf16, rows of 64 bytes, 16-byte chunk index ^= (row>>2)&3 so
// that the ds_read_b128 fragment reads of a 16-lane group cover 16 distinct 16-byte slots.
// Fragment: lane l holds row l&31, k = 16*kk + 8*(l>>5) .. +7 (one ds_read_b128) for A and B alike.
#include "igemm_common.hpp"
#include "split_weight.hpp"
#include <stdlib.h>
#include <string.h>

namespace evk {

template <int BM, int BN, int WAVES_M, int WAVES_N, int NBUF, int NPX>
__global__ __launch_bounds__(256) void conv_igemm_x3_kernel(const IGemmArgs p) {
  constexpr int NP = X3Mode<NPX>::NP;
  constexpr bool PK = X3Mode<NPX>::PK;   // the activation operand arrives packed (x3_common.hpp)
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MB = WM / 32, NB = WN / 32;
  constexpr int AR = BM / 64, BR = BN / 64;  // row slots per thread (64 rows x 4 sixteen-byte chunks per pass)
  constexpr int kStage = 3 * (BM + BN) * kRowBytes;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % p.tiles_n;
  const int tile_m = bid / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int c4 = tid & 3;   // 8-float group inside the K step
  const int rb = tid >> 2;  // base row 0..63

  int a_y0[AR], a_x0[AR], a_base[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    const int m = m0 + rb + 64 * j;
    if (m < p.M) {
      const int hw = p.Hm * p.Wm;
      const int n = m / hw;
      const int rem = m - n * hw;
      const int gy = rem / p.Wm;
      const int gx = rem - gy * p.Wm;
      a_y0[j] = gy * p.ash + p.oy0;
      a_x0[j] = gx * p.asw + p.ox0;
      a_base[j] = ((n * p.Hs + a_y0[j]) * p.Ws + a_x0[j]) * p.Cs;
    } else {
      a_y0[j] = -(1 << 28);
      a_x0[j] = 0;
      a_base[j] = 0;
    }
  }
  // B rows past Cd re-read the last row (their columns are never stored); K is zero padded in HBM
  int b_off[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    int co = n0 + rb + 64 * j;
    co = co < p.Cd ? co : p.Cd - 1;
    b_off[j] = co * p.Kpad + c4 * 8;
  }
  const int plane = p.Cd * p.Kpad;  // bf16 elements per weight plane

  const int cp8 = p.Cs >> 3;  // 8-float groups per tap
  int cc, kx, ky;
  {
    const int tap = c4 / cp8;
    cc = c4 - tap * cp8;
    ky = tap / p.kw;
    kx = tap - ky * p.kw;
  }

  f32x4 ra[AR][2];
  u32x4 rbv[BR][3];
  uint32_t okmask = 0;
  float a_inv = 1.f, out_scale = 1.f;   // f16x2: 1 / activation scale; activation scale x weight scale
  if constexpr (NP == 2) {
    const OpScale sa = op_scale(act_absmax(p.a_scale)), sw = op_scale(*p.w_scale);
    a_inv = sa.inv;
    out_scale = sa.s * sw.s;
  }

  auto load_tiles = [&](int kt) {
    okmask = 0;
    const bool kvalid = ky < p.kh;
    const int oy = ky * p.oys, ox = kx * p.oxs;
    const int tapoff = (oy * p.Ws + ox) * p.Cs + cc * 8;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      const int sy = a_y0[j] + oy, sx = a_x0[j] + ox;
      const bool ok = kvalid && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      okmask |= ok ? (1u << j) : 0u;
      const float* src = p.src + (ok ? a_base[j] + tapoff : 0);
      ra[j][0] = *reinterpret_cast<const f32x4*>(src);
      ra[j][1] = *reinterpret_cast<const f32x4*>(src + 4);
    }
#pragma unroll
    for (int j = 0; j < BR; ++j)
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
        rbv[j][pt] = *reinterpret_cast<const u32x4*>(p.wgt3 + (size_t)pt * plane + b_off[j] + kt * BK3);
    if (cp8 >= 4) {
      cc += 4;
      const bool wrap = cc >= cp8;
      cc = wrap ? cc - cp8 : cc;
      kx += wrap ? 1 : 0;
      const bool wrapx = kx == p.kw;
      kx = wrapx ? 0 : kx;
      ky += wrapx ? 1 : 0;
    } else {
      const int q = (kt + 1) * 4 + c4;
      const int tap = q / cp8;
      cc = q - tap * cp8;
      ky = tap / p.kw;
      kx = tap - ky * p.kw;
    }
  };

  auto store_tiles = [&](int buf) {
    unsigned char* Ab = smem3 + buf * kStage;
    unsigned char* Bb = Ab + 3 * BM * kRowBytes;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      const int row = rb + 64 * j;
      const int off = row * kRowBytes + ((c4 ^ ((row >> 2) & 3)) << 4);
      const bool ok = (okmask >> j) & 1u;
      u32x4 H, M, L;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 v = ra[j][e >> 1];
        const float x0 = ok ? v[2 * (e & 1)] : 0.f, x1 = ok ? v[2 * (e & 1) + 1] : 0.f;
        uint32_t h, m = 0, l = 0;
        split_op<NP, PK>(x0, x1, a_inv, h, m, l);
        H[e] = h; M[e] = m; L[e] = l;
      }
      *reinterpret_cast<u32x4*>(Ab + off) = H;
      if (NP >= 2) *reinterpret_cast<u32x4*>(Ab + BM * kRowBytes + off) = M;
      if (NP == 3) *reinterpret_cast<u32x4*>(Ab + 2 * BM * kRowBytes + off) = L;
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      const int row = rb + 64 * j;
      const int off = row * kRowBytes + ((c4 ^ ((row >> 2) & 3)) << 4);
#pragma unroll
      for (int pt = 0; pt < NP; ++pt) *reinterpret_cast<u32x4*>(Bb + pt * BN * kRowBytes + off) = rbv[j][pt];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = p.Kpad / BK3;

  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  // fragment byte offsets inside a plane (row part fixed per lane; chunk index depends on kk)
  int fa_off[MB][2], fb_off[NB][2];
#pragma unroll
  for (int a = 0; a < MB; ++a) {
    const int row = wm * WM + a * 32 + li;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fa_off[a][kk] = row * kRowBytes + (((2 * kk + lh) ^ ((row >> 2) & 3)) << 4);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int row = wn * WN + b * 32 + li;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fb_off[b][kk] = row * kRowBytes + (((2 * kk + lh) ^ ((row >> 2) & 3)) << 4);
  }

  auto half_step = [&](const unsigned char* Ab, const unsigned char* Bb, int kk, int first, int last) {
    bf16x8 fa[MB][3], fb[NB][3];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
        fa[a][pt] = *reinterpret_cast<const bf16x8*>(Ab + pt * BM * kRowBytes + fa_off[a][kk]);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
        fb[b][pt] = *reinterpret_cast<const bf16x8*>(Bb + pt * BN * kRowBytes + fb_off[b][kk]);
#pragma unroll
    for (int t = first; t < (last < X3Prod<NP>::N ? last : X3Prod<NP>::N); ++t)
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a][b] = mfma_np<NP>(fb[b][x3_pb(NP, t)], fa[a][x3_pa(NP, t)], acc[a][b]);
  };

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = NBUF == 2 ? (kt & 1) : 0;
    const bool more = kt + 1 < nk;
    const unsigned char* Ab = smem3 + buf * kStage;
    const unsigned char* Bb = Ab + 3 * BM * kRowBytes;
    if (more) load_tiles(kt + 1);
    half_step(Ab, Bb, 0, 0, 6);
    if (NBUF == 2 && more) store_tiles(buf ^ 1);
    half_step(Ab, Bb, 1, 0, 6);
    __syncthreads();
    if (NBUF == 1 && more) {
      store_tiles(0);
      __syncthreads();
    }
  }

  if constexpr (NP == 2) igemm_scale_acc<MB, NB>(acc, out_scale);
  if (p.bn_part) {
    // every wave is done with the last stage (the loop ends on a barrier): the ring becomes epilogue scratch
    igemm_epilogue_stats<MB, NB, WM, WN, WAVES_M, WAVES_N>(p, acc, m0, n0, wm, wn, li, lh, reinterpret_cast<float*>(smem3));
    return;
  }
  AmaxAcc amax_l{0u, p.out_amax != nullptr};
  igemm_epilogue<MB, NB, WM, WN>(p, acc, m0, n0, wm, wn, li, lh, amax_l);
  if (p.out_amax) amax_commit(p.out_amax, amax_l.m);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NBUF, int NP>
static int launch_cfg3_np(IGemmArgs& a, hipStream_t stream) {
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.Cd, BN);
  bn_stats_setup(a, BM, BN, WAVES_M, a.tiles_m);
  size_t lds = (size_t)NBUF * 3 * (BM + BN) * kRowBytes;
  // statistics epilogue: per-wave scratch + the row waves' exchange area
  const size_t scratch = ((size_t)4 * 32 * (BN / WAVES_N + 4) + (size_t)3 * BN) * sizeof(float);
  if (lds < scratch) lds = scratch;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_x3_kernel<BM, BN, WAVES_M, WAVES_N, NBUF, NP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const long long nwg = (long long)a.tiles_m * a.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffLL) {
    set_error("conv_igemm_x3: bad grid %lld", nwg);
    return EVK_E_INVALID;
  }
  hipLaunchKernelGGL((conv_igemm_x3_kernel<BM, BN, WAVES_M, WAVES_N, NBUF, NP>), dim3((unsigned)nwg), dim3(256), lds, stream, a);
  return check_launch("conv_igemm_x3");
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NBUF>
static int launch_cfg3(IGemmArgs& a, hipStream_t stream) {
  if (a.planes == 1) return launch_cfg3_np<BM, BN, WAVES_M, WAVES_N, NBUF, 1>(a, stream);
  if (a.planes == 2)
    return a.a_packed ? launch_cfg3_np<BM, BN, WAVES_M, WAVES_N, NBUF, 4>(a, stream)
                      : launch_cfg3_np<BM, BN, WAVES_M, WAVES_N, NBUF, 2>(a, stream);
  return launch_cfg3_np<BM, BN, WAVES_M, WAVES_N, NBUF, 3>(a, stream);
}

int launch_igemm_x3(IGemmArgs& a, hipStream_t stream) {
  const long long src_elems = (long long)a.N * a.Hs * a.Ws * a.Cs;
  const long long wgt_elems = (long long)a.Cd * a.Kpad * 3;
  if (src_elems >= 0x7fffffffLL || wgt_elems >= 0x7fffffffLL) {
    set_error("conv_igemm_x3: tensors of 2^31 or more elements are not supported");
    return EVK_E_UNSUPPORTED;
  }
  if ((a.Cs & 7) != 0) {
    set_error("conv_igemm_x3: source channels (%d) must be a multiple of 8", a.Cs);
    return EVK_E_UNSUPPORTED;
  }
  static const bool tune = getenv("EVK_TUNE") != nullptr;
  if (tune) {   // tools/autotune_convs.py: EVK_X3_FORCE names the tile shape (read on every launch)
    const char* f = getenv("EVK_X3_FORCE");
    if (f && *f) {
      if (!strcmp(f, "w256")) return launch_igemm_x3ws_forced(a, 256, stream);
      if (!strcmp(f, "w128")) return launch_igemm_x3ws_forced(a, 128, stream);
      if (!strcmp(f, "w64")) return launch_igemm_x3ws_forced(a, 64, stream);
      if (!strcmp(f, "d256") && conv1x1_dma_applicable(a)) return launch_conv1x1_dma_forced(a, 256, stream);
      if (!strcmp(f, "d128") && conv1x1_dma_applicable(a)) return launch_conv1x1_dma_forced(a, 128, stream);
      if (!strcmp(f, "d64") && conv1x1_dma_applicable(a)) return launch_conv1x1_dma_forced(a, 64, stream);
      if (!strcmp(f, "e128") && conv1x1_dma_applicable(a)) return launch_conv1x1_dma_forced(a, 2128, stream);
      if (!strcmp(f, "e64") && conv1x1_dma_applicable(a)) return launch_conv1x1_dma_forced(a, 2064, stream);
      if (!strcmp(f, "q128") && conv1x1_ps2_applicable(a)) return launch_conv1x1_ps2(a, stream);
      if (!strcmp(f, "s128") && conv1x1_sp_applicable(a)) return launch_conv1x1_sp_forced(a, 128, stream);
      if (!strcmp(f, "s64") && conv1x1_sp_applicable(a)) return launch_conv1x1_sp_forced(a, 64, stream);
      if (!strcmp(f, "t128") && conv1x1_sp_applicable(a)) return launch_conv1x1_sp_forced(a, 3128, stream);
      if (!strcmp(f, "t64") && conv1x1_sp_applicable(a)) return launch_conv1x1_sp_forced(a, 3064, stream);
      if (!strcmp(f, "c128x128")) return launch_cfg3<128, 128, 2, 2, 1>(a, stream);
      if (!strcmp(f, "c64x128")) return launch_cfg3<64, 128, 2, 2, 1>(a, stream);
      if (!strcmp(f, "c128x64")) return launch_cfg3<128, 64, 2, 2, 1>(a, stream);
      if (!strcmp(f, "c64x64")) return launch_cfg3<64, 64, 2, 2, 1>(a, stream);
    }
  }
  {
    const int rc = launch_conv1x1_dma(a, stream);  // both operands by LDS-DMA: the one-tap layers of the f16x2 arithmetic
    if (rc != 1) return rc;
  }
  {
    const int rc = launch_igemm_x3ws(a, stream);  // wave-specialised form for the large layers
    if (rc != 1) return rc;
  }
  const int bn = (a.Cd <= 64) ? 64 : 128;
  const long long tn = ceil_div(a.Cd, bn);
  auto tiles = [&](int bm) { return (long long)ceil_div(a.M, bm) * tn; };
  // single LDS stage (48 KB at 128x128 => 2-3 workgroups per CU, whose split / MFMA phases interleave)
  // measured faster than a double-buffered stage at 1 workgroup per CU: 180 vs 157 TFLOP/s on 3x3x256 @128^2
  if (bn == 64) {
    if (tiles(128) >= 256) return launch_cfg3<128, 64, 2, 2, 1>(a, stream);
    return launch_cfg3<64, 64, 2, 2, 1>(a, stream);
  }
  if (tiles(128) >= 256) return launch_cfg3<128, 128, 2, 2, 1>(a, stream);
  if (tiles(64) >= 256) return launch_cfg3<64, 128, 2, 2, 1>(a, stream);
  // 16^2 maps (M = 4096 rows at batch 16): 64 x 64 tiles are the only ones that give every CU a workgroup
  return launch_cfg3<64, 64, 2, 2, 1>(a, stream);
}

// Weight planes for the split kernel.  Forward (`classes == nullptr` form): row co, k = (ky, kx, ci) as in
// the OHWI parameter.  out[pt][row][Kpad] bf16, zero padded along K.
__global__ void split_weight_fwd_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int rows, int K,
                                        int Kpad, const uint32_t* __restrict__ wscale) {
  split_fwd_body(w, out, rows, K, Kpad, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x,
                 wscale);
}

// Data-gradient planes of one residue class: row ci, k = (jy, jx, co) with ky = ky0 + jy*ksy, kx = kx0 + jx*ksx
// (the class-ordered layout of pack_dgrad_weight_kernel, produced straight from the OHWI parameter).
__global__ void split_weight_dgrad_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int Cout, int kh,
                                          int kw, int Cin, int ky0, int ksy, int nty, int kx0, int ksx, int ntx,
                                          int Kpad, const uint32_t* __restrict__ wscale) {
  split_dgrad_body(w, out, Cout, kh, kw, Cin, ky0, ksy, nty, kx0, ksx, ntx, Kpad,
                   (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x, wscale);
}

static inline int kpad32(int k) { return (k + 31) & ~31; }

}  // namespace evk

using namespace evk;

extern "C" size_t evk_conv2d_split_weight_bytes(const evk_conv_desc* d, int32_t for_dgrad) {
  if (!d) return 0;
  // (3x3: the LDS-halo kernel's layout pads the reduction channels to 16 per tap, the generic one the whole row to 32:
  // with channels % 16 != 0 the first can be the larger)
  auto k3 = [&](int K) { return d->kh == 3 && d->kw == 3 ? 9 * ((K + 15) / 16 * 16) : 0; };
  if (!for_dgrad) {
    const int kp = kpad32(d->kh * d->kw * d->Cin);
    return (size_t)3 * d->Cout * (kp > k3(d->Cin) ? kp : k3(d->Cin)) * sizeof(uint16_t);
  }
  if (d->stride_h == 1 && d->stride_w == 1 && d->kh == 3 && d->kw == 3) {
    const int kp = kpad32(9 * d->Cout);
    return (size_t)3 * d->Cin * (kp > k3(d->Cout) ? kp : k3(d->Cout)) * sizeof(uint16_t);
  }
  size_t total = 0;
  for (int cy = 0; cy < d->stride_h; ++cy)
    for (int cx = 0; cx < d->stride_w; ++cx) {
      const AxisPlan py = plan_axis(cy, d->pad_h, d->dil_h, d->stride_h, d->kh);
      const AxisPlan px = plan_axis(cx, d->pad_w, d->dil_w, d->stride_w, d->kw);
      total += (size_t)3 * d->Cin * kpad32(py.nt * px.nt * d->Cout) * sizeof(uint16_t);
    }
  return total;
}

// wscale == nullptr: the three bf16 planes; else the two fp16 planes of w / s(wscale) (planes 0 and 1 of the same layout)
static int split_weight_any(const evk_conv_desc* d, const float* w, int32_t for_dgrad, void* wsplit, const uint32_t* wscale,
                            void* stream) {
  EVK_REQUIRE(d && w && wsplit, EVK_E_INVALID, "split_weight: null pointer");
  EVK_REQUIRE(d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0, EVK_E_INVALID,
              "split_weight: bad stride/dilation");
  hipStream_t st = (hipStream_t)stream;
  uint16_t* out = reinterpret_cast<uint16_t*>(wsplit);
  // 3x3 'same' convolutions served by the LDS-halo kernel take their planes in that kernel's own order (the
  // decision is a pure function of the descriptor, so the consumer makes the same one); never larger than the
  // generic layout.
  if (wscale && conv_desc_uses_wino(d, for_dgrad ? 1 : 0))   // (12 x 2 planes of taps: inside the 9 x 3 the buffer is sized for)
    return launch_split_weight_wino(w, out, d->Cout, d->Cin, for_dgrad ? 1 : 0, st, wscale);
  if (d->kh == 3 && d->kw == 3 && conv_desc_uses_halo(d, for_dgrad ? 1 : 0))
    return launch_split_weight_halo(w, out, d->Cout, d->Cin, for_dgrad ? 1 : 0, st, wscale);
  if (!for_dgrad) {
    const int K = d->kh * d->kw * d->Cin, Kp = kpad32(K);
    const size_t total = (size_t)d->Cout * (Kp >> 1);
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(split_weight_fwd_kernel, dim3(blocks), dim3(256), 0, st, w, out, d->Cout, K, Kp, wscale);
    return check_launch("split_weight_fwd");
  }
  size_t off = 0;
  for (int cy = 0; cy < d->stride_h; ++cy)
    for (int cx = 0; cx < d->stride_w; ++cx) {
      const AxisPlan py = plan_axis(cy, d->pad_h, d->dil_h, d->stride_h, d->kh);
      const AxisPlan px = plan_axis(cx, d->pad_w, d->dil_w, d->stride_w, d->kw);
      const int K = py.nt * px.nt * d->Cout, Kp = kpad32(K);
      if (K > 0) {
        const size_t total = (size_t)d->Cin * (Kp >> 1);
        const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        hipLaunchKernelGGL(split_weight_dgrad_kernel, dim3(blocks), dim3(256), 0, st, w, out + off, d->Cout, d->kh,
                           d->kw, d->Cin, py.k0, py.kstep, py.nt, px.k0, px.kstep, px.nt, Kp, wscale);
        int rc = check_launch("split_weight_dgrad");
        if (rc) return rc;
      }
      off += (size_t)3 * d->Cin * Kp;
    }
  return EVK_OK;
}

extern "C" int evk_conv2d_split_weight(const evk_conv_desc* d, const float* w, int32_t for_dgrad, void* wsplit,
                                       void* stream) {
  return split_weight_any(d, w, for_dgrad, wsplit, nullptr, stream);
}
extern "C" int evk_conv2d_split_weight_f16x2(const evk_conv_desc* d, const float* w, int32_t for_dgrad, void* wsplit,
                                             const uint32_t* w_absmax_bits, void* stream) {
  EVK_REQUIRE(w_absmax_bits, EVK_E_INVALID, "split_weight_f16x2: null scale");
  return split_weight_any(d, w, for_dgrad, wsplit, w_absmax_bits, stream);
}
