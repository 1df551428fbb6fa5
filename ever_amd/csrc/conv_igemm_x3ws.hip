// Wave-specialised form of the bf16-split implicit-GEMM convolution (arithmetic: conv_igemm_x3.hip).
//
// One workgroup = 8 waves on one CU = 2 waves per SIMD with different jobs:
//   waves 0-3  "matrix" waves: ds_read_b128 fragments + v_mfma_f32_32x32x16_bf16, nothing else.  One per
//              SIMD, so each owns its SIMD's matrix pipe and issues MFMAs back to back.
//   waves 4-7  "staging" waves: im2col gather (global -> VGPR), exact 3-way bf16 split (~6 VALU per
//              element), ds_write_b128 of the three planes, and the pre-split weight planes
//              global -> VGPR -> LDS.  Their VALU / VMEM / LDS-write instructions issue from a different
//              wave than the MFMAs, so they fill the matrix pipe's shadow instead of serialising with
//              it (in the single-role kernel the split is a ~900-cycle MFMA-free phase per step:
//              measured 56 % matrix-pipe occupancy at 2 workgroups per CU).
// LDS is a ring of NSTAGE stages; stage s of step kt is written while the matrix waves read step kt's
// stage, one s_barrier per step for all 8 waves.
#include "igemm_common.hpp"
#include "x3_common.hpp"
#include <stdlib.h>
#include <type_traits>

namespace evk {

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int NPX>
__global__ __launch_bounds__(512) void conv_igemm_x3ws_kernel(const IGemmArgs p) {
  constexpr int NP = X3Mode<NPX>::NP;
  constexpr bool PK = X3Mode<NPX>::PK;   // the activation operand arrives packed (x3_common.hpp)
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MB = WM / 32, NB = WN / 32;
  constexpr int AR = BM / 64, BR = BN / 64;
  constexpr int kStage = 3 * (BM + BN) * kRowBytes;
  static_assert(WAVES_M * WAVES_N == 4, "4 matrix waves");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];

  // PERSISTENT: workgroup b walks tiles b, b + gridDim.x, ... and treats their K loops as ONE sequence of steps
  // g = 0 .. G-1 (G = my tiles x nk).  The LDS ring, the register sets and the one-barrier-per-step protocol run on
  // across tile boundaries, so while the matrix waves store tile t's accumulators the staging waves are already two
  // steps into tile t+1 — and the two roles' memory instructions are counted by different waves' vmcnt, so the
  // stores' drain never sits in front of a wait for new loads (which is what made persistent tiles useless in the
  // single-role kernel).  With gridDim.x == tiles it is the one-tile-per-workgroup kernel it used to be.
  const int ntiles = p.tiles_m * p.tiles_n;
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nk = p.Kpad / BK3;
  const int G = my_tiles * nk;
  const int tid = threadIdx.x;

  if (tid >= 256) {
    // ------------------------------------------------------------------ staging waves
    const int ptid = tid - 256;
    const int c4 = ptid & 3;   // 8-float group inside the K step
    const int rb = ptid >> 2;  // base row 0..63

    int a_y0[AR], a_x0[AR], a_base[AR];
    int b_off[BR];
    const int plane = p.Cd * p.Kpad;
    const int cp8 = p.Cs >> 3;
    int cc = 0, kx = 0, ky = 0;
    int lt_tile = 0, lt_kt = 0;  // load cursor: (index among my tiles, K step) of the next load_tiles call

    auto begin_tile = [&](int i) {
      const int bid = xcd_remap((int)blockIdx.x + i * (int)gridDim.x, ntiles);
      const int tile_n = bid % p.tiles_n;
      const int tile_m = bid / p.tiles_n;
      const int m0 = tile_m * BM, n0 = tile_n * BN;
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
#pragma unroll
      for (int j = 0; j < BR; ++j) {
        int co = n0 + rb + 64 * j;
        co = co < p.Cd ? co : p.Cd - 1;
        b_off[j] = co * p.Kpad + c4 * 8;
      }
      const int tap = c4 / cp8;
      cc = c4 - tap * cp8;
      ky = tap / p.kw;
      kx = tap - ky * p.kw;
    };

    // two register sets: the loads of step t+2 are issued before step t+1's registers are consumed, so a
    // global / L2 round trip has two full steps (~1.5 us) to land (with one set the staging waves sat
    // in s_waitcnt vmcnt for most of every step and the matrix waves waited for them at the barrier:
    // measured 209 -> 273 TFLOP/s on 3x3x256 @128^2 when the loads were removed)
    f32x4 ra[2][AR][2];
    u32x4 rbv[2][BR][3];
    uint32_t okmask[2] = {0, 0};
    float a_inv = 1.f;   // f16x2: 1 / activation scale
    if constexpr (NP == 2) a_inv = op_scale(act_absmax(p.a_scale)).inv;

    auto load_tiles = [&](auto SET) {
      constexpr int s = decltype(SET)::value;
      if (lt_kt == 0) begin_tile(lt_tile);
      const int kt = lt_kt;
      okmask[s] = 0;
      const bool kvalid = ky < p.kh;
      const int oy = ky * p.oys, ox = kx * p.oxs;
      const int tapoff = (oy * p.Ws + ox) * p.Cs + cc * 8;
#pragma unroll
      for (int j = 0; j < AR; ++j) {
        const int sy = a_y0[j] + oy, sx = a_x0[j] + ox;
        const bool ok = kvalid && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
        okmask[s] |= ok ? (1u << j) : 0u;
        const float* src = p.src + (ok ? a_base[j] + tapoff : 0);
        ra[s][j][0] = *reinterpret_cast<const f32x4*>(src);
        ra[s][j][1] = *reinterpret_cast<const f32x4*>(src + 4);
      }
#pragma unroll
      for (int j = 0; j < BR; ++j)
#pragma unroll
        for (int pt = 0; pt < NP; ++pt)
          rbv[s][j][pt] = *reinterpret_cast<const u32x4*>(p.wgt3 + (size_t)pt * plane + b_off[j] + kt * BK3);
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
      if (++lt_kt == nk) {
        lt_kt = 0;
        ++lt_tile;
      }
    };

    auto store_tiles = [&](auto SET, int stage) {
      constexpr int s = decltype(SET)::value;
      unsigned char* Ab = smem3 + stage * kStage;
      unsigned char* Bb = Ab + 3 * BM * kRowBytes;
#pragma unroll
      for (int j = 0; j < AR; ++j) {
        const int row = rb + 64 * j;
        const int off = plane_off(row, c4);
        const bool ok = (okmask[s] >> j) & 1u;
        u32x4 H, M, L;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 v = ra[s][j][e >> 1];
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
        const int off = plane_off(rb + 64 * j, c4);
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) *reinterpret_cast<u32x4*>(Bb + pt * BN * kRowBytes + off) = rbv[s][j][pt];
      }
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    static_assert(NSTAGE == 2, "staging schedule below is written for a two-stage ring");
    // step g lives in register set g & 1 and LDS stage g & 1.  Prologue: step 0 -> stage 0, steps 1 and 2 in flight.
    load_tiles(S0{});
    if (1 < G) load_tiles(S1{});
    store_tiles(S0{}, 0);
    if (2 < G) load_tiles(S0{});
    __syncthreads();
    // iteration g: the matrix waves read stage g & 1; step g+1 goes to the other stage and its
    // register set is refilled with step g+3
    for (int g = 0; g < G; g += 2) {
      if (g + 1 < G) {
        store_tiles(S1{}, 1);
        if (g + 3 < G) load_tiles(S1{});
      }
      __syncthreads();
      if (g + 1 < G) {
        if (g + 2 < G) {
          store_tiles(S0{}, 0);
          if (g + 4 < G) load_tiles(S0{});
        }
        __syncthreads();
      }
    }
    return;
  }

  // -------------------------------------------------------------------- matrix waves
  // issue arbitration on a SIMD is by priority, then age: the MFMA stream must never queue behind the
  // staging wave's VALU work
  __builtin_amdgcn_s_setprio(3);
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[MB][NB];

  int fa_off[MB][2], fb_off[NB][2];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fa_off[a][kk] = plane_off(wm * WM + a * 32 + li, 2 * kk + lh);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      fb_off[b][kk] = 3 * BM * kRowBytes + plane_off(wn * WN + b * 32 + li, 2 * kk + lh);

  float out_scale = 1.f;   // f16x2: activation scale x weight scale
  if constexpr (NP == 2) out_scale = op_scale(act_absmax(p.a_scale)).s * op_scale(*p.w_scale).s;
  bf16x8 fa[2][MB][3], fb[2][NB][3];  // fragment registers, double buffered across the two k-halves
  auto read_frags = [&](const unsigned char* S, int kk, int slot) {
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
        fa[slot][a][pt] = *reinterpret_cast<const bf16x8*>(S + pt * BM * kRowBytes + fa_off[a][kk]);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
        fb[slot][b][pt] = *reinterpret_cast<const bf16x8*>(S + pt * BN * kRowBytes + fb_off[b][kk]);
  };
  auto mfmas = [&](int slot) {
#pragma unroll
    for (int t = 0; t < X3Prod<NP>::N; ++t)
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a][b] = mfma_np<NP>(fb[slot][b][x3_pb(NP, t)], fa[slot][a][x3_pa(NP, t)], acc[a][b]);
  };

  __syncthreads();
  int g = 0;
  AmaxAcc amax_l{0u, p.out_amax != nullptr};
  for (int i = 0; i < my_tiles; ++i) {
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt, ++g) {
      const unsigned char* S = smem3 + (g & 1) * kStage;
      read_frags(S, 0, 0);
      read_frags(S, 1, 1);
      mfmas(0);
      mfmas(1);
      __syncthreads();
    }
    const int bid = xcd_remap((int)blockIdx.x + i * (int)gridDim.x, ntiles);
    if constexpr (NP == 2) igemm_scale_acc<MB, NB>(acc, out_scale);
    if (p.bn_part) {
      // statistics mode is launched one tile per workgroup: the staging waves are done (the epilogue's one barrier
      // must not meet a staging wave's step barrier, which it would in a persistent workgroup), every matrix wave
      // has passed the loop's last barrier, and the ring is free to serve as the epilogue's scratch
      igemm_epilogue_stats<MB, NB, WM, WN, WAVES_M, WAVES_N>(p, acc, (bid / p.tiles_n) * BM, (bid % p.tiles_n) * BN, wm, wn,
                                                             li, lh, reinterpret_cast<float*>(smem3 + p.bn_scratch_off));
      continue;
    }
    igemm_epilogue<MB, NB, WM, WN>(p, acc, (bid / p.tiles_n) * BM, (bid % p.tiles_n) * BN, wm, wn, li, lh,
                                   amax_l);
  }
  if (p.out_amax) amax_commit(p.out_amax, amax_l.m);   // once per wave, over all the tiles of this workgroup
}

// persistent workgroups, and the wave-specialised form also for short reductions (one tile per workgroup with the short
// reductions on the single-role kernel: 3-9 % behind on the K <= 512 layers with Cd >= 128)
static constexpr int ws_persist() { return 1; }

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int NP>
static int launch_ws_np(IGemmArgs& a, hipStream_t stream) {
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.Cd, BN);
  bn_stats_setup(a, BM, BN, WAVES_M, a.tiles_m);
  const size_t lds = (size_t)NSTAGE * 3 * (BM + BN) * kRowBytes;   // >= the statistics epilogue's scratch
  a.bn_scratch_off = 0;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_x3ws_kernel<BM, BN, WAVES_M, WAVES_N, NSTAGE, NP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const long long nwg = (long long)a.tiles_m * a.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffLL) {
    set_error("conv_igemm_x3ws: bad grid %lld", nwg);
    return EVK_E_INVALID;
  }
  // one workgroup per CU (LDS), each walking ceil(tiles / 256) tiles; 256 is a multiple of 8, so a workgroup's
  // tiles stay on its XCD under the remap
  // statistics mode: one tile per workgroup (the epilogue parks the tile in the LDS ring, which a persistent
  // workgroup's staging waves would already be refilling)
  const unsigned grid = (ws_persist() && nwg > 256 && !a.bn_part) ? 256u : (unsigned)nwg;
  hipLaunchKernelGGL((conv_igemm_x3ws_kernel<BM, BN, WAVES_M, WAVES_N, NSTAGE, NP>), dim3(grid), dim3(512), lds, stream, a);
  return check_launch("conv_igemm_x3ws");
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE>
static int launch_ws(IGemmArgs& a, hipStream_t stream) {
  if (a.planes == 1) return launch_ws_np<BM, BN, WAVES_M, WAVES_N, NSTAGE, 1>(a, stream);
  if (a.planes == 2)
    return a.a_packed ? launch_ws_np<BM, BN, WAVES_M, WAVES_N, NSTAGE, 4>(a, stream)
                      : launch_ws_np<BM, BN, WAVES_M, WAVES_N, NSTAGE, 2>(a, stream);
  return launch_ws_np<BM, BN, WAVES_M, WAVES_N, NSTAGE, 3>(a, stream);
}

// returns 1 when this form does not apply (caller falls back to the single-role kernel)
// tuning aid (EVK_TUNE=1, tools/autotune_convs.py): run the named tile shape whatever the heuristics below would pick
int launch_igemm_x3ws_forced(IGemmArgs& a, int bn, hipStream_t stream) {
  if (bn == 256) return launch_ws<128, 256, 2, 2, 2>(a, stream);
  if (bn == 128) return launch_ws<128, 128, 2, 2, 2>(a, stream);
  return launch_ws<128, 64, 2, 2, 2>(a, stream);
}

int launch_igemm_x3ws(IGemmArgs& a, hipStream_t stream) {
  // EVK_X3_WS: 0 never, 1 (default) where measured faster, 2 wherever the tile shapes allow
  static const int mode = getenv("EVK_X3_WS") ? atoi(getenv("EVK_X3_WS")) : 1;
  if (mode == 0) return 1;
  const int bn = (a.Cd <= 64) ? 64 : 128;
  const long long tn = ceil_div(a.Cd, bn);
  const long long t128 = (long long)ceil_div(a.M, 128) * tn;
  if (t128 < 256) {
    // 16^2 maps with wide outputs and a long reduction (stage-4 layers: 512->512 3x3, 2048<->512 1x1; M = 4096 rows):
    // 128 x 64 tiles still give every CU a workgroup, and the two-role form beats 64-row single-role tiles there
    // (tools/autotune_convs.py, same process: 64-65 vs 71-83 us on the 1x1 layers, 138 vs 152 us on the strided 3x3)
    if (mode >= 1 && a.Kpad >= 1024 && a.Cd >= 128 && (long long)ceil_div(a.M, 128) * ceil_div(a.Cd, 64) >= 256)
      return launch_ws<128, 64, 2, 2, 2>(a, stream);
    return 1;  // cannot fill the chip at one workgroup per CU
  }
  // One 8-wave workgroup per CU: nothing overlaps a tile's prologue / epilogue, so short reductions
  // (1x1 convolutions, K <= 512: 2..16 steps) run better as 2-3 single-role workgroups per CU, unless the
  // grid is below two per CU anyway.  Measured on the FarSeg-R50 layer set (tools/bench_conv_x3.py).
  // Persistent workgroups recover part of that (64->256 @128^2: 121 -> 111 us, 256->256: 267 -> 251, 512->256 @64^2:
  // 106 -> 99; tools/ab_conv1x1.py) except for 64-wide outputs (256->64 @128^2: 80 -> 95 us).
  // An accumulate / residual epilogue (one more load per store, in the matrix waves) turns the gain into a loss:
  // folded-BatchNorm inference 1370 -> 1325 tiles/s.
  if (mode == 1 && a.Kpad < 1024 && t128 >= 512 && (!ws_persist() || a.Cd < 128 || a.accum)) return 1;
  // with the statistics epilogue the workgroups are not persistent: on the 128^2 maps (>= 2048 tiles, K <= 256) the
  // single-role kernel is ahead again (64->256: 129 -> 117 us, 256->128: 158 -> 140), on the smaller maps it is not
  if (mode == 1 && a.bn_want && a.Kpad < 1024 &&
      (long long)ceil_div(a.M, 128) * ceil_div(a.Cd, a.Cd >= 256 ? 256 : 128) >= 2048)
    return 1;
  // 128x256 tiles where the output is wide enough: the activation split (VALU) and the L2 -> CU bytes per MFMA
  // drop by half / a fifth
  if (a.Cd >= 256 && (long long)ceil_div(a.M, 128) * ceil_div(a.Cd, 256) >= 256)
    return launch_ws<128, 256, 2, 2, 2>(a, stream);
  if (bn == 128) return launch_ws<128, 128, 2, 2, 2>(a, stream);
  return launch_ws<128, 64, 2, 2, 2>(a, stream);
}

}  // namespace evk
