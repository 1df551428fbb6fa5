// One-tap (1x1, any stride) convolutions of the f16x2 arithmetic, forward and data gradient, SOFTWARE-PIPELINED with LOADER
// WAVES:   dst[m][co] = sum_k src[row(m)][k] * w[co][k]      (arithmetic: conv_igemm_x3.hip / x3_common.hpp)
//
// Round 5.  conv1x1_dma.hip moves both operands global -> LDS by DMA, but its K step is
//     wait(DMA) | barrier | issue DMA | read 6 fragments | WAIT | 6 MFMA | read 6 fragments | WAIT | 6 MFMA
// (seen in its ISA: four exposed `s_waitcnt lgkmcnt(0)` per step): the LDS latency, the DMA issue (~60-100 cycles per
// instruction in the issuing wave's in-order stream) and the barrier all ADD to the 12 MFMAs.  Measured on 1024 -> 256 @32^2
// (tools/ab_c1_small.py, us): everything 35.9, the K loop with no DMA and no stores 26.0 — against 10.3 for its MFMAs at
// the matrix pipe's nominal rate; the DMA alone 19.1.  Two more facts from tools/probes/dma_issue.hip + dma_patterns.hip:
// a CU takes ~32-40 outstanding 1 KB vector-memory instructions, the wave that issues the next one BLOCKS until one
// returns; and the streaming rate of LDS-DMA scales with the number of ISSUING waves (2 waves: 3.0 TB/s chip-wide at any
// depth, 4: 5.8, 8: 7.1), not with the access pattern (128-byte row segments 4 KB apart = whole rows).  Hence:
//  * the eight compute waves (4 x 2, 32 x BN/2 each) issue NO vector-memory instruction inside the K loop; four LOADER waves
//    issue the whole ring (activation tile as raw 4-byte words in 128-byte rows, weights as their two fp16 planes in 64-byte
//    rows, the XOR swizzles on the source side: the same LDS images as conv1x1_dma.hip) and absorb the queue's back-pressure;
//  * the fragment registers are double-buffered ACROSS the barrier: a step is
//        read k-half 1 | MFMAs of k-half 0 | barrier (stage kt+1 has landed) | read k-half 0 of stage kt+1 | MFMAs of k-half 1
//    so every fragment read has a whole MFMA group (192 cycles) to return and the only wait left is the barrier itself; this
//    needs stage kt+1 complete one half step early: ring of NST = 4 stages, NST - 1 in flight;
//  * no run-time ablation switch inside the loop (compile-time EVK_SP_ABL: 1 no loader DMA, 4 no compute, 8 no stores).
// Epilogues: the shared ones of igemm_common.hpp; the loader waves end before them (an ended wave is not waited for).
#include "igemm_common.hpp"
#include "x3_common.hpp"
#include "lds_dma.hpp"
#include <stdlib.h>
#include <string.h>

#ifndef EVK_SP_ABL
#define EVK_SP_ABL 0
#endif

namespace evk {

namespace {

constexpr int kSpRow = BK3 * 4;  // bytes of one activation row of a K step (32 four-byte words)
constexpr int kSpCW = 8;         // compute waves

__device__ __forceinline__ void sp_dma16s(i32x4 rsrc, uint32_t lds_byte, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
               :: "v"(voff), "s"(lds_byte), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ u32x4 sp_lds_read16(uint32_t lds_byte) {
  return *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)lds_byte;
}
// (ablations 16: no barrier; 32: the fragments are read once, in front of the loop; 64: no operand split / byte permutes)
__device__ __forceinline__ void sp_barrier() { if (!(EVK_SP_ABL & 16)) ring_barrier(); }
__device__ __forceinline__ int sp_arow_off(int row, int c) { return row * kSpRow + ((c ^ ((row >> 1) & 7)) << 4); }

}  // namespace

// LW loader waves (4); NST ring stages
template <int BN, bool PK, int NST, int LW>
__global__ __launch_bounds__(64 * (kSpCW + LW)) void conv1x1_sp_kernel(const IGemmArgs p, uint32_t src_bytes, uint32_t wgt_bytes) {
  constexpr int BM = 128, WAVES_M = 4, WAVES_N = 2, WM = 32, WN = BN / 2, NB = WN / 32;
  constexpr int kAStage = BM * kSpRow, kBPlane = BN * kRowBytes, kStage = kAStage + 2 * kBPlane;
  constexpr int AI = kAStage / 1024 / LW;        // activation DMA instructions per loader wave and stage (8 rows each)
  constexpr int BI = 2 * kBPlane / 1024 / LW;    // weight-plane DMA instructions per loader wave and stage
  constexpr int PER = AI + BI;
  static_assert(LW == 4 && AI >= 1 && BI >= 1 && NB >= 1 && NST >= 3, "tile shape");
  static_assert((NST - 2) * PER <= 62, "vmcnt range");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_sp[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap((int)blockIdx.x, ntiles);
  const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = p.Kpad / BK3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_sp;

  if (wave >= kSpCW) {
    // ================================================================== loader waves
    const int lw = wave - kSpCW;
    const i32x4 rs_a = make_rsrc(p.src, src_bytes), rs_b = make_rsrc(p.wgt3, wgt_bytes);
    uint32_t a_voff[AI], b_voff[BI];
#pragma unroll
    for (int t = 0; t < AI; ++t) {
      const int row = 8 * (AI * lw + t) + (lane >> 3);    // row of the tile this lane's 16 bytes belong to
      const int c = (lane & 7) ^ ((row >> 1) & 7);        // source chunk that lands on LDS chunk (lane & 7)
      const int m = m0 + row;
      uint32_t off = kDmaOOB;
      if (m < p.M) {
        const int hw = p.Hm * p.Wm;
        const int n = m / hw;
        const int rem = m - n * hw;
        const int gy = rem / p.Wm;
        const int gx = rem - gy * p.Wm;
        const int sy = gy * p.ash + p.oy0, sx = gx * p.asw + p.ox0;
        if ((unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws)
          off = (uint32_t)(((n * p.Hs + sy) * p.Ws + sx) * p.Cs) * 4u + (uint32_t)c * 16u;
      }
      a_voff[t] = off;
    }
    const uint32_t plane_bytes = (uint32_t)p.Cd * (uint32_t)p.Kpad * 2u;
#pragma unroll
    for (int t = 0; t < BI; ++t) {
      const int s = 64 * (BI * lw + t) + lane;  // 16-byte slot among the stage's 2 * BN * 4 weight slots
      const int pt = s / (BN * 4);
      const int row = (s - pt * BN * 4) >> 2;
      const int c = (s & 3) ^ ((row >> 2) & 3);
      const int co = n0 + row;
      b_voff[t] = co < p.Cd ? (uint32_t)pt * plane_bytes + (uint32_t)co * (uint32_t)p.Kpad * 2u + (uint32_t)c * 16u : kDmaOOB;
    }
    auto issue = [&](int kt, uint32_t S) {
      if (EVK_SP_ABL & 1) return;
      const uint32_t ka = (uint32_t)kt * kSpRow, kb = (uint32_t)kt * kRowBytes;
#pragma unroll
      for (int t = 0; t < AI; ++t) sp_dma16s(rs_a, S + (AI * lw + t) * 1024, a_voff[t], ka);
#pragma unroll
      for (int t = 0; t < BI; ++t) sp_dma16s(rs_b, S + kAStage + (BI * lw + t) * 1024, b_voff[t], kb);
    };
    // prologue: stages 0 .. NST-2 in flight
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nk) issue(s, lds0 + s * kStage);
    // stage 0 landed: at most the later prologue stages outstanding (fewer than NST - 2 of them exist when nk is short: the
    // count is then an over-estimate of what may stay in flight only if nk - 1 < NST - 2, handled by the full wait)
    if (nk - 1 >= NST - 2) wait_vmcnt<(NST - 2) * PER>(); else wait_vmcnt<0>();
    sp_barrier();                                   // P0
    uint32_t S_i = lds0 + (NST - 1) * kStage;         // slot of stage kt + NST - 1 (= the slot stage kt - 1 has left)
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + NST - 1 < nk) issue(kt + NST - 1, S_i);
      S_i = S_i == lds0 + (NST - 1) * kStage ? lds0 : S_i + kStage;
      // stage kt + 1 landed: the younger stages may stay in flight — NST - 2 of them, fewer in the tail
      const int younger = nk - 2 - kt;                // stages kt + 2 .. nk - 1 exist
      if (younger >= NST - 2) wait_vmcnt<(NST - 2) * PER>();
      else if (NST > 3 && younger == 1) wait_vmcnt<PER>();
      else wait_vmcnt<0>();
      sp_barrier();                                 // B(kt)
    }
    return;
  }

  // ==================================================================== compute waves
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  uint32_t fa_off[2][2], fb_off[2];   // fragment read offsets inside a stage (lane constants)
  {
    const int row = wm * WM + li;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int h = 0; h < 2; ++h) fa_off[kk][h] = (uint32_t)sp_arow_off(row, 4 * kk + 2 * lh + h);
    const int brow = wn * WN + li;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fb_off[kk] = (uint32_t)(kAStage + plane_off(brow, 2 * kk + lh));
  }
  float a_inv, out_scale;
  {
    const OpScale sa = op_scale(act_absmax(p.a_scale)), sw = op_scale(*p.w_scale);
    a_inv = sa.inv;
    out_scale = sa.s * sw.s;
  }
  f32x16 acc[1][NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;

  struct Frag {
    u32x4 a0, a1;            // raw activation words: 8 consecutive k of this lane's row
    bf16x8 b[NB][2];         // weight planes h, l
  };
  auto read_frag_now = [&](uint32_t S, int kk, Frag& f) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
        f.b[b][pt] = __builtin_bit_cast(bf16x8, sp_lds_read16(S + fb_off[kk] + pt * kBPlane + b * 32 * kRowBytes));
    f.a0 = sp_lds_read16(S + fa_off[kk][0]);
    f.a1 = sp_lds_read16(S + fa_off[kk][1]);
  };
  auto read_frag = [&](uint32_t S, int kk, Frag& f) {
    if (!(EVK_SP_ABL & 32)) read_frag_now(S, kk, f);
  };
  auto mma = [&](const Frag& f) {
    // (read as floats: a bit_cast of an ext-vector ELEMENT is miscompiled by this hipcc; conv1x1_dma.hip)
    const f32x4 w0 = __builtin_bit_cast(f32x4, f.a0), w1 = __builtin_bit_cast(f32x4, f.a1);
    u32x4 H, L;
    uint32_t h, l, unused = 0;
    if (EVK_SP_ABL & 64) {
      const bf16x8 fr[2] = {__builtin_bit_cast(bf16x8, f.a0), __builtin_bit_cast(bf16x8, f.a1)};
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[0][b] = mfma_np<2>(f.b[b][kHB[t]], fr[kHA[t]], acc[0][b]);
      return;
    }
    split_op<2, PK>(w0.x, w0.y, a_inv, h, l, unused); H[0] = h; L[0] = l;
    split_op<2, PK>(w0.z, w0.w, a_inv, h, l, unused); H[1] = h; L[1] = l;
    split_op<2, PK>(w1.x, w1.y, a_inv, h, l, unused); H[2] = h; L[2] = l;
    split_op<2, PK>(w1.z, w1.w, a_inv, h, l, unused); H[3] = h; L[3] = l;
    const bf16x8 fa[2] = {__builtin_bit_cast(bf16x8, H), __builtin_bit_cast(bf16x8, L)};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[0][b] = mfma_np<2>(f.b[b][kHB[t]], fa[kHA[t]], acc[0][b]);
  };

  Frag fx, fy;
  sp_barrier();                                     // P0: stage 0 has landed
  uint32_t S_c = lds0;
  if (!(EVK_SP_ABL & 4)) read_frag_now(opaque(S_c), 0, fx);
  if (EVK_SP_ABL & 32) read_frag_now(opaque(S_c), 1, fy);
  // (the last step is peeled: a conditional read of the next stage would merge two LDS-counter states behind the barrier and
  // make hipcc wait for the reads just issued — seen in the ISA as lgkmcnt(1) in front of the second MFMA group)
  for (int kt = 0; kt + 1 < nk; ++kt) {
    const uint32_t S = opaque(S_c);
    S_c = S_c == lds0 + (NST - 1) * kStage ? lds0 : S_c + kStage;
    const uint32_t Sn = opaque(S_c);
    if (EVK_SP_ABL & 4) {
      sp_barrier();
      continue;
    }
    read_frag(S, 1, fy);
    __builtin_amdgcn_sched_barrier(0);
    mma(fx);
    __builtin_amdgcn_sched_barrier(0);
    sp_barrier();                                   // B(kt): stage kt + 1 has landed; fy has returned long ago
    read_frag(Sn, 0, fx);
    __builtin_amdgcn_sched_barrier(0);
    mma(fy);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (EVK_SP_ABL & 4) {
    sp_barrier();
  } else {
    read_frag(opaque(S_c), 1, fy);
    __builtin_amdgcn_sched_barrier(0);
    mma(fx);
    __builtin_amdgcn_sched_barrier(0);
    sp_barrier();                                   // B(nk - 1): the loader waves' last barrier
    mma(fy);
  }
  if (EVK_SP_ABL & 8) {   // (every accumulator register stays live: nothing of the loop may be optimised away)
    float sum = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[0][b][r];
    if (sum == 12345.f) p.dst[0] = 0.f;
    return;
  }
  igemm_scale_acc<1, NB>(acc, out_scale);
  if (p.bn_part) {
    __syncthreads();   // the ring becomes the statistics epilogue's scratch: every compute wave is done reading the last stage
    igemm_epilogue_stats<1, NB, WM, WN, WAVES_M, WAVES_N>(p, acc, m0, n0, wm, wn, li, lh, reinterpret_cast<float*>(smem_sp));
    return;
  }
  AmaxAcc amax_l{0u, p.out_amax != nullptr};
  igemm_epilogue<1, NB, WM, WN>(p, acc, m0, n0, wm, wn, li, lh, amax_l);
  if (p.out_amax) amax_commit(p.out_amax, amax_l.m);
}

template <int BN, bool PK, int NST>
static int launch_sp(IGemmArgs& a, hipStream_t stream) {
  constexpr int BM = 128, LW = 4;
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.Cd, BN);
  bn_stats_setup(a, BM, BN, 4, a.tiles_m);
  size_t lds = (size_t)NST * (BM * kSpRow + 2 * BN * kRowBytes);
  const size_t scratch = ((size_t)2 * 4 * 32 * (BN / 2 + 4) + (size_t)3 * 2 * 3 * (BN / 2)) * sizeof(float);
  if (lds < scratch) lds = scratch;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_sp_kernel<BN, PK, NST, LW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const long long nwg = (long long)a.tiles_m * a.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffLL) {
    set_error("conv1x1_sp: bad grid %lld", nwg);
    return EVK_E_INVALID;
  }
  const unsigned long long sb = (unsigned long long)a.N * a.Hs * a.Ws * a.Cs * 4ull;
  const unsigned long long wb = 2ull * a.Cd * a.Kpad * 2ull;
  hipLaunchKernelGGL((conv1x1_sp_kernel<BN, PK, NST, LW>), dim3((unsigned)nwg), dim3(64 * (kSpCW + LW)), lds, stream, a, (uint32_t)sb,
                     (uint32_t)wb);
  return check_launch("conv1x1_sp");
}

bool conv1x1_sp_applicable(const IGemmArgs& a) { return conv1x1_dma_applicable(a) && a.Kpad / BK3 >= 2; }

// bn: 128 / 64 = column tile with a ring of four stages; 3128 / 3064 = three stages
int launch_conv1x1_sp_forced(IGemmArgs& a, int bn, hipStream_t stream) {
  if (!conv1x1_sp_applicable(a)) {
    set_error("conv1x1_sp: shape not supported (1x1, Cs %% 32 == 0, Cs >= 64, f16x2 arithmetic)");
    return EVK_E_UNSUPPORTED;
  }
  if (bn == 128) return a.a_packed ? launch_sp<128, true, 4>(a, stream) : launch_sp<128, false, 4>(a, stream);
  if (bn == 64) return a.a_packed ? launch_sp<64, true, 4>(a, stream) : launch_sp<64, false, 4>(a, stream);
  if (bn == 3128) return a.a_packed ? launch_sp<128, true, 3>(a, stream) : launch_sp<128, false, 3>(a, stream);
  return a.a_packed ? launch_sp<64, true, 3>(a, stream) : launch_sp<64, false, 3>(a, stream);
}

}  // namespace evk
