// One-tap (1x1, any stride) convolutions of the f16x2 arithmetic, forward and data gradient, as a PERSISTENT kernel with a
// STORE ROLE:   dst[m][co] = sum_k src[row(m)][k] * w[co][k]      (arithmetic: conv_igemm_x3.hip / x3_common.hpp)
//
// conv1x1_dma.hip moves both operands global -> LDS by DMA, but its workgroup is "K loop, then a burst of 64 KB of stores".
// Ablated on 256 -> 256 @128^2 (tools/ab_c1dma.py, us): loads alone 75, stores alone 57, loads + stores WITHOUT any compute
// 159, compute alone 85 (the MFMAs themselves: 47), everything 192 — a tile's stores and the next tile's loads share one
// in-order vector-memory path per CU and one in-order vmcnt per wave, and two co-resident workgroups do not find the
// opposite phase by themselves.  Here
//  * one workgroup per CU walks a run of ROW tiles of ONE column tile (row tiles g, g + groups, ...: the tiles in flight are
//    neighbours in memory); the workgroups that take the other column tiles of the same rows sit on the same XCD (workgroup
//    id mod 8) and walk in step, so the activation rows come from HBM once and are L2 hits for the others (one workgroup
//    doing a row's column tiles one after the other found them evicted: loads alone 113 us against 76); the DMA ring (three
//    stages of 32 KB, two K steps in flight) runs ACROSS tile boundaries, so a tile has no prologue;
//  * eight compute waves (4 x 2, two per SIMD) issue the DMA, read fragments and run the MFMAs as in conv1x1_dma.hip, the
//    fragment reads of one k-half issued under the MFMAs of the other, the two waves of a SIMD half a step apart; a
//    finished tile goes accumulator -> LDS (64 KB staging image, XOR-swizzled rows) and the waves go straight on;
//  * four STORE waves, which never load, drain the staging image during the NEXT tile's K steps, a slice per step:
//    row-contiguous 16-byte reads, bias / ReLU / BatchNorm statistics on the way, 256 contiguous bytes per row and four rows
//    per store instruction.  Loads and stores reach the memory path interleaved at K-step granularity.
// One s_barrier per K step, shared by all twelve waves (gfx950 has no named barriers): the store waves run the same step
// sequence and do their slice between two of them.
//
// What was measured on the way (same 256 -> 256 layer; compile-time ablations, tools/build_variant.sh -DEVK_PS_ABL=bits):
// the barrier skeleton alone 14 us, MFMAs alone 56, fragment reads alone 49, both 100-105 — a ds_read_b128 return takes
// ~6 cycles of its SIMD's matrix pipe even without a dependency (tools/probes/mfma_lds_mix.hip: 12 reads per 12 MFMAs = +19 %
// at two waves per SIMD, +40 % at one), and this tile shape reads one fragment per MFMA; run-time ablation switches cost more
// than what they measure (their branches: ~30 cycles per taken branch, tools/probes/barrier_cost.hip).  Moving the DMA issue
// to two dedicated loader waves (8 + 2 + 2) gained 5 % on this layer and lost 25 % on the store-bound 64 -> 256 (two store
// waves): not kept.  Result (us, best other form -> this): 64 -> 256 @128^2 107 -> 90, 256 -> 128 104 -> 94, 256 -> 256
// 175 -> 170, 128 -> 512 @64^2 52 -> 47 (without statistics); level or behind on the 32^2 / 16^2 maps (one or two tiles per
// workgroup: nothing to overlap) — the dispatcher gives it the 128^2-map layers only.
#include "igemm_common.hpp"
#include "x3_common.hpp"
#include "lds_dma.hpp"
#include <stdlib.h>
#include <string.h>

namespace evk {

namespace {

constexpr int kPsRow = BK3 * 4;      // bytes of one activation row of a K step (32 four-byte words)
constexpr int kPsBM = 128, kPsBN = 128, kPsNST = 3;
constexpr int kPsAStage = kPsBM * kPsRow, kPsBPlane = kPsBN * kRowBytes, kPsStage = kPsAStage + 2 * kPsBPlane;
constexpr int kPsRing = kPsNST * kPsStage;
constexpr int kPsOutRow = kPsBN * 4;                 // bytes of one staged output row
constexpr int kPsStaging = kPsBM * kPsOutRow;        // 64 KB
constexpr int kPsLds = kPsRing + kPsStaging;         // 160 KB: the whole LDS of a CU
constexpr int kPsCW = 8, kPsSW = 4;                  // compute / store waves
constexpr int kPsWaves = kPsCW + kPsSW;
constexpr int kPsInstr = 16;                         // store instructions (4 rows x 256 B) per store wave and tile

__device__ __forceinline__ void ps_dma16s(i32x4 rsrc, uint32_t lds_byte, uint32_t voff, uint32_t soff) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 ps_lds_read16(uint32_t lds_byte) {
  return *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)lds_byte;
}
__device__ __forceinline__ void ps_lds_write16(uint32_t lds_byte, f32x4 v) {
  *(__attribute__((address_space(3))) f32x4*)(uintptr_t)lds_byte = v;
}
__device__ __forceinline__ int ps_arow_off(int row, int c) { return row * kPsRow + ((c ^ ((row >> 1) & 7)) << 4); }
// staged output: 16-byte chunk c (0..31) of row `row`; the XOR gives the 8 consecutive rows a ds_write_b128 lane group
// covers 8 distinct 16-byte slots of a 128-byte window, and leaves every aligned run of 16 chunks of a row a permutation of
// itself (the readers' ds_read_b128 lane groups take one such run each)
__device__ __forceinline__ uint32_t ps_out_off(int row, int c) { return (uint32_t)(row * kPsOutRow + ((c ^ (row & 7)) << 4)); }

}  // namespace

template <bool PK, bool STATS, bool DBG>
__global__ __launch_bounds__(64 * kPsWaves) void conv1x1_ps_kernel(const IGemmArgs p, uint32_t src_bytes, uint32_t wgt_bytes,
                                                                   int dbg_arg) {
#ifndef EVK_PS_ABL
#define EVK_PS_ABL 0                 // (tools/build_variant.sh: the same ablation bits as compile-time constants, for timings
#endif                               //  that do not carry the run-time switches' own branches)
  const int dbg = DBG ? dbg_arg : EVK_PS_ABL;   // (ablation switches compiled out of the production instantiations)
  constexpr int BM = kPsBM, BN = kPsBN, WM = 32, WN = 64, NB = 2;
  constexpr int AI = kPsAStage / 1024 / kPsCW;        // activation DMA instructions per compute wave and step (8 rows each)
  constexpr int BI = 2 * kPsBPlane / 1024 / kPsCW;    // weight-plane DMA instructions per compute wave and step
  constexpr int PER = AI + BI;
  static_assert(AI == 2 && BI == 2, "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_ps[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // tile order: XCD x = id & 7 holds gridDim / 8 workgroups; tiles_n of them form a group that walks the same row tiles
  const int slots = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int gpx = slots / p.tiles_n;                   // groups per XCD
  const int tile_n = slot % p.tiles_n, grp = xcd * gpx + slot / p.tiles_n;
  // row tiles grp, grp + ngroups, ...: the tiles in flight at any moment are neighbours in memory
  const int ngroups = 8 * gpx;
  const int nmine = grp < p.tiles_m ? (p.tiles_m - grp + ngroups - 1) / ngroups : 0;
  if (nmine <= 0) return;                              // (the whole workgroup: no barrier is left waiting)
  const int n0 = tile_n * BN;
  const int nk = p.Kpad / BK3;                         // >= 2 (conv1x1_ps_applicable)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_ps;
  const uint32_t stg0 = lds0 + kPsRing;
  // Every role runs the same sequence of barriers: nk per tile, + 2 at the end (the last tile parked; the last tile staged).
  // The K-step loops are kept free of tile bookkeeping — a taken branch costs ~30 cycles on this part (tools/probes/
  // barrier_cost.hip) against ~800 cycles of matrix work per step.

  if (wave < kPsCW) {
    // ================================================================ compute waves
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave & 3, wn = wave >> 2;
    const i32x4 rs_a = make_rsrc(p.src, src_bytes), rs_b = make_rsrc(p.wgt3, wgt_bytes);
    const uint32_t plane_bytes = (uint32_t)p.Cd * (uint32_t)p.Kpad * 2u;
    uint32_t a_voff[AI], a_voff_nx[AI], b_voff[BI];
#pragma unroll
    for (int t = 0; t < BI; ++t) {     // weights: the same column tile for every row tile of this workgroup
      const int s = 64 * (BI * wave + t) + lane;   // 16-byte slot among the stage's 2 * BN * 4 weight slots
      const int pt = s / (BN * 4);
      const int row = (s - pt * BN * 4) >> 2;
      const int c = (s & 3) ^ ((row >> 2) & 3);
      const int co = n0 + row;
      b_voff[t] = co < p.Cd ? (uint32_t)pt * plane_bytes + (uint32_t)co * (uint32_t)p.Kpad * 2u + (uint32_t)c * 16u : kDmaOOB;
    }
    auto tile_offsets = [&](int item, uint32_t (&voff)[AI]) {
      const int m0 = item * BM;
#pragma unroll
      for (int t = 0; t < AI; ++t) {
        const int row = 8 * (AI * wave + t) + (lane >> 3);   // row of the tile this lane's 16 bytes belong to
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
        voff[t] = off;
      }
    };
    uint32_t S_i = lds0 + 2 * kPsStage;   // the slot two steps ahead of the one being computed
    auto issue = [&](const uint32_t (&voff)[AI], int kt_i, uint32_t S) {   // DMA of K step kt_i of a tile into slot S
      const uint32_t ka = (uint32_t)kt_i * kPsRow, kb = (uint32_t)kt_i * kRowBytes;
      if (!(dbg & 1)) {
#pragma unroll
        for (int t = 0; t < AI; ++t) ps_dma16s(rs_a, S + (AI * wave + t) * 1024, voff[t], ka);
      }
      if (!(dbg & 2)) {
#pragma unroll
        for (int t = 0; t < BI; ++t) ps_dma16s(rs_b, S + kPsAStage + (BI * wave + t) * 1024, b_voff[t], kb);
      }
    };
    uint32_t fa_off[2][2], fb_off[2];   // fragment read offsets inside a stage (lane constants)
    {
      const int row = wm * WM + li;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h) fa_off[kk][h] = (uint32_t)ps_arow_off(row, 4 * kk + 2 * lh + h);
      const int brow = wn * WN + li;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fb_off[kk] = (uint32_t)(kPsAStage + plane_off(brow, 2 * kk + lh));
    }
    // staging offsets of this lane's accumulator quads: chunk = wn * 16 + b * 8 + (2 r4 + lh), and the row's XOR only touches
    // the low three bits, so block b is a 128-byte immediate on top of four per-lane values
    uint32_t st_off[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) st_off[r4] = stg0 + ps_out_off(wm * WM + li, wn * 16 + 2 * r4 + lh);

    float a_inv, out_scale, bias_max = 0.f;
    {
      const OpScale sa = op_scale(act_absmax(p.a_scale)), sw = op_scale(*p.w_scale);
      a_inv = sa.inv;
      out_scale = sa.s * sw.s;
      if (p.bias != nullptr && p.out_amax != nullptr) {   // max |bias| over this column tile, the same in every lane
        const int c0 = n0 + 2 * lane;
        float m = 0.f;
        if (c0 < p.Cd) m = fabsf(p.bias[c0]);
        if (c0 + 1 < p.Cd) m = fmaxf(m, fabsf(p.bias[c0 + 1]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        bias_max = m;
      }
    }

    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    struct Frag {
      u32x4 a0, a1;            // raw activation words: 8 consecutive k of this lane's row
      bf16x8 b[NB][2];         // weight planes h, l
    };
    auto read_frag = [&](uint32_t S, int kk, Frag& f) {
      if (dbg & 128) return;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
          f.b[b][pt] = __builtin_bit_cast(bf16x8, ps_lds_read16(S + fb_off[kk] + pt * kPsBPlane + b * 32 * kRowBytes));
      f.a0 = ps_lds_read16(S + fa_off[kk][0]);
      f.a1 = ps_lds_read16(S + fa_off[kk][1]);
    };
    auto mma = [&](const Frag& f) {
      // (read as floats: a bit_cast of an ext-vector ELEMENT is miscompiled by this hipcc; conv1x1_dma.hip)
      const f32x4 w0 = __builtin_bit_cast(f32x4, f.a0), w1 = __builtin_bit_cast(f32x4, f.a1);
      u32x4 H, L;
      uint32_t h, l, unused = 0;
      split_op<2, PK>(w0.x, w0.y, a_inv, h, l, unused); H[0] = h; L[0] = l;
      split_op<2, PK>(w0.z, w0.w, a_inv, h, l, unused); H[1] = h; L[1] = l;
      split_op<2, PK>(w1.x, w1.y, a_inv, h, l, unused); H[2] = h; L[2] = l;
      split_op<2, PK>(w1.z, w1.w, a_inv, h, l, unused); H[3] = h; L[3] = l;
      const bf16x8 fa[2] = {__builtin_bit_cast(bf16x8, H), __builtin_bit_cast(bf16x8, L)};
      if (dbg & 64) {
        acc[0][0] += __builtin_bit_cast(float, H[0] ^ L[1] ^ H[2] ^ L[3]) + (float)f.b[0][0][0] + (float)f.b[1][1][0] + (float)f.b[0][1][1] + (float)f.b[1][0][1];
        return;
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = mfma_np<2>(f.b[b][kHB[t]], fa[kHA[t]], acc[b]);
    };
    // the output's operand-scale maximum is taken here, from the accumulators (a store wave has no instruction to spare);
    // a bias is added by the store waves afterwards, so it enters as max|bias| over the tile's columns at the end
    uint32_t amax_m = 0;
    auto park_tile = [&]() {           // accumulators -> staging image, then start the next tile from zero
      if (!(dbg & 16)) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 v = {acc[b][4 * r4] * out_scale, acc[b][4 * r4 + 1] * out_scale, acc[b][4 * r4 + 2] * out_scale,
                             acc[b][4 * r4 + 3] * out_scale};
            ps_lds_write16(st_off[r4] + b * 128, v);
            amax_m = max(amax_m, max(max(__builtin_bit_cast(uint32_t, v.x) & 0x7fffffffu, __builtin_bit_cast(uint32_t, v.y) & 0x7fffffffu),
                                     max(__builtin_bit_cast(uint32_t, v.z) & 0x7fffffffu, __builtin_bit_cast(uint32_t, v.w) & 0x7fffffffu)));
          }
      }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    };

    // One step, two phase orders.  Waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically), and
    // the barrier releases them together: with the same order in both, their fragment reads + operand permutes coincide and
    // their MFMA groups coincide — ~150 cycles per k-half in which the SIMD's matrix pipe has no taker (compile-time
    // ablation, 256 -> 256 @128^2: MFMAs alone 56 us, fragment reads alone 49, both 105).  So the row waves of column half 0
    // ("early") run       reads k-half 0 | MFMAs of the previous step's k-half 1 | reads k-half 1 | MFMAs k-half 0
    // and those of column half 1 ("late"), one k-half behind,
    //                     MFMAs of the previous step's k-half 0 | reads k-half 0 | MFMAs previous k-half 1 | reads k-half 1:
    // one wave of a SIMD reads while the other multiplies.  Both read a stage only during its own step (the ring protocol is
    // the same), both refill a fragment set right after the MFMAs that consumed it (two sets each).
    uint32_t S_c = lds0;               // ring slot of the step being computed (S_i: two steps ahead)
    Frag fx, fy;
    auto next_slot = [&]() {
      S_c = S_c == lds0 + 2 * kPsStage ? lds0 : S_c + kPsStage;
      S_i = S_i == lds0 + 2 * kPsStage ? lds0 : S_i + kPsStage;
    };
    auto compute_early = [&](bool first_of_tile, bool park) {
      const uint32_t S = opaque(S_c);
      if (dbg & 4) {
        if (park) park_tile();
      } else {
        // (scheduling fences: left alone, hipcc sinks each fragment read to just above its first use and waits lgkmcnt(0)
        // there — the LDS latency exposed three or four times per step)
        read_frag(S, 0, fx);
        __builtin_amdgcn_sched_barrier(0);
        if (!first_of_tile || park) mma(fy);             // second k-half of the previous step, under the reads just issued
        if (park) park_tile();                           // ... which closed a tile
        __builtin_amdgcn_sched_barrier(0);
        read_frag(S, 1, fy);
        __builtin_amdgcn_sched_barrier(0);
        mma(fx);
        __builtin_amdgcn_sched_barrier(0);
      }
      next_slot();
    };
    auto compute_late = [&](bool first_of_tile, bool park) {
      const uint32_t S = opaque(S_c);
      if (dbg & 4) {
        if (park) park_tile();
      } else {
        if (!first_of_tile || park) mma(fx);             // the previous step's k-half 0
        __builtin_amdgcn_sched_barrier(0);
        read_frag(S, 0, fx);
        __builtin_amdgcn_sched_barrier(0);
        if (!first_of_tile || park) mma(fy);             // the previous step's k-half 1
        if (park) park_tile();                           // ... which closed a tile
        __builtin_amdgcn_sched_barrier(0);
        read_frag(S, 1, fy);
        __builtin_amdgcn_sched_barrier(0);
      }
      next_slot();
    };
    // my DMA of a step has landed when at most the next step's instructions are outstanding; after the barrier everybody's
    // has, and everybody has finished reading the slot of the step before, which the step two ahead overwrites
    auto sync = [&]() {
      wait_vmcnt<PER>();
      ring_barrier();
    };
    auto tile_loop = [&](auto&& compute) {
      for (int j = 0; j < nmine; ++j) {
        const bool has_next = j + 1 < nmine;
        // ---- K step 0
        sync();
        if (nk > 2) issue(a_voff, 2, S_i);
        if (has_next) {
          tile_offsets(grp + (j + 1) * ngroups, a_voff_nx);
          if (nk == 2) issue(a_voff_nx, 0, S_i);
        }
        compute(true, j > 0);
        // ---- K steps whose look-ahead stays inside this tile
        for (int kt = 1; kt + 2 < nk; ++kt) {
          sync();
          issue(a_voff, kt + 2, S_i);
          compute(false, false);
        }
        // ---- the last two: the look-ahead is the next tile's first two steps
        if (nk > 2) {
          sync();
          if (has_next) issue(a_voff_nx, 0, S_i);
          compute(false, false);
        }
        if (has_next) {
          sync();
          issue(a_voff_nx, 1, S_i);
        } else {
          wait_vmcnt<0>();
          ring_barrier();
        }
        compute(false, false);
#pragma unroll
        for (int t = 0; t < AI; ++t) a_voff[t] = a_voff_nx[t];
      }
    };
    tile_offsets(grp, a_voff);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the scale words: nothing but the ring's DMA is counted below)
    issue(a_voff, 0, lds0);
    issue(a_voff, 1, lds0 + kPsStage);
    if (wn == 0) {
      tile_loop(compute_early);
      ring_barrier();                  // the last step's k-half 1 is still to do
      if (!(dbg & 4)) mma(fy);
    } else {
      tile_loop(compute_late);
      ring_barrier();                  // the whole last step is still to do
      if (!(dbg & 4)) {
        mma(fx);
        mma(fy);
      }
    }
    park_tile();
    ring_barrier();                    // the last tile is staged
    if (p.out_amax) {
      // |acc + bias| <= |acc| + max|bias|; ReLU only lowers it: an upper bound is all the consumer's operand scale needs
      amax_m = __builtin_bit_cast(uint32_t, __builtin_bit_cast(float, amax_m) + bias_max);
      amax_commit(p.out_amax, amax_m);
    }
    return;
  }

  // ================================================================== store waves
  // (no load of global memory inside their loop: gfx950 counts loads and stores in ONE in-order vmcnt, and a wait for a load
  // would wait for every store issued before it — a first version re-read the bias per tile and ran at a store per ~900
  // cycles.  The column tile is fixed per workgroup, so the bias quad is read once, before the first store.)
  const int sw = wave - kPsCW;
  const int rh = sw >> 1, chh = sw & 1;            // 64-row half, 64-column half of the tile
  // lane -> (row of the instruction's four, 16-byte chunk of the 256-byte half row): the 16 lanes of a ds_read_b128 service
  // group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32) take ONE half row each
  const int l5 = lane & 31;
  const bool ga = l5 < 4 || (l5 >= 12 && l5 < 16) || (l5 >= 20 && l5 < 28);
  const int rank = ga ? (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12)) : (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16));
  const int rsub = 2 * (lane >> 5) + (ga ? 0 : 1);
  // the lane of the other service group (same half of the wave) that holds the same chunk
  const int twin5 = ga ? (rank < 8 ? rank + 4 : (rank < 12 ? rank + 8 : rank + 16)) : (rank < 4 ? rank : (rank < 8 ? rank + 8 : rank + 12));
  const int twin = (lane & 32) | twin5;
  const int chunk = chh * 16 + rank;                // chunk of 4 floats within the 128-column tile
  const int col = n0 + chunk * 4;
  const bool col_ok = col < p.Cd;

  BnLaneStat stt;
  bn_stat_init(stt);
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr && col_ok) bias = *reinterpret_cast<const f32x4*>(p.bias + col);
  const float floor_v = p.relu ? 0.f : -__builtin_inff();   // ReLU as an unconditional max
  // row r = rh * 64 + 4 i + rsub of the staging image: (r & 7) alternates between two values with i
  const uint32_t ro_even = stg0 + ps_out_off(rh * 64 + rsub, chunk), ro_odd = stg0 + ps_out_off(rh * 64 + 4 + rsub, chunk);
  const size_t pitch = (size_t)4 * p.Cd;            // floats between two instructions' rows
  // the drain cursor: instruction i (0..16) of the tile being drained, its destination, its first row
  int d_i = kPsInstr, d_row = 0, d_tile_m = 0;
  float* d_ptr = nullptr;

  // The store waves sit on the critical path of every K step's barrier, so this loop is kept to ~15 instructions per store
  // (~35 with statistics): no flag tests, running pointers, one LDS read in flight ahead of the store it feeds.
  auto lds_of = [&](int i) { return ((i & 1) ? ro_odd : ro_even) + (uint32_t)(i >> 1) * (8 * kPsOutRow); };
  auto drain = [&](int count) {                      // the next `count` instructions of this wave's tile half
    f32x4 v = __builtin_bit_cast(f32x4, ps_lds_read16(lds_of(d_i)));
    for (int u = 0; u < count; ++u) {
      const int nx = d_i + 1 < kPsInstr ? d_i + 1 : d_i;
      const f32x4 vn = __builtin_bit_cast(f32x4, ps_lds_read16(lds_of(nx)));
      if (d_row < p.M && col_ok) {
        f32x4 t = v + bias;
        t.x = fmaxf(t.x, floor_v); t.y = fmaxf(t.y, floor_v); t.z = fmaxf(t.z, floor_v); t.w = fmaxf(t.w, floor_v);
        if (!(dbg & 8)) *reinterpret_cast<f32x4*>(d_ptr) = t;
        if (STATS) {
          if (stt.n == 0.f) stt.piv = t;
          const f32x4 d = t - stt.piv;
          stt.s += d;
          stt.q += d * d;
          stt.n += 1.f;
        }
      }
      v = vn;
      ++d_i;
      d_row += 4;
      d_ptr += pitch;
    }
  };
  auto set_drain_tile = [&](int item) {
    d_tile_m = item;
    d_i = 0;
    d_row = item * BM + rh * 64 + rsub;
    d_ptr = p.dst + (size_t)d_row * p.Cd + col;
  };
  auto close_tile = [&]() {            // one (count, mean, M2) record per 64-row half tile; this wave owns 64 of its columns
    if (!STATS) return;
    float n = stt.n;
    const float inv = n > 0.f ? 1.f / n : 0.f;
    f32x4 mean = stt.piv + stt.s * inv;
    f32x4 m2 = stt.q - stt.s * stt.s * inv;
    auto merge_from = [&](int src_lane) {
      const float n2 = __shfl(n, src_lane, 64);
      f32x4 mean2, m22;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mean2[e] = __shfl(mean[e], src_lane, 64);
        m22[e] = __shfl(m2[e], src_lane, 64);
      }
      const float nt = n + n2;
      const float w2 = nt > 0.f ? n2 / nt : 0.f;
      const f32x4 dlt = mean2 - mean;
      mean += dlt * w2;
      m2 += m22 + dlt * dlt * (n * w2);
      n = nt;
    };
    merge_from(twin);                  // the other row of this half of the wave
    merge_from(lane ^ 32);             // the other two rows
    if (lane < 32 && ga && col_ok) {
      float* rec = p.bn_part + (size_t)(2 * d_tile_m + rh) * 3 * p.Cd + col;
      *reinterpret_cast<f32x4*>(rec) = f32x4{n, n, n, n};
      *reinterpret_cast<f32x4*>(rec + p.Cd) = mean;
      *reinterpret_cast<f32x4*>(rec + 2 * p.Cd) = m2;
    }
    bn_stat_init(stt);
  };

  // slices of the previous tile over K steps 1 .. nk - 1 of the current one
  const int per = (kPsInstr + nk - 2) / (nk - 1);
  for (int kt = 0; kt < nk; ++kt) ring_barrier();   // first tile: nothing to drain yet
  for (int j = 1; j < nmine; ++j) {
    ring_barrier();
    set_drain_tile(grp + (j - 1) * ngroups);
    for (int kt = 1; kt < nk; ++kt) {
      ring_barrier();
      const int count = min(per, kPsInstr - d_i);
      if (count > 0 && !(dbg & 32)) drain(count);
    }
    close_tile();
  }
  ring_barrier();
  ring_barrier();                      // the last tile is staged
  set_drain_tile(grp + (nmine - 1) * ngroups);
  drain(kPsInstr);
  close_tile();
}

bool conv1x1_ps_applicable(const IGemmArgs& a) {
  if (!conv1x1_dma_applicable(a)) return false;
  const int nk = a.Kpad / BK3;
  // (accumulate epilogues and strided destinations stay on conv1x1_dma.hip: the store waves may not read global memory)
  const int tn = ceil_div(a.Cd, kPsBN);
  return nk >= 2 && (a.Cd & 3) == 0 && a.Cd >= 64 && a.dense_dst && !a.accum && tn <= 32;
}

int launch_conv1x1_ps(IGemmArgs& a, hipStream_t stream) {
  if (!conv1x1_ps_applicable(a)) {
    set_error("conv1x1_ps: launch not supported (1x1, Cs %% 32 == 0, Cs >= 64, Cout %% 4 == 0, dense destination, no accumulate)");
    return EVK_E_UNSUPPORTED;
  }
  static const bool tune = getenv("EVK_TUNE") != nullptr;
  const int dbg = tune && getenv("EVK_C1_DMA_DBG") ? atoi(getenv("EVK_C1_DMA_DBG")) : 0;
  a.tiles_m = ceil_div(a.M, kPsBM);
  a.tiles_n = ceil_div(a.Cd, kPsBN);
  // statistics: two records per row tile (one per 64-row half), within evk_conv2d_stats_max_parts' M / 64 + 1
  a.bn_part = nullptr;
  a.bn_parts = 0;
  if (a.bn_want && a.bn_buf && a.dense_dst && !a.accum && 2LL * a.tiles_m <= a.bn_cap) {
    a.bn_part = a.bn_buf;
    a.bn_parts = 2 * a.tiles_m;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error("conv1x1_ps: cannot query the device");
    return EVK_E_LAUNCH;
  }
  static int cus_of[64];                 // per device id (one process may drive devices of different sizes)
  if (dev < 0 || dev >= 64) dev = 0;
  if (!cus_of[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      set_error("conv1x1_ps: cannot query the device");
      return EVK_E_LAUNCH;
    }
    cus_of[dev] = prop.multiProcessorCount;
  }
  const int cus = cus_of[dev];
  const long long items = (long long)a.tiles_m * a.tiles_n;
  if (items <= 0 || items > 0x7fffffffLL) {
    set_error("conv1x1_ps: bad grid %lld", items);
    return EVK_E_INVALID;
  }
  // gridDim / 8 workgroups per XCD, a multiple of tiles_n (the kernel's tile order), at most one workgroup per CU
  const int slots = (cus / 8) / a.tiles_n * a.tiles_n;
  if (slots <= 0) {
    set_error("conv1x1_ps: %d column tiles do not fit %d CUs per XCD", a.tiles_n, cus / 8);
    return EVK_E_UNSUPPORTED;
  }
  const int grid = 8 * slots;
  const unsigned long long sb = (unsigned long long)a.N * a.Hs * a.Ws * a.Cs * 4ull;
  const unsigned long long wb = 2ull * a.Cd * a.Kpad * 2ull;
  const dim3 g((unsigned)grid), b(64 * kPsWaves);
  const int which = (a.a_packed ? 4 : 0) | (a.bn_part != nullptr ? 2 : 0) | (dbg ? 1 : 0);
  auto go = [&](auto kern) {
    // (once per instantiation would do; the call is a table lookup in the runtime)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kPsLds);
    hipLaunchKernelGGL(kern, g, b, kPsLds, stream, a, (uint32_t)sb, (uint32_t)wb, dbg);
  };
  switch (which) {
    case 0: go(&conv1x1_ps_kernel<false, false, false>); break;
    case 1: go(&conv1x1_ps_kernel<false, false, true>); break;
    case 2: go(&conv1x1_ps_kernel<false, true, false>); break;
    case 3: go(&conv1x1_ps_kernel<false, true, true>); break;
    case 4: go(&conv1x1_ps_kernel<true, false, false>); break;
    case 5: go(&conv1x1_ps_kernel<true, false, true>); break;
    case 6: go(&conv1x1_ps_kernel<true, true, false>); break;
    default: go(&conv1x1_ps_kernel<true, true, true>); break;
  }
  return check_launch("conv1x1_ps");
}

}  // namespace evk
