// One-tap (1x1, any stride) convolutions of the f16x2 arithmetic, forward and data gradient: PERSISTENT workgroups with three
// wave roles —  dst[m][co] = sum_k src[row(m)][k] * w[co][k]      (arithmetic: conv_igemm_x3.hip / x3_common.hpp)
//
// Round 5: conv1x1_ps.hip (a workgroup walks row tiles of one column tile, the DMA ring runs across tile boundaries, store
// waves drain a finished tile from an LDS staging image under the next tile's K steps) with the two lessons of
// conv1x1_sp.hip / tools/probes/dma_issue.hip built in:
//  * LOADER waves.  A CU takes ~32-40 outstanding 1 KB vector-memory instructions; the wave that issues one more blocks
//    until one returns, and an in-order wave that is blocked cannot issue its MFMAs.  The eight compute waves issue no
//    vector-memory instruction at all; four loader waves issue the ring (their streaming rate scales with the number of
//    issuing waves: 2 -> 3.0 TB/s, 4 -> 5.8, dma_patterns.hip) and absorb the back-pressure; four store waves, which never
//    load, write the output (gfx950 counts loads and stores in one in-order vmcnt per wave: the roles must not mix).
//  * A K step with no exposed LDS latency: the fragment registers are double-buffered across the barrier,
//        read k-half 1 | MFMAs k-half 0 | barrier (stage g+1 has landed) | read k-half 0 of stage g+1 | MFMAs k-half 1
//    (conv1x1_sp.hip has the ISA argument), and nothing in the loop tests a run-time switch.
// Sixteen waves, one workgroup per CU, <= 128 registers per lane; LDS = ring of three 32 KB stages + 64 KB staging image.
// One s_barrier per K step shared by all roles (gfx950 has no named barriers).  Barrier sequence of a workgroup with T tiles
// of nk steps (G = T nk global steps): P0, B(0) .. B(G-1), E1.
//   compute: [read Y(g,1); MFMA X; B(g); read X(g+1,0); MFMA Y; after a tile's last step: park accumulators in the staging
//            image] — the parked image is complete at the next barrier (its ds_writes are waited for there);
//   loader : after B(g-1): issue stage g+2 into the slot stage g-1 has left; wait for stage g+1; B(g);
//   store  : tile j's image is complete after B(first step of tile j+1) and is overwritten after B(last step of tile j+1):
//            nk - 1 slices between those barriers; the last tile after E1.
// Compile-time ablations (tools/build_variant.sh -DEVK_PS2_ABL=bits): 1 no loader DMA, 4 no compute, 8 no global stores,
// 16 store waves idle, 32 no staging writes.
#include "igemm_common.hpp"
#include "x3_common.hpp"
#include "lds_dma.hpp"
#include <stdlib.h>
#include <string.h>

#ifndef EVK_PS2_ABL
#define EVK_PS2_ABL 0
#endif

namespace evk {

namespace {

constexpr int kP2Row = BK3 * 4;      // bytes of one activation row of a K step (32 four-byte words)
constexpr int kP2BM = 128, kP2BN = 128, kP2NST = 3;
constexpr int kP2AStage = kP2BM * kP2Row, kP2BPlane = kP2BN * kRowBytes, kP2Stage = kP2AStage + 2 * kP2BPlane;
constexpr int kP2Ring = kP2NST * kP2Stage;
constexpr int kP2OutRow = kP2BN * 4;                 // bytes of one staged output row
constexpr int kP2Staging = kP2BM * kP2OutRow;        // 64 KB
constexpr int kP2Lds = kP2Ring + kP2Staging;         // 160 KB: the whole LDS of a CU
constexpr int kP2CW = 8, kP2LW = 4, kP2SW = 4;       // compute / loader / store waves
constexpr int kP2Waves = kP2CW + kP2LW + kP2SW;
constexpr int kP2Instr = 16;                         // store instructions (4 rows x 256 B) per store wave and tile

__device__ __forceinline__ void p2_dma16s(i32x4 rsrc, uint32_t lds_byte, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
               :: "v"(voff), "s"(lds_byte), "s"(rsrc), "s"(soff) : "memory", "m0");
}
// the activation stream with the non-temporal hint: a row tile is read by tiles_n workgroups of one XCD — with one or two
// column tiles the hint is ahead (same box, us: 256 -> 256 @128^2 197 -> 187, 64 -> 256 92 -> 87, 256 -> 128 94 -> 90), with
// four it evicts what the neighbours are about to read (128 -> 512 @64^2 51 -> 58): the launcher decides (template NT)
template <bool NT>
__device__ __forceinline__ void p2_dma16s_a(i32x4 rsrc, uint32_t lds_byte, uint32_t voff, uint32_t soff) {
  if constexpr (NT) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen nt lds"
                 :: "v"(voff), "s"(lds_byte), "s"(rsrc), "s"(soff) : "memory", "m0");
  } else {
    p2_dma16s(rsrc, lds_byte, voff, soff);
  }
}
// (ablation bit 64: no barrier at all — the loop's own speed; results are garbage)
__device__ __forceinline__ void p2_barrier() {
  if (EVK_PS2_ABL & 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else ring_barrier();
}
__device__ __forceinline__ u32x4 p2_lds_read16(uint32_t lds_byte) {
  return *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)lds_byte;
}
__device__ __forceinline__ void p2_lds_write16(uint32_t lds_byte, f32x4 v) {
  *(__attribute__((address_space(3))) f32x4*)(uintptr_t)lds_byte = v;
}
__device__ __forceinline__ int p2_arow_off(int row, int c) { return row * kP2Row + ((c ^ ((row >> 1) & 7)) << 4); }
// staged output: 16-byte chunk c (0..31) of row `row` (conv1x1_ps.hip: ps_out_off)
__device__ __forceinline__ uint32_t p2_out_off(int row, int c) { return (uint32_t)(row * kP2OutRow + ((c ^ (row & 7)) << 4)); }

}  // namespace

template <bool PK, bool STATS, bool NT>
__global__ __launch_bounds__(64 * kP2Waves) void conv1x1_ps2_kernel(const IGemmArgs p, uint32_t src_bytes, uint32_t wgt_bytes) {
  constexpr int BM = kP2BM, BN = kP2BN, WM = 32, WN = 64, NB = 2, NST = kP2NST;
  constexpr int AI = kP2AStage / 1024 / kP2LW;        // activation DMA instructions per loader wave and stage (8 rows each)
  constexpr int BI = 2 * kP2BPlane / 1024 / kP2LW;    // weight-plane DMA instructions per loader wave and stage
  constexpr int PER = AI + BI;
  static_assert(AI == 4 && BI == 4 && NST == 3, "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p2[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // tile order (conv1x1_ps.hip): XCD x = id & 7 holds gridDim / 8 workgroups; tiles_n of them form a group that walks the
  // same row tiles grp, grp + ngroups, ...
  const int slots = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int gpx = slots / p.tiles_n;
  const int tile_n = slot % p.tiles_n, grp = xcd * gpx + slot / p.tiles_n;
  const int ngroups = 8 * gpx;
  const int nmine = grp < p.tiles_m ? (p.tiles_m - grp + ngroups - 1) / ngroups : 0;
  if (nmine <= 0) return;                              // (the whole workgroup: no barrier is left waiting)
  const int n0 = tile_n * BN;
  const int nk = p.Kpad / BK3;                         // >= 2 (conv1x1_ps2_applicable)
  const int G = nmine * nk;                            // global K steps of this workgroup
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_p2;
  const uint32_t stg0 = lds0 + kP2Ring;

  if (wave >= kP2CW && wave < kP2CW + kP2LW) {
    // ================================================================== loader waves
    const int lw = wave - kP2CW;
    const i32x4 rs_a = make_rsrc(p.src, src_bytes), rs_b = make_rsrc(p.wgt3, wgt_bytes);
    const uint32_t plane_bytes = (uint32_t)p.Cd * (uint32_t)p.Kpad * 2u;
    uint32_t a_voff[AI], b_voff[BI];
#pragma unroll
    for (int t = 0; t < BI; ++t) {     // weights: the same column tile for every row tile of this workgroup
      const int s = 64 * (BI * lw + t) + lane;
      const int pt = s / (BN * 4);
      const int row = (s - pt * BN * 4) >> 2;
      const int c = (s & 3) ^ ((row >> 2) & 3);
      const int co = n0 + row;
      b_voff[t] = co < p.Cd ? (uint32_t)pt * plane_bytes + (uint32_t)co * (uint32_t)p.Kpad * 2u + (uint32_t)c * 16u : kDmaOOB;
    }
    auto tile_offsets = [&](int item) {
      const int m0 = item * BM;
#pragma unroll
      for (int t = 0; t < AI; ++t) {
        const int row = 8 * (AI * lw + t) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
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
    };
    // the issue cursor: (tile i_j, step i_kt) = global stage i_g, into slot S_i
    int i_j = 0, i_kt = 0;
    uint32_t S_i = lds0;
    tile_offsets(grp);
    auto issue_next = [&]() {
      if (!(EVK_PS2_ABL & 1)) {
        const uint32_t ka = (uint32_t)i_kt * kP2Row, kb = (uint32_t)i_kt * kRowBytes;
#pragma unroll
        for (int t = 0; t < AI; ++t) p2_dma16s_a<NT>(rs_a, S_i + (AI * lw + t) * 1024, a_voff[t], ka);
#pragma unroll
        for (int t = 0; t < BI; ++t) p2_dma16s(rs_b, S_i + kP2AStage + (BI * lw + t) * 1024, b_voff[t], kb);
      }
      S_i = S_i == lds0 + (NST - 1) * kP2Stage ? lds0 : S_i + kP2Stage;
      if (++i_kt == nk) {
        i_kt = 0;
        ++i_j;
        if (i_j < nmine) tile_offsets(grp + i_j * ngroups);
      }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (nothing but the ring's DMA is counted below)
    issue_next();                                      // stage 0
    if (G > 1) issue_next();                           // stage 1
    if (G > 1) wait_vmcnt<PER>(); else wait_vmcnt<0>();
    p2_barrier();                                    // P0: stage 0 has landed
    for (int g = 0; g < G; ++g) {
      if (g + 2 < G) issue_next();                     // stage g + 2 into the slot stage g - 1 has left
      if (g + 2 < G) wait_vmcnt<PER>(); else wait_vmcnt<0>();   // stage g + 1 has landed
      p2_barrier();                                  // B(g)
    }
    return;                                            // (E1: an ended wave is not waited for)
  }

  if (wave < kP2CW) {
    // ================================================================ compute waves
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave & 3, wn = wave >> 2;
    uint32_t fa_off[2][2], fb_off[2];   // fragment read offsets inside a stage (lane constants)
    {
      const int row = wm * WM + li;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h) fa_off[kk][h] = (uint32_t)p2_arow_off(row, 4 * kk + 2 * lh + h);
      const int brow = wn * WN + li;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) fb_off[kk] = (uint32_t)(kP2AStage + plane_off(brow, 2 * kk + lh));
    }
    // staging offsets of this lane's accumulator quads: chunk = wn * 16 + b * 8 + (2 r4 + lh); block b is a 128-byte immediate
    uint32_t st_off[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) st_off[r4] = stg0 + p2_out_off(wm * WM + li, wn * 16 + 2 * r4 + lh);

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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

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
      if (EVK_PS2_ABL & 128) {       // (ablation: no fragment reads — the registers keep whatever they hold, made opaque)
        asm volatile("" : "+v"(f.a0), "+v"(f.a1));
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) asm volatile("" : "+v"(f.b[b][pt]));
        return;
      }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
          f.b[b][pt] = __builtin_bit_cast(bf16x8, p2_lds_read16(S + fb_off[kk] + pt * kP2BPlane + b * 32 * kRowBytes));
      f.a0 = p2_lds_read16(S + fa_off[kk][0]);
      f.a1 = p2_lds_read16(S + fa_off[kk][1]);
    };
    auto mma = [&](const Frag& f) {
      // (read as floats: a bit_cast of an ext-vector ELEMENT is miscompiled by this hipcc; conv1x1_dma.hip)
      const f32x4 w0 = __builtin_bit_cast(f32x4, f.a0), w1 = __builtin_bit_cast(f32x4, f.a1);
      u32x4 H, L;
      uint32_t h, l, unused = 0;
      if (EVK_PS2_ABL & 256) {       // (ablation: no operand split — the raw words go to the matrix pipe)
        H = f.a0; L = f.a1;
      } else {
      split_op<2, PK>(w0.x, w0.y, a_inv, h, l, unused); H[0] = h; L[0] = l;
      split_op<2, PK>(w0.z, w0.w, a_inv, h, l, unused); H[1] = h; L[1] = l;
      split_op<2, PK>(w1.x, w1.y, a_inv, h, l, unused); H[2] = h; L[2] = l;
      split_op<2, PK>(w1.z, w1.w, a_inv, h, l, unused); H[3] = h; L[3] = l;
      }
      const bf16x8 fa[2] = {__builtin_bit_cast(bf16x8, H), __builtin_bit_cast(bf16x8, L)};
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = mfma_np<2>(f.b[b][kHB[t]], fa[kHA[t]], acc[b]);
    };
    uint32_t amax_m = 0;
    auto park_tile = [&]() {           // accumulators -> staging image, then start the next tile from zero
      if (EVK_PS2_ABL & 32) {          // (ablation: keep the accumulators live without the LDS writes)
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[b][r];
        amax_m = max(amax_m, __builtin_bit_cast(uint32_t, sum));
      } else
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 v = {acc[b][4 * r4] * out_scale, acc[b][4 * r4 + 1] * out_scale, acc[b][4 * r4 + 2] * out_scale,
                           acc[b][4 * r4 + 3] * out_scale};
          p2_lds_write16(st_off[r4] + b * 128, v);
          amax_m = max(amax_m, max(max(__builtin_bit_cast(uint32_t, v.x) & 0x7fffffffu, __builtin_bit_cast(uint32_t, v.y) & 0x7fffffffu),
                                   max(__builtin_bit_cast(uint32_t, v.z) & 0x7fffffffu, __builtin_bit_cast(uint32_t, v.w) & 0x7fffffffu)));
        }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    };

    Frag fx, fy;
    p2_barrier();                                    // P0: stage 0 has landed
    uint32_t S_c = lds0;
    int kt = 0;
    if (!(EVK_PS2_ABL & 4)) read_frag(opaque(S_c), 0, fx);
    // (the last global step is peeled: a conditional read behind the barrier would make hipcc wait for the reads just issued)
    for (int g = 0; g + 1 < G; ++g) {
      const uint32_t S = opaque(S_c);
      S_c = S_c == lds0 + (NST - 1) * kP2Stage ? lds0 : S_c + kP2Stage;
      const uint32_t Sn = opaque(S_c);
      const bool last_of_tile = ++kt == nk;
      if (last_of_tile) kt = 0;
      if (EVK_PS2_ABL & 4) {
        p2_barrier();
        if (last_of_tile) park_tile();
        continue;
      }
      read_frag(S, 1, fy);
      __builtin_amdgcn_sched_barrier(0);
      mma(fx);
      __builtin_amdgcn_sched_barrier(0);
      p2_barrier();                                  // B(g): stage g + 1 has landed; fy returned long ago
      read_frag(Sn, 0, fx);
      __builtin_amdgcn_sched_barrier(0);
      mma(fy);
      __builtin_amdgcn_sched_barrier(0);
      if (last_of_tile) park_tile();                   // complete at B(g + 1): ring_barrier waits for the ds_writes
    }
    if (EVK_PS2_ABL & 4) {
      p2_barrier();
    } else {
      read_frag(opaque(S_c), 1, fy);
      __builtin_amdgcn_sched_barrier(0);
      mma(fx);
      __builtin_amdgcn_sched_barrier(0);
      p2_barrier();                                  // B(G - 1)
      mma(fy);
    }
    park_tile();
    p2_barrier();                                    // E1: the last tile is staged
    if (p.out_amax) {
      // |acc + bias| <= |acc| + max|bias|; ReLU only lowers it: an upper bound is all the consumer's operand scale needs
      amax_m = __builtin_bit_cast(uint32_t, __builtin_bit_cast(float, amax_m) + bias_max);
      amax_commit(p.out_amax, amax_m);
    }
    return;
  }

  // ================================================================== store waves (conv1x1_ps.hip's, unchanged in substance)
  const int sw = wave - kP2CW - kP2LW;
  const int rh = sw >> 1, chh = sw & 1;            // 64-row half, 64-column half of the tile
  // lane -> (row of the instruction's four, 16-byte chunk of the 256-byte half row): the 16 lanes of a ds_read_b128 service
  // group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32) take ONE half row each
  const int l5 = lane & 31;
  const bool ga = l5 < 4 || (l5 >= 12 && l5 < 16) || (l5 >= 20 && l5 < 28);
  const int rank = ga ? (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12)) : (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16));
  const int rsub = 2 * (lane >> 5) + (ga ? 0 : 1);
  const int twin5 = ga ? (rank < 8 ? rank + 4 : (rank < 12 ? rank + 8 : rank + 16)) : (rank < 4 ? rank : (rank < 8 ? rank + 8 : rank + 12));
  const int twin = (lane & 32) | twin5;
  const int chunk = chh * 16 + rank;                // chunk of 4 floats within the 128-column tile
  const int col = n0 + chunk * 4;
  const bool col_ok = col < p.Cd;

  BnLaneStat stt;
  bn_stat_init(stt);
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr && col_ok) bias = *reinterpret_cast<const f32x4*>(p.bias + col);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the one load of this role, before its first store)
  const float floor_v = p.relu ? 0.f : -__builtin_inff();   // ReLU as an unconditional max
  const uint32_t ro_even = stg0 + p2_out_off(rh * 64 + rsub, chunk), ro_odd = stg0 + p2_out_off(rh * 64 + 4 + rsub, chunk);
  const size_t pitch = (size_t)4 * p.Cd;            // floats between two instructions' rows
  int d_i = kP2Instr, d_row = 0, d_tile_m = 0;
  float* d_ptr = nullptr;

  auto lds_of = [&](int i) { return ((i & 1) ? ro_odd : ro_even) + (uint32_t)(i >> 1) * (8 * kP2OutRow); };
  auto drain = [&](int count) {                      // the next `count` instructions of this wave's tile half
    f32x4 v = __builtin_bit_cast(f32x4, p2_lds_read16(lds_of(d_i)));
    for (int u = 0; u < count; ++u) {
      const int nx = d_i + 1 < kP2Instr ? d_i + 1 : d_i;
      const f32x4 vn = __builtin_bit_cast(f32x4, p2_lds_read16(lds_of(nx)));
      if (d_row < p.M && col_ok) {
        f32x4 t = v + bias;
        t.x = fmaxf(t.x, floor_v); t.y = fmaxf(t.y, floor_v); t.z = fmaxf(t.z, floor_v); t.w = fmaxf(t.w, floor_v);
        if (!(EVK_PS2_ABL & 8)) *reinterpret_cast<f32x4*>(d_ptr) = t;
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

  // slices of tile j-1 in the nk - 1 intervals between B(first step of tile j) and B(last step of tile j)
  const int per = (kP2Instr + nk - 2) / (nk - 1);
  p2_barrier();                                    // P0
  for (int k = 0; k < nk; ++k) p2_barrier();       // tile 0: B(0) .. B(nk - 1), nothing to drain yet
  for (int j = 1; j < nmine; ++j) {
    p2_barrier();                                  // B(first step of tile j): tile j-1 is staged
    set_drain_tile(grp + (j - 1) * ngroups);
    for (int k = 1; k < nk; ++k) {
      const int count = min(per, kP2Instr - d_i);
      if (count > 0 && !(EVK_PS2_ABL & 16)) drain(count);
      p2_barrier();                                // B(j nk + k)   (waits for this wave's LDS reads: lgkmcnt(0))
    }
    close_tile();
  }
  p2_barrier();                                    // E1: the last tile is staged
  set_drain_tile(grp + (nmine - 1) * ngroups);
  if (!(EVK_PS2_ABL & 16)) drain(kP2Instr);
  close_tile();
}

bool conv1x1_ps2_applicable(const IGemmArgs& a) {
  if (!conv1x1_dma_applicable(a)) return false;
  const int nk = a.Kpad / BK3;
  // (accumulate epilogues and strided destinations stay on conv1x1_dma.hip: the store waves may not read global memory)
  const int tn = ceil_div(a.Cd, kP2BN);
  return nk >= 2 && (a.Cd & 3) == 0 && a.Cd >= 64 && a.dense_dst && !a.accum && tn <= 32;
}

int launch_conv1x1_ps2(IGemmArgs& a, hipStream_t stream) {
  if (!conv1x1_ps2_applicable(a)) {
    set_error("conv1x1_ps2: launch not supported (1x1, Cs %% 32 == 0, Cs >= 64, Cout %% 4 == 0, dense destination, no accumulate)");
    return EVK_E_UNSUPPORTED;
  }
  a.tiles_m = ceil_div(a.M, kP2BM);
  a.tiles_n = ceil_div(a.Cd, kP2BN);
  // statistics: two records per row tile (one per 64-row half), within evk_conv2d_stats_max_parts' M / 64 + 1
  a.bn_part = nullptr;
  a.bn_parts = 0;
  if (a.bn_want && a.bn_buf && a.dense_dst && !a.accum && 2LL * a.tiles_m <= a.bn_cap) {
    a.bn_part = a.bn_buf;
    a.bn_parts = 2 * a.tiles_m;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error("conv1x1_ps2: cannot query the device");
    return EVK_E_LAUNCH;
  }
  static int cus_of[64];                 // per device id (one process may drive devices of different sizes)
  if (dev < 0 || dev >= 64) dev = 0;
  if (!cus_of[dev]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      set_error("conv1x1_ps2: cannot query the device");
      return EVK_E_LAUNCH;
    }
    cus_of[dev] = prop.multiProcessorCount;
  }
  const int cus = cus_of[dev];
  const long long items = (long long)a.tiles_m * a.tiles_n;
  if (items <= 0 || items > 0x7fffffffLL) {
    set_error("conv1x1_ps2: bad grid %lld", items);
    return EVK_E_INVALID;
  }
  // gridDim / 8 workgroups per XCD, a multiple of tiles_n (the kernel's tile order), at most one workgroup per CU
  const int slots = (cus / 8) / a.tiles_n * a.tiles_n;
  if (slots <= 0) return 1;   // (more column tiles than this device has CUs per XCD: not this kernel's launch — the caller falls
                              // through to the non-persistent forms)
  const int grid = 8 * slots;
  const unsigned long long sb = (unsigned long long)a.N * a.Hs * a.Ws * a.Cs * 4ull;
  const unsigned long long wb = 2ull * a.Cd * a.Kpad * 2ull;
  const dim3 g((unsigned)grid), b(64 * kP2Waves);
  // non-temporal activation loads where at most two workgroups read a row tile (256 -> 256 @128^2 197 -> 187 us; with four
  // column tiles the hint loses: 128 -> 512 @64^2 51 -> 58)
  const int which = (a.a_packed ? 2 : 0) | (a.bn_part != nullptr ? 1 : 0) | (a.tiles_n <= 2 ? 4 : 0);
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kP2Lds);
    hipLaunchKernelGGL(kern, g, b, kP2Lds, stream, a, (uint32_t)sb, (uint32_t)wb);
  };
  switch (which) {
    case 0: go(&conv1x1_ps2_kernel<false, false, false>); break;
    case 1: go(&conv1x1_ps2_kernel<false, true, false>); break;
    case 2: go(&conv1x1_ps2_kernel<true, false, false>); break;
    case 3: go(&conv1x1_ps2_kernel<true, true, false>); break;
    case 4: go(&conv1x1_ps2_kernel<false, false, true>); break;
    case 5: go(&conv1x1_ps2_kernel<false, true, true>); break;
    case 6: go(&conv1x1_ps2_kernel<true, false, true>); break;
    default: go(&conv1x1_ps2_kernel<true, true, true>); break;
  }
  return check_launch("conv1x1_ps2");
}

}  // namespace evk
