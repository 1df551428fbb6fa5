// One-tap (1x1, any stride) convolutions of the f16x2 arithmetic, forward and data gradient, as a PERSISTENT kernel with a
// STORE ROLE:   dst[m][co] = sum_k src[row(m)][k] * w[co][k]      (arithmetic: conv_igemm_x3.hip / x3_common.hpp)
//
// conv1x1_dma.hip moves both operands global -> LDS by DMA, but its workgroup is "K loop, then a burst of 64 KB of stores":
// ablated on 256 -> 256 @128^2 (tools/ab_c1dma.py, us): loads alone 75, stores alone 57, loads + stores WITHOUT any compute
// 159, everything 192 — the tile's stores and the next tile's loads share one in-order vector-memory path per CU and one
// in-order vmcnt per wave, and two co-resident workgroups do not find the opposite phase by themselves.  Here
//  * one workgroup per CU walks a contiguous run of ROW tiles of ONE column tile; the workgroups that take the other column
//    tiles of the same rows sit on the same XCD (workgroup id mod 8) and walk in step, so the activation rows are fetched
//    from HBM once and found in that XCD's L2 by the others (a single workgroup doing the column tiles one after the other
//    found them evicted: loads alone 113 us against 76); the DMA ring (three stages of 32 KB, two K steps in flight) runs
//    ACROSS tile boundaries, so a tile has no prologue;
//  * eight compute waves (4 x 2, two per SIMD) issue the DMA, read fragments and run the MFMAs exactly as in conv1x1_dma.hip,
//    with the fragment reads of one k-half issued under the MFMAs of the other (two register sets, no extra barrier);
//    a finished tile goes accumulator -> LDS (64 KB staging image, XOR-swizzled rows) and the waves go straight on;
//  * four STORE waves, which never wait for a load of the ring, drain the staging image during the NEXT tile's K steps, a
//    slice per step: row-contiguous 16-byte reads, bias / accumulate (+ ReLU bits) / ReLU / operand-scale maximum /
//    BatchNorm statistics on the way, 256 contiguous bytes per row and four rows per store instruction.  Loads and stores
//    reach the memory path interleaved at K-step granularity instead of in 64 KB bursts.
// One s_barrier per K step, shared by all twelve waves (gfx950 has no named barriers): the store waves run the same step
// sequence and do their slice between two of them.
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
__global__ __launch_bounds__(64 * (kPsCW + kPsSW)) void conv1x1_ps_kernel(const IGemmArgs p, uint32_t src_bytes,
                                                                          uint32_t wgt_bytes, int dbg_arg) {
  const int dbg = DBG ? dbg_arg : 0;   // (ablation switches compiled out of the production instantiations)
  constexpr int BM = kPsBM, BN = kPsBN, NST = kPsNST, WM = 32, WN = 64, NB = 2;
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
  // row tiles grp, grp + ngroups, ...: the tiles in flight at any moment are NEIGHBOURS in memory.  (Contiguous runs per
  // workgroup put all 256 streams at the same offset of 2 MB-aligned regions — the same HBM channels at the same time:
  // loads alone 106 us, stores alone 103 us, against 75 / 58 for the tile order of conv1x1_dma.hip.)
  const int ngroups = 8 * gpx;
  const int nmine = grp < p.tiles_m ? (p.tiles_m - grp + ngroups - 1) / ngroups : 0;
  if (nmine <= 0) return;                              // (the whole workgroup: no barrier is left waiting)
  const int n0 = tile_n * BN;
  const int nk = p.Kpad / BK3;
  const int G = nmine * nk;                            // K steps of this workgroup
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_ps;
  const uint32_t stg0 = lds0 + kPsRing;

  if (wave < kPsCW) {
    // ================================================================ compute waves
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave & 3, wn = wave >> 2;
    const i32x4 rs_a = make_rsrc(p.src, src_bytes), rs_b = make_rsrc(p.wgt3, wgt_bytes);
    const uint32_t plane_bytes = (uint32_t)p.Cd * (uint32_t)p.Kpad * 2u;

    // issue cursor: the tile and K step of the next DMA
    int is_item = grp, is_kt = 0, is_g = 0;
    uint32_t a_voff[AI], b_voff[BI];
#pragma unroll
    for (int t = 0; t < BI; ++t) {     // weights: the same column tile for every row tile of this workgroup
      const int s = 64 * (BI * wave + t) + lane;
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
        const int row = 8 * (AI * wave + t) + (lane >> 3);
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
    auto issue_next = [&]() {          // DMA of global step is_g into ring slot is_g % NST
      if (is_kt == 0) tile_offsets(is_item);
      const uint32_t S = lds0 + (uint32_t)(is_g % NST) * kPsStage;
      const uint32_t ka = (uint32_t)is_kt * kPsRow, kb = (uint32_t)is_kt * kRowBytes;
      if (!(dbg & 1)) {
#pragma unroll
        for (int t = 0; t < AI; ++t) ps_dma16s(rs_a, S + (AI * wave + t) * 1024, a_voff[t], ka);
      }
      if (!(dbg & 2)) {
#pragma unroll
        for (int t = 0; t < BI; ++t) ps_dma16s(rs_b, S + kPsAStage + (BI * wave + t) * 1024, b_voff[t], kb);
      }
      ++is_g;
      if (++is_kt == nk) { is_kt = 0; is_item += ngroups; }
    };

    // fragment read offsets inside a stage (lane constants)
    uint32_t fa_off[2][2], fb_off[2];
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
    // staging offsets of this lane's 8 accumulator quads
    uint32_t st_off[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) st_off[b][r4] = ps_out_off(wm * WM + li, wn * 16 + b * 8 + 2 * r4 + lh);

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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the scale words: nothing but the ring's DMA is counted below

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
    // the output's operand-scale maximum is taken here, from the accumulators (the store waves have no slot to spare); a
    // bias is added by the store waves afterwards, so it enters as max|bias| over the tile's columns below
    uint32_t amax_m = 0;
    auto park_tile = [&]() {           // accumulators -> staging image, then start the next tile from zero
      if (!(dbg & 16)) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 v = {acc[b][4 * r4] * out_scale, acc[b][4 * r4 + 1] * out_scale, acc[b][4 * r4 + 2] * out_scale,
                             acc[b][4 * r4 + 3] * out_scale};
            ps_lds_write16(stg0 + st_off[b][r4], v);
            amax_m = max(amax_m, max(max(__builtin_bit_cast(uint32_t, v.x) & 0x7fffffffu, __builtin_bit_cast(uint32_t, v.y) & 0x7fffffffu),
                                     max(__builtin_bit_cast(uint32_t, v.z) & 0x7fffffffu, __builtin_bit_cast(uint32_t, v.w) & 0x7fffffffu)));
          }
      }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    };

    issue_next();
    if (G > 1) issue_next();
    Frag fx, fy;
    int kt = 0;                        // K step (within its tile) of global step g
    for (int g = 0; g < G; ++g) {
      // my DMA of step g has landed when at most step g + 1's instructions are outstanding; after the barrier everybody's
      // has, and everybody has finished reading the slot of step g - 1, which step g + 2 overwrites
      if (g + 1 < G) {
        wait_vmcnt<PER>();
      } else {
        wait_vmcnt<0>();
      }
      ring_barrier();
      if (is_g < G) issue_next();
      const uint32_t S = opaque(lds0 + (uint32_t)(g % NST) * kPsStage);
      if (dbg & 4) {
        if (g > 0 && kt == 0) park_tile();
      } else {
        read_frag(S, 0, fx);
        if (g > 0) {
          mma(fy);                     // second k-half of step g - 1, under the reads just issued
          if (kt == 0) park_tile();    // ... which closed a tile
        }
        read_frag(S, 1, fy);
        mma(fx);
      }
      if (++kt == nk) kt = 0;
    }
    ring_barrier();                    // barrier G
    if (!(dbg & 4)) mma(fy);
    park_tile();
    ring_barrier();                    // barrier G + 1: the last tile is staged
    if (p.out_amax) {
      // |acc + bias| <= |acc| + max|bias|; ReLU only lowers it: an upper bound is all the consumer's operand scale needs
      amax_m = __builtin_bit_cast(uint32_t, __builtin_bit_cast(float, amax_m) + bias_max);
      amax_commit(p.out_amax, amax_m);
    }
    return;
  }

  // ================================================================== store waves
  // (no load of global memory inside their loop: gfx950 counts loads and stores in ONE in-order vmcnt, and a wait for a load
  // would wait for every store issued before it — the first version re-read the bias per tile and ran at a store per ~900
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
  auto drain = [&](int count) {                      // the next `count` instructions of this wave's 16
    f32x4 v = __builtin_bit_cast(f32x4, ps_lds_read16(((d_i & 1) ? ro_odd : ro_even) + (uint32_t)(d_i >> 1) * (8 * kPsOutRow)));
    for (int u = 0; u < count; ++u) {
      const int nx = d_i + 1 < kPsInstr ? d_i + 1 : d_i;
      const f32x4 vn = __builtin_bit_cast(f32x4, ps_lds_read16(((nx & 1) ? ro_odd : ro_even) + (uint32_t)(nx >> 1) * (8 * kPsOutRow)));
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
  const int per = nk > 1 ? (kPsInstr + nk - 2) / (nk - 1) : kPsInstr;
  int kt = 0, item = grp;
  for (int g = 0; g < G; ++g) {
    ring_barrier();
    if (item > grp && kt >= 1) {
      if (kt == 1) set_drain_tile(item - ngroups);
      const int count = min(per, kPsInstr - d_i);
      if (count > 0 && !(dbg & 32)) drain(count);
      if (kt == nk - 1) close_tile();
    }
    if (++kt == nk) { kt = 0; item += ngroups; }
  }
  ring_barrier();                      // barrier G
  ring_barrier();                      // barrier G + 1: the last tile is staged
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
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      set_error("conv1x1_ps: cannot query the device");
      return EVK_E_LAUNCH;
    }
    cus = prop.multiProcessorCount;
  }
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
  const dim3 g((unsigned)grid), b(64 * (kPsCW + kPsSW));
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
