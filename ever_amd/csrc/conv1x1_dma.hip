// One-tap (1x1, any stride) convolutions of the f16x2 arithmetic, forward and data gradient, with BOTH operands moved
// global -> LDS by DMA:   dst[m][co] = sum_k src[row(m)][k] * w[co][k]      (arithmetic: conv_igemm_x3.hip / x3_common.hpp)
//
// These layers are HBM-bound (64..2048 reduction channels, 0.2-0.3 of their byte bound in the register-staged
// kernels, whose one step of look-ahead leaves 16-32 KB per CU in flight against the ~64 KB that 8 TB/s x ~2 us of latency
// asks of each of 256 CUs).  Here
//  * a K step's activation tile lands as RAW 4-byte words — fp32 values, or the producer's packed (h | l << 16) words —
//    in 128-byte rows (whole cache lines), the weights as their two pre-split fp16 planes in 64-byte rows; the DMA's LDS
//    side is lane-linear, so the conflict-free images are made by permuting the per-lane SOURCE chunk and XOR-ing the
//    fragment reads the same way.  Rows past M, columns past Cd and pixels outside the image get an out-of-range offset:
//    the DMA writes zeros;
//  * ring of three stages (48 KB each at 128 x 256), two steps = up to 96 KB per CU in flight, counted vmcnt, raw
//    s_barrier; the DMA is inline asm (lds_dma.hpp: the builtin makes hipcc drain it before every LDS read);
//  * the split of an fp32 activation happens on the fragment a lane has just read (8 floats -> h, l fp16x8: 4 packed
//    converts + 4 subtract pairs + 4 packed converts), once per 32 rows x BN/2 columns, in the shadow of the other wave of
//    the SIMD's MFMAs; a packed operand needs 8 byte permutes;
//  * eight symmetric waves, 4 (rows) x 2 (column halves), two per SIMD: each issues its eighth of the step's DMA, reads
//    its own fragments and owns 32 rows x BN/2 columns of the tile.
// Epilogue: the shared ones of igemm_common.hpp (bias / accumulate (+ ReLU bits) / ReLU / operand-scale slots, or the
// BatchNorm statistics form through LDS, which reuses the ring).
#include "igemm_common.hpp"
#include "x3_common.hpp"
#include "lds_dma.hpp"
#include <stdlib.h>
#include <string.h>

#ifndef EVK_C1_DMA_ABL
#define EVK_C1_DMA_ABL 0   // timing ablations (tools/build_variant.sh -DEVK_C1_DMA_ABL=n; wrong results): 1 no activation DMA, 2 no weight
#endif                     // DMA, 4 no compute, 8 no stores, 16 every tile twice with half of the K loop each
namespace evk {

namespace {

constexpr int kC1Row = BK3 * 4;  // bytes of one activation row of a K step (32 four-byte words)

// 64 lanes x 16 bytes with a scalar byte offset on top of the per-lane one (the K step; not part of the range check, so an
// out-of-range lane offset stays out of range)
__device__ __forceinline__ void dma16s(i32x4 rsrc, uint32_t lds_byte, uint32_t voff, uint32_t soff) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 lds_read16(uint32_t lds_byte) {
  return *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)lds_byte;
}
// byte offset of 16-byte chunk c (0..7) of activation row `row`: rows are 128 B, two per 256-byte bank row; the XOR puts
// the same logical chunk of the 16 rows a ds_read_b128 service group touches on 16 distinct 16-byte slots
__device__ __forceinline__ int c1_arow_off(int row, int c) { return row * kC1Row + ((c ^ ((row >> 1) & 7)) << 4); }

}  // namespace

template <int BN, bool PK, int NST, int WAVES_M>
__global__ __launch_bounds__(128 * WAVES_M) void conv1x1_dma_kernel(const IGemmArgs p, uint32_t src_bytes, uint32_t wgt_bytes) {
  constexpr int dbg = EVK_C1_DMA_ABL;
  constexpr int BM = 128, WAVES_N = 2, NW = WAVES_M * WAVES_N, WM = BM / WAVES_M, MB = WM / 32, WN = BN / 2, NB = WN / 32;
  static_assert(NST == 2 || NST == 3, "ring depth");
  static_assert(WAVES_M == 4 || WAVES_M == 2, "row waves");
  constexpr int kAStage = BM * kC1Row, kBPlane = BN * kRowBytes, kStage = kAStage + 2 * kBPlane;
  constexpr int AI = BM * kC1Row / 1024 / NW;     // activation DMA instructions per wave and stage (8 rows each)
  constexpr int BI = 2 * kBPlane / 1024 / NW;     // weight-plane DMA instructions per wave and stage (16 rows of a plane)
  constexpr int PER = AI + BI;
  static_assert(AI >= 2 && BI >= 1 && NB >= 1 && MB >= 1, "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_c1[];

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  const int ntiles = p.tiles_m * p.tiles_n;
  // (timing probe, dbg & 16: the grid is doubled and each copy of a tile runs one half of the K loop — what a split of the
  // reduction over two co-resident workgroups would cost per half; results are garbage)
  const int khalf = (dbg & 16) ? (int)blockIdx.x >= ntiles : 0;
  const int bid = xcd_remap((int)blockIdx.x - khalf * ntiles, ntiles);
  const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk_all = p.Kpad / BK3;
  const int kbeg = (dbg & 16) ? khalf * (nk_all / 2) : 0;
  const int nk = (dbg & 16) ? kbeg + nk_all / 2 : nk_all;

  const i32x4 rs_a = make_rsrc(p.src, src_bytes), rs_b = make_rsrc(p.wgt3, wgt_bytes);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_c1;

  // ---- per-lane DMA source offsets (constant over the K loop; the K step advances the scalar offset)
  uint32_t a_voff[AI], b_voff[BI];
#pragma unroll
  for (int t = 0; t < AI; ++t) {
    const int row = 8 * (AI * wave + t) + (lane >> 3);  // row of the tile this lane's 16 bytes belong to
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
    const int s = 64 * (BI * wave + t) + lane;  // 16-byte slot among the stage's 2 * BN * 4 weight slots
    const int pt = s / (BN * 4);
    const int row = (s - pt * BN * 4) >> 2;
    const int c = (s & 3) ^ ((row >> 2) & 3);
    const int co = n0 + row;
    b_voff[t] = co < p.Cd ? (uint32_t)pt * plane_bytes + (uint32_t)co * (uint32_t)p.Kpad * 2u + (uint32_t)c * 16u : kDmaOOB;
  }

  auto issue = [&](int kt, int slot) {
    const uint32_t S = lds0 + slot * kStage;
    const uint32_t ka = (uint32_t)kt * kC1Row, kb = (uint32_t)kt * kRowBytes;
    if (!(dbg & 1)) {
#pragma unroll
      for (int t = 0; t < AI; ++t) dma16s(rs_a, S + (AI * wave + t) * 1024, a_voff[t], ka);
    }
    if (!(dbg & 2)) {
#pragma unroll
      for (int t = 0; t < BI; ++t) dma16s(rs_b, S + kAStage + (BI * wave + t) * 1024, b_voff[t], kb);
    }
  };

  // ---- fragment read offsets inside a stage (lane constants)
  uint32_t fa_off[2][2], fb_off[2];   // (row blocks of a wave are 32 rows = 4096 bytes apart: same XOR pattern)
  {
    const int row = wm * WM + li;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int h = 0; h < 2; ++h) fa_off[kk][h] = (uint32_t)c1_arow_off(row, 4 * kk + 2 * lh + h);
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

  f32x16 acc[MB][NB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- prologue: NST - 1 steps in flight
  issue(kbeg, 0);
  if (NST == 3 && nk > kbeg + 1) issue(kbeg + 1, 1);

  int slot = 0;
  for (int kt = kbeg; kt < nk; ++kt) {
    // my DMA of step kt has landed when at most the next step's instructions are outstanding; after the barrier
    // everybody's has, and everybody is done reading the stage that step kt + 2 overwrites (read in step kt - 1)
    if (NST == 3 && kt + 1 < nk) {
      wait_vmcnt<PER>();
    } else {
      wait_vmcnt<0>();
    }
    ring_barrier();
    if (kt + NST - 1 < nk) issue(kt + NST - 1, slot == 0 ? NST - 1 : slot - 1);
    const uint32_t S = opaque(lds0 + slot * kStage);
    slot = slot == NST - 1 ? 0 : slot + 1;
    if (dbg & 4) continue;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[MB][2], fb[NB][2];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
          fb[b][pt] = __builtin_bit_cast(bf16x8, lds_read16(S + fb_off[kk] + pt * kBPlane + b * 32 * kRowBytes));
#pragma unroll
      for (int a = 0; a < MB; ++a) {
        // (read as floats: a bit_cast of an ext-vector ELEMENT is miscompiled by this hipcc — seen in the ISA as one element
        // used twice; packed words only travel through split_op's scalar bit_cast)
        const f32x4 w0 = __builtin_bit_cast(f32x4, lds_read16(S + fa_off[kk][0] + a * 32 * kC1Row));
        const f32x4 w1 = __builtin_bit_cast(f32x4, lds_read16(S + fa_off[kk][1] + a * 32 * kC1Row));
        u32x4 H, L;
        uint32_t h, l, unused = 0;
        split_op<2, PK>(w0.x, w0.y, a_inv, h, l, unused); H[0] = h; L[0] = l;
        split_op<2, PK>(w0.z, w0.w, a_inv, h, l, unused); H[1] = h; L[1] = l;
        split_op<2, PK>(w1.x, w1.y, a_inv, h, l, unused); H[2] = h; L[2] = l;
        split_op<2, PK>(w1.z, w1.w, a_inv, h, l, unused); H[3] = h; L[3] = l;
        fa[a][0] = __builtin_bit_cast(bf16x8, H);
        fa[a][1] = __builtin_bit_cast(bf16x8, L);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[a][b] = mfma_np<2>(fb[b][kHB[t]], fa[a][kHA[t]], acc[a][b]);
    }
  }
  if (dbg & 8) {
    if (acc[0][0][0] == 12345.f) p.dst[0] = 0.f;
    return;
  }
  igemm_scale_acc<MB, NB>(acc, out_scale);
  if (p.bn_part) {
    __syncthreads();   // the ring becomes the statistics epilogue's scratch: every wave is done reading the last stage
    igemm_epilogue_stats<MB, NB, WM, WN, WAVES_M, WAVES_N>(p, acc, m0, n0, wm, wn, li, lh, reinterpret_cast<float*>(smem_c1));
    return;
  }
  AmaxAcc amax_l{0u, p.out_amax != nullptr};
  igemm_epilogue<MB, NB, WM, WN>(p, acc, m0, n0, wm, wn, li, lh, amax_l);
  if (p.out_amax) amax_commit(p.out_amax, amax_l.m);
}

template <int BN, bool PK, int NST, int WAVES_M>
static int launch_c1(IGemmArgs& a, hipStream_t stream) {
  constexpr int dbg = EVK_C1_DMA_ABL;
  constexpr int BM = 128;
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.Cd, BN);
  bn_stats_setup(a, BM, BN, WAVES_M, a.tiles_m);
  size_t lds = (size_t)NST * (BM * kC1Row + 2 * BN * kRowBytes);
  const size_t scratch = ((size_t)2 * WAVES_M * 32 * (BN / 2 + 4) + (size_t)3 * 2 * 3 * (BN / 2)) * sizeof(float);
  if (lds < scratch) lds = scratch;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_dma_kernel<BN, PK, NST, WAVES_M>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const long long nwg = (long long)a.tiles_m * a.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffLL) {
    set_error("conv1x1_dma: bad grid %lld", nwg);
    return EVK_E_INVALID;
  }
  const unsigned long long sb = (unsigned long long)a.N * a.Hs * a.Ws * a.Cs * 4ull;
  const unsigned long long wb = 2ull * a.Cd * a.Kpad * 2ull;
  hipLaunchKernelGGL((conv1x1_dma_kernel<BN, PK, NST, WAVES_M>), dim3((unsigned)((dbg & 16) ? 2 * nwg : nwg)), dim3(128 * WAVES_M), lds, stream, a,
                     (uint32_t)sb, (uint32_t)wb);
  return check_launch("conv1x1_dma");
}

template <int BN, int NST, int WAVES_M = 4>
static int launch_c1_pk(IGemmArgs& a, hipStream_t stream) {
  return a.a_packed ? launch_c1<BN, true, NST, WAVES_M>(a, stream) : launch_c1<BN, false, NST, WAVES_M>(a, stream);
}

bool conv1x1_dma_applicable(const IGemmArgs& a) {
  if (a.planes != 2 || a.kh != 1 || a.kw != 1 || (a.Cs & 31) != 0 || a.Kpad != a.Cs) return false;
  const unsigned long long sb = (unsigned long long)a.N * a.Hs * a.Ws * a.Cs * 4ull;
  const unsigned long long wb = 2ull * a.Cd * a.Kpad * 2ull;
  return sb < 0x80000000ull && wb < 0x80000000ull;  // 32-bit buffer offsets, kDmaOOB above every valid one
}

int launch_conv1x1_dma_forced(IGemmArgs& a, int bn, hipStream_t stream) {
  if (!conv1x1_dma_applicable(a)) {
    set_error("conv1x1_dma: shape not supported (1x1, Cs %% 32 == 0, f16x2 arithmetic)");
    return EVK_E_UNSUPPORTED;
  }
  if (bn == 256) return launch_c1_pk<256, 3>(a, stream);
  if (bn == 128) return launch_c1_pk<128, 3>(a, stream);
  if (bn == 64) return launch_c1_pk<64, 3>(a, stream);
  // two stages: two workgroups per CU (one's epilogue under the other's loop)
  if (bn == 2128) return launch_c1_pk<128, 2>(a, stream);
  return launch_c1_pk<64, 2>(a, stream);
  // (four-wave forms, 2 x 2 waves of 64 x BN/2 — a third fewer LDS bytes per MFMA — measured behind the eight-wave ones on
  // every shape: 182-190 vs 178-180 us on 256->256 @128^2; instantiate launch_c1_pk<BN, NST, 2> to try them again)
}

// returns 1 when this form does not apply (the caller goes on to the register-staged kernels)
int launch_conv1x1_dma(IGemmArgs& a, hipStream_t stream) {
  // EVK_C1_DMA: 0 never; 1 (default) where measured faster; 2 wherever the shape allows
  static const int mode = getenv("EVK_C1_DMA") ? atoi(getenv("EVK_C1_DMA")) : 1;
  if (mode == 0 || !conv1x1_dma_applicable(a)) return 1;
  // The three-role persistent form (conv1x1_ps2.hip: loader / compute / store waves, software-pipelined K step) takes the
  // 128^2-map layers and the short-reduction layers of the 64^2 maps (round 4's two-role persistent kernel, conv1x1_ps.hip,
  // which it superseded on every shape, was deleted in round 6) — measured (tools/ab_c1sp.py, us, best other
  // form -> this): 64 -> 256 @128^2 93 -> 78, 256 -> 256 173 -> 152, 256 -> 128 98 -> 83, 128 -> 512 @64^2 52 -> 46; level on
  // the longer reductions of the 64^2 / 32^2 maps, behind on 2048 -> 512 @16^2 (one tile per workgroup: nothing to overlap).
  // EVK_C1_PS2: 0 never, 1 (default) by that rule, 2 wherever it applies (tests)
  static const int ps2_mode = getenv("EVK_C1_PS2") ? atoi(getenv("EVK_C1_PS2")) : 1;
  if (ps2_mode != 0 && conv1x1_ps2_applicable(a) && a.Cd >= 128) {
    const int tm = ceil_div(a.M, 128), nk = a.Kpad / BK3;
    if (ps2_mode == 2 || tm >= 1024 || (tm >= 512 && nk <= 4)) {
      const int rc = launch_conv1x1_ps2(a, stream);   // (1: the column tiles do not fit this device's CUs per XCD — go on)
      if (rc != 1) return rc;
    }
  }
  if (mode == 2) return launch_conv1x1_dma_forced(a, a.Cd >= 128 ? 2128 : 2064, stream);
  // Measured on the FarSeg-R50 one-tap shapes, fp32 and packed operands, with and without the statistics epilogue
  // (tools/ab_c1dma.py, us, register-staged default -> this kernel): the two-stage ring with TWO workgroups per CU (one's
  // store burst under the other's loop) is ahead of the three-stage ring at one workgroup per CU and of the default wherever
  // the output is at least 128 channels wide: 256->256 @128^2 225-244 -> 178-185, 256->128 115-136 -> 104-107, 128->512 @64^2
  // 62-69 -> 56-60, 512->128 41-50 -> 38-40, 512->256 75-86 -> 70-77, 1024->256 @32^2 36-42 -> 35-39, level on 64->256
  // (108-138 -> 108-113), 256->1024 and 512->2048; 64-wide outputs stay on the default (256->64: 66-76 vs 72-78).
  // 16^2 maps: 64-wide column tiles where 128-wide ones leave CUs without a workgroup (2048->512: 47-55 -> 41-50).
  if (a.Cd < 128) return 1;
  const long long tm = ceil_div(a.M, 128);
  int bn = 2128;
  // (below how many 128-wide tiles the 64-wide ones are taken: swept in the step in round 5, where the chip is shared with the
  // side stream — 128 / 320 / 640 / 1100 all behind or level with 224)
  constexpr long long fill_wg = 224;
  if (tm * ceil_div(a.Cd, 128) < fill_wg) {
    if (tm * ceil_div(a.Cd, 64) < 224) return 1;   // cannot fill the chip
    // long reductions on the 16^2 maps (2048 -> 512: 64 steps, one 128 x 64 tile per CU): the software-pipelined form with
    // loader waves (conv1x1_sp.hip, ring of four) — 42.9 -> 32.8 us, 43.3 -> 33.6 with the statistics epilogue (tools/ab_c1sp.py)
    static const int sp_mode = getenv("EVK_C1_SP") ? atoi(getenv("EVK_C1_SP")) : 1;
    if (sp_mode && a.Kpad / BK3 >= 32 && conv1x1_sp_applicable(a)) return launch_conv1x1_sp_forced(a, 64, stream);
    bn = 2064;
  }
  return launch_conv1x1_dma_forced(a, bn, stream);
}

}  // namespace evk
