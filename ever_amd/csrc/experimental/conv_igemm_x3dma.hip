// LDS-DMA form of the bf16-split implicit GEMM for ONE-TAP convolutions (1x1 kernels, any stride; forward and data
// gradient): dst[m][co] = sum_k src[row(m)][k] * w[co][k]   (arithmetic: conv_igemm_x3.hip).
//
// What differs from the register-staged kernels (conv_igemm_x3 / x3ws):
//  * BOTH operands go global -> LDS by `buffer_load_dwordx4 ... lds` (LDS-DMA): no VGPR round trip, no staging VALU,
//    no ds_write.  Activations land as RAW fp32 rows, the weights as their pre-split bf16 planes.  The DMA writes
//    lane-linearly (wave-uniform LDS base + 16 B x lane), so the bank-conflict-free LDS images are produced by
//    permuting the per-lane SOURCE chunk and applying the same XOR on the fragment reads.  Rows past M / columns
//    past Cd / source pixels outside the image are given an out-of-range buffer offset: the DMA writes zeros.
//  * the exact 3-way bf16 split of the activations happens in the MATRIX waves, on the fragment a lane has just read
//    (8 floats -> three bf16x8 operands, ~44 VALU), in the shadow of that wave's own MFMAs: a wave owns WM rows x ALL
//    BN columns of the tile, so every activation element is split exactly once per tile and each split fragment
//    feeds 6 x NB MFMAs (NB = BN / 32).
//  * no wave specialisation: 4 waves per workgroup, one per SIMD; the ring of NST LDS stages is filled NST-1 steps
//    ahead, one s_barrier per K step, counted vmcnt (the DMA of later steps stays in flight across the barrier).
// EXPERIMENTAL (off by default, EVK_X3_DMA=1): correct (tests/test_dma_gpu.py) and at parity with the register-staged
// kernels; the measured breakdown (compute ~145 us + unhidden DMA wait ~55 us + store burst ~58 us on 256->256 @128^2)
// says these layers need their three parts OVERLAPPED, not any one of them made faster.
#include "../igemm_common.hpp"
#include "../x3_common.hpp"
#include <stdlib.h>

namespace evk {

constexpr int kARow = BK3 * 4;  // bytes of one fp32 activation row of a K step (128)

// byte offset of 16-byte chunk c (0..7) of fp32 row `row` in the activation image: rows are 128 B, two per 256-byte
// bank row; the XOR puts the same logical chunk of 16 consecutive rows on 16 distinct 16-byte slots
__device__ __forceinline__ int arow_off(int row, int c) { return row * kARow + ((c ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, uint32_t voff, uint32_t soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0,
                                           0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


constexpr uint32_t kOOB = 0x80000000u;
using S0 = std::integral_constant<int, 0>;
using S1 = std::integral_constant<int, 1>;  // beyond every buffer's num_records: the DMA writes zeros

template <int BM, int BN, int NST>
__global__ __launch_bounds__(256) void conv_igemm_x3dma_kernel(const IGemmArgs p, uint32_t src_bytes, uint32_t wgt_bytes, int dbg) {
  constexpr int WM = BM / 4, MB = WM / 32, NB = BN / 32;
  constexpr int kAStage = BM * kARow, kBStage = 3 * BN * kRowBytes, kStage = kAStage + kBStage;
  constexpr int AI = BM / 32;           // activation DMA instructions per wave and stage (8 rows each)
  constexpr int BI = 3 * BN * 4 / 256;  // weight-plane DMA instructions per wave and stage (16 rows of one plane each)
  constexpr int PER = AI + BI;
  static_assert(WM % 32 == 0 && BN % 64 == 0, "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_dma[];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int ntiles = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap((int)blockIdx.x, ntiles);
  const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = p.Kpad / BK3;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt3, 0, (int)wgt_bytes, 0x00020000);

  // ---- per-lane DMA source offsets (constant over the K loop; the K step advances the scalar offset)
  uint32_t a_voff[AI], b_voff[BI];
#pragma unroll
  for (int t = 0; t < AI; ++t) {
    const int row = 8 * (AI * wave + t) + (lane >> 3);  // row of the tile this lane's 16 bytes belong to
    const int c = (lane & 7) ^ ((row >> 1) & 7);        // source chunk that lands on LDS chunk (lane & 7)
    const int m = m0 + row;
    uint32_t off = kOOB;
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
    const int s = 64 * (BI * wave + t) + lane;  // 16-byte slot among the stage's 3 * BN * 4 weight slots
    const int pt = s / (BN * 4);
    const int row = (s - pt * BN * 4) >> 2;
    const int c = (s & 3) ^ ((row >> 2) & 3);
    const int co = n0 + row;
    b_voff[t] = co < p.Cd ? (uint32_t)pt * plane_bytes + (uint32_t)co * (uint32_t)p.Kpad * 2u + (uint32_t)c * 16u : kOOB;
  }

  auto issue = [&](int kt) {
    unsigned char* S = smem_dma + (kt % NST) * kStage;
    const uint32_t ka = (uint32_t)kt * kARow, kb = (uint32_t)kt * kRowBytes;
    if (!(dbg & 1)) {
#pragma unroll
      for (int t = 0; t < AI; ++t) dma16(rs_a, S + (AI * wave + t) * 1024, a_voff[t], ka);
    }
    if (!(dbg & 2)) {
#pragma unroll
      for (int t = 0; t < BI; ++t) dma16(rs_b, S + kAStage + (BI * wave + t) * 1024, b_voff[t], kb);
    }
  };

  // ---- fragment read offsets
  int fa_off[MB][2][2], fb_off[NB][2];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int h = 0; h < 2; ++h) fa_off[a][kk][h] = arow_off(wave * WM + a * 32 + li, 4 * kk + 2 * lh + h);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fb_off[b][kk] = kAStage + plane_off(b * 32 + li, 2 * kk + lh);

  f32x16 acc[MB][NB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- prologue: NST-1 steps in flight
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(s);

  for (int kt = 0; kt < nk; ++kt) {
    // my DMA of step kt has landed when at most the later steps' instructions are outstanding; after the barrier
    // everybody's has, and everybody is done reading the stage that step kt + NST - 1 overwrites (read in step kt - 1)
    if (kt + NST - 1 <= nk) {
      wait_vmcnt<(NST - 2) * PER>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (kt + NST - 1 < nk) issue(kt + NST - 1);
    const unsigned char* S = smem_dma + (kt % NST) * kStage;
    if (dbg & 4) continue;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[MB][3], fb[NB][3];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int pt = 0; pt < 3; ++pt)
          fb[b][pt] = *reinterpret_cast<const bf16x8*>(S + pt * BN * kRowBytes + fb_off[b][kk]);
#pragma unroll
      for (int a = 0; a < MB; ++a) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(S + fa_off[a][kk][0]);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(S + fa_off[a][kk][1]);
        u32x4 H, M, L;
        {
          uint32_t h, m, l;
          split2(v0.x, v0.y, h, m, l); H[0] = h; M[0] = m; L[0] = l;
          split2(v0.z, v0.w, h, m, l); H[1] = h; M[1] = m; L[1] = l;
          split2(v1.x, v1.y, h, m, l); H[2] = h; M[2] = m; L[2] = l;
          split2(v1.z, v1.w, h, m, l); H[3] = h; M[3] = m; L[3] = l;
        }
        fa[a][0] = __builtin_bit_cast(bf16x8, H);
        fa[a][1] = __builtin_bit_cast(bf16x8, M);
        fa[a][2] = __builtin_bit_cast(bf16x8, L);
      }
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b][kPB[t]], fa[a][kPA[t]], acc[a][b], 0, 0, 0);
    }
  }
  if (dbg & 8) {
    if (acc[0][0][0] == 12345.f) p.dst[0] = 0.f;
    return;
  }
  igemm_epilogue<MB, NB, WM, BN>(p, acc, m0, n0, wave, 0, li, lh);
}

template <int BM, int BN, int NST>
static int launch_dma(IGemmArgs& a, hipStream_t stream) {
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.Cd, BN);
  const size_t lds = (size_t)NST * (BM * kARow + 3 * BN * kRowBytes);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_x3dma_kernel<BM, BN, NST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const long long nwg = (long long)a.tiles_m * a.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffLL) {
    set_error("conv_igemm_x3dma: bad grid %lld", nwg);
    return EVK_E_INVALID;
  }
  const unsigned long long sb = (unsigned long long)a.N * a.Hs * a.Ws * a.Cs * 4ull;
  const unsigned long long wb = 3ull * a.Cd * a.Kpad * 2ull;
  hipLaunchKernelGGL((conv_igemm_x3dma_kernel<BM, BN, NST>), dim3((unsigned)nwg), dim3(256), lds, stream, a, (uint32_t)sb,
                     (uint32_t)wb, getenv("EVK_X3_DMA_DBG") ? atoi(getenv("EVK_X3_DMA_DBG")) : 0);
  return check_launch("conv_igemm_x3dma");
}



// returns 1 when this form does not apply (the caller goes on to the register-staged kernels)
int launch_igemm_x3dma(IGemmArgs& a, hipStream_t stream) {
  // EVK_X3_DMA: 0 (default) never; 1 one-tap convolutions whose reduction is a multiple of 32 channels.
  // OFF by default: measured at parity with the register-staged kernels (tools/ab_conv1x1.py: within +-5 % on all eight
  // FarSeg 1x1 shapes), and its ablation switches (EVK_X3_DMA_DBG: 1 no activation DMA, 2 no weight DMA, 4 no
  // compute, 8 no stores) are how DESIGN 2.2c's breakdown of these layers was measured.
  // EVK_X3_DMA_CFG picks a tile / ring for A/B runs (0 = by shape)
  static const int mode = getenv("EVK_X3_DMA") ? atoi(getenv("EVK_X3_DMA")) : 0;
  static const int cfg = getenv("EVK_X3_DMA_CFG") ? atoi(getenv("EVK_X3_DMA_CFG")) : 0;
  if (mode == 0 || a.bn_want || a.planes == 2) return 1;   // (no statistics epilogue in this kernel)
  if (a.kh != 1 || a.kw != 1 || (a.Cs & 31) != 0 || a.Kpad != a.Cs) return 1;
  const unsigned long long sb = (unsigned long long)a.N * a.Hs * a.Ws * a.Cs * 4ull;
  const unsigned long long wb = 3ull * a.Cd * a.Kpad * 2ull;
  if (sb >= 0x80000000ull || wb >= 0x80000000ull) return 1;  // 32-bit buffer offsets, kOOB above every valid one
  switch (cfg) {
    case 1: return launch_dma<128, 128, 2>(a, stream);
    case 2: return launch_dma<128, 128, 3>(a, stream);
    case 3: return launch_dma<256, 128, 2>(a, stream);
    case 4: return launch_dma<128, 64, 2>(a, stream);
    case 5: return launch_dma<128, 64, 3>(a, stream);
    case 6: return launch_dma<128, 256, 2>(a, stream);
    default: break;
  }
  if (a.Cd <= 64) return launch_dma<128, 64, 3>(a, stream);
  return launch_dma<128, 128, 2>(a, stream);
}

}  // namespace evk
