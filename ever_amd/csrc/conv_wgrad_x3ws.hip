// Wave-specialised, wide-tile form of the split-MFMA weight gradient (arithmetic and LDS image: see
// conv_wgrad_x3.hip / conv_igemm_x3.hip).  One 8-wave workgroup per CU computes a 128 (Cout) x 256 (k) tile
// of dw over its pixel chunk:
//   waves 0-3  matrix waves, 64 x 128 each (2 x 4 MFMA blocks): fragment reads + v_mfma_f32_32x32x16_bf16;
//   waves 4-7  staging waves: every thread gathers an 8-pixel x 4-channel micro-block of im2col(x) (256 per
//              step), threads of waves 4-5 one of dy in addition (128 per step); exact 3-way bf16 split and
//              the register transpose into the pixel-contiguous planes; two register sets deep, two LDS stages.
// Against the 128x128 single-role kernel: 25 % fewer operand elements (loads and splits) per MFMA, and the
// split no longer shares a wave with the MFMA stream.
#include "wgrad_common.hpp"
#include "x3_common.hpp"
#include <stdlib.h>
#include <type_traits>

#ifndef EVK_WG_ABL
#define EVK_WG_ABL 0   // timing ablations (tools/build_variant.sh -DEVK_WG_ABL=n; wrong results): 1 no loads, 2 no split / LDS writes,
#endif                 // 4 no fragment reads / MFMAs, 8 no stores
namespace evk {
constexpr int kWgAbl = EVK_WG_ABL;


struct WGather {
  const float* src;
  int Hs, Ws, Cs, ssh, ssw, offy, offx, coff;
  bool cvalid;
};

// W8: Wo % 8 == 0 (every map of this network): the 8 pixels of a micro-block lie in one image row, one decomposition
// (two fast divisions) per micro-block, running sums for the column and the address.  The instantiation carries no
// code of the per-pixel path — the staging waves are VALU-issue bound (PMC: 83 % VALU-busy with the matrix waves idle).
template <bool W8>
__device__ __forceinline__ void gather8(const WGradArgs& p, const WGather& g, int m0, int pend, f32x4 (&rv)[8],
                                        uint32_t& okmask) {
  okmask = 0;
  if (W8) {
    const uint32_t mm = (uint32_t)min(m0, p.M - 1);
    const uint32_t n = fdiv(mm, p.fd_hw);
    const uint32_t rem = mm - n * p.fd_hw.div;
    const uint32_t oy = fdiv(rem, p.fd_w);
    const int ox = (int)(rem - oy * p.fd_w.div);
    const int sy = (int)oy * g.ssh + g.offy;
    const bool rowok = g.cvalid && m0 < pend && (unsigned)sy < (unsigned)g.Hs;
    int sx = ox * g.ssw + g.offx;
    int off = (((int)n * g.Hs + sy) * g.Ws + sx) * g.Cs + g.coff;
    const int dsx = g.ssw, doff = g.ssw * g.Cs;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = rowok && (unsigned)sx < (unsigned)g.Ws;
      okmask |= ok ? (1u << j) : 0u;
      rv[j] = *reinterpret_cast<const f32x4*>(g.src + (ok ? off : 0));
      sx += dsx;
      off += doff;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = m0 + j;
      const uint32_t mm = (uint32_t)min(m, p.M - 1);
      const uint32_t n = fdiv(mm, p.fd_hw);
      const uint32_t rem = mm - n * p.fd_hw.div;
      const uint32_t oy = fdiv(rem, p.fd_w);
      const int ox = (int)(rem - oy * p.fd_w.div);
      const int sy = (int)oy * g.ssh + g.offy, sx = ox * g.ssw + g.offx;
      const bool ok = g.cvalid && m < pend && (unsigned)sy < (unsigned)g.Hs && (unsigned)sx < (unsigned)g.Ws;
      okmask |= ok ? (1u << j) : 0u;
      rv[j] = *reinterpret_cast<const f32x4*>(g.src + (ok ? (((int)n * g.Hs + sy) * g.Ws + sx) * g.Cs + g.coff : 0));
    }
  }
}

// split the micro-block and write channel 4*cq+e to LDS row e*Q + cq, 16-byte chunk pg, of the three planes
template <int NP>
__device__ __forceinline__ void split_store8(f32x4 (&rv)[8], uint32_t okmask, unsigned char* base, int plane_bytes, int Q,
                                             int cq, int pg, float inv) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool ok = (okmask >> j) & 1u;
    rv[j].x = ok ? rv[j].x : 0.f; rv[j].y = ok ? rv[j].y : 0.f;
    rv[j].z = ok ? rv[j].z : 0.f; rv[j].w = ok ? rv[j].w : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int off = wg_off(e * Q + cq, pg);
    u32x4 H, M, L;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint32_t h, m = 0, l = 0;
      split_np<NP>(rv[2 * t][e], rv[2 * t + 1][e], inv, h, m, l);
      H[t] = h; M[t] = m; L[t] = l;
    }
    *reinterpret_cast<u32x4*>(base + off) = H;
    if (NP >= 2) *reinterpret_cast<u32x4*>(base + plane_bytes + off) = M;
    if (NP == 3) *reinterpret_cast<u32x4*>(base + 2 * plane_bytes + off) = L;
  }
}

// Half micro-block of dy: 8 pixels x 2 channels (8-byte loads, 512 B contiguous per pixel across a wave).  dy is the
// plain [M][Cout] matrix: no tap, no spatial bound, only the pixel range.  Channel 2*cq+e goes to LDS row e*Q2 + cq.
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gather8h(const float* __restrict__ dy, int Cout, int coff, bool cvalid, int m0, int pend,
                                         f32x2v (&rv)[8], uint32_t& okmask) {
  okmask = 0;
  int off = m0 * Cout + coff;      // (both tensors are below 2^31 bytes: plan_wgrad's fits32)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool ok = cvalid && m0 + j < pend;
    okmask |= ok ? (1u << j) : 0u;
    rv[j] = *reinterpret_cast<const f32x2v*>(dy + (ok ? off : 0));
    off += Cout;
  }
}
// ---- buffer-load gathers (the f16x2 instantiation, where they measure faster: staging alone 874 -> 771 us, whole
// kernel 1326 -> 1301 / 320 -> 277 / 90 -> 80 us on 3x3x256 @128^2 / 64^2 / 32^2; under bf16x3 they are slower, 1567 ->
// 1726): 32-bit byte offsets on a buffer resource, out-of-image pixels get an
// out-of-range offset and the hardware returns zeros — no 64-bit address arithmetic, no selects on the loaded data.
// The pixel decomposition (two fast divisions) is done once per micro-block when Wo % 8 == 0 (always the case for the
// maps of this network); the generic path decomposes every pixel.
constexpr uint32_t kOOBOff = 0x80000000u;
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 bload16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
__device__ __forceinline__ f32x2v bload8(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
}
template <bool W8>
__device__ __forceinline__ void gather8_buf(const WGradArgs& p, const WGather& g, __amdgpu_buffer_rsrc_t rs, int m0, int pend,
                                            f32x4 (&rv)[8]) {
  if (W8) {
    const uint32_t mm = (uint32_t)min(m0, p.M - 1);
    const uint32_t n = fdiv(mm, p.fd_hw);
    const uint32_t rem = mm - n * p.fd_hw.div;
    const uint32_t oy = fdiv(rem, p.fd_w);
    const int ox = (int)(rem - oy * p.fd_w.div);
    const int sy = (int)oy * g.ssh + g.offy;
    const bool rowok = g.cvalid && m0 < pend && (unsigned)sy < (unsigned)g.Hs;
    int sx = ox * g.ssw + g.offx;
    const uint32_t step = (uint32_t)(g.ssw * g.Cs) << 2;
    uint32_t off = (uint32_t)((((int)n * g.Hs + sy) * g.Ws + sx) * g.Cs + g.coff) << 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = rowok && (unsigned)sx < (unsigned)g.Ws;
      rv[j] = bload16(rs, ok ? off : kOOBOff);
      sx += g.ssw;
      off += step;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = m0 + j;
      const uint32_t mm = (uint32_t)min(m, p.M - 1);
      const uint32_t n = fdiv(mm, p.fd_hw);
      const uint32_t rem = mm - n * p.fd_hw.div;
      const uint32_t oy = fdiv(rem, p.fd_w);
      const int ox = (int)(rem - oy * p.fd_w.div);
      const int sy = (int)oy * g.ssh + g.offy, sx = ox * g.ssw + g.offx;
      const bool ok = g.cvalid && m < pend && (unsigned)sy < (unsigned)g.Hs && (unsigned)sx < (unsigned)g.Ws;
      rv[j] = bload16(rs, ok ? (uint32_t)((((int)n * g.Hs + sy) * g.Ws + sx) * g.Cs + g.coff) << 2 : kOOBOff);
    }
  }
}
__device__ __forceinline__ void gather8h_buf(__amdgpu_buffer_rsrc_t rs, int Cout, int coff, bool cvalid, int m0, int pend,
                                             f32x2v (&rv)[8]) {
  uint32_t off = (uint32_t)(m0 * Cout + coff) << 2;
  const uint32_t step = (uint32_t)Cout << 2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    rv[j] = bload8(rs, (cvalid && m0 + j < pend) ? off : kOOBOff);
    off += step;
  }
}
// split the micro-block (already zero where out of range) and write channel 4*cq+e to LDS row e*Q + cq, chunk pg
template <int NP, int NCH, bool PK, typename V>
__device__ __forceinline__ void split_store_buf(V (&rv)[8], unsigned char* base, int plane_bytes, int Q, int cq, int pg,
                                                float inv) {
#pragma unroll
  for (int e = 0; e < NCH; ++e) {
    const int off = wg_off(e * Q + cq, pg);
    u32x4 H, M, L;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint32_t h, m = 0, l = 0;
      split_op<NP, PK>(rv[2 * t][e], rv[2 * t + 1][e], inv, h, m, l);
      H[t] = h; M[t] = m; L[t] = l;
    }
    *reinterpret_cast<u32x4*>(base + off) = H;
    if (NP >= 2) *reinterpret_cast<u32x4*>(base + plane_bytes + off) = M;
    if (NP == 3) *reinterpret_cast<u32x4*>(base + 2 * plane_bytes + off) = L;
  }
}

template <int NP>
__device__ __forceinline__ void split_store8h(f32x2v (&rv)[8], uint32_t okmask, unsigned char* base, int plane_bytes, int Q2,
                                              int cq, int pg, float inv) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool ok = (okmask >> j) & 1u;
    rv[j].x = ok ? rv[j].x : 0.f;
    rv[j].y = ok ? rv[j].y : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int off = wg_off(e * Q2 + cq, pg);
    u32x4 H, M, L;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint32_t h, m = 0, l = 0;
      split_np<NP>(rv[2 * t][e], rv[2 * t + 1][e], inv, h, m, l);
      H[t] = h; M[t] = m; L[t] = l;
    }
    *reinterpret_cast<u32x4*>(base + off) = H;
    if (NP >= 2) *reinterpret_cast<u32x4*>(base + plane_bytes + off) = M;
    if (NP == 3) *reinterpret_cast<u32x4*>(base + 2 * plane_bytes + off) = L;
  }
}

// PKX / PKD: x / dy arrive packed (f16x2 only; x3_common.hpp)
template <int BM, int BN, int NP, bool W8, bool PKX = false, bool PKD = false>
__global__ __launch_bounds__(512) void conv_wgrad_x3ws_kernel(const WGradArgs p) {
  static_assert(NP == 2 || (!PKX && !PKD), "packed operands exist for the f16x2 arithmetic only");
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MB = WM / 32, NB = WN / 32;
  constexpr int QA = BM / 4, QB = BN / 4;
  constexpr int kStage = 3 * (BM + BN) * kRowBytes;
  static_assert(QB * 4 == 256 && QA * 4 <= 256, "staging waves: one im2col micro-block per thread, dy on the first ones");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tiles_co * p.tiles_k;
  const int z = bid / ntile;
  const int tile = bid - z * ntile;
  const int tile_k = tile % p.tiles_k;
  const int tile_co = tile / p.tiles_k;
  const int co0 = tile_co * BM, k0 = tile_k * BN;
  const int pbeg = z * p.chunk;
  const int pend = min(p.M, pbeg + p.chunk);
  const int nk = (pend - pbeg + BKP - 1) / BKP;
  const int tid = threadIdx.x;

  if (tid >= 256) {
    // ------------------------------------------------------------------ staging waves
    const int ptid = tid - 256;
    // every staging thread: one im2col micro-block (8 px x 4 ch) + one HALF dy micro-block (8 px x 2 ch) — the
    // dy tile dealt over all four waves (with whole dy micro-blocks on two waves those two set the pace:
    // measured 51 % matrix-pipe occupancy, the matrix waves waiting at the barrier)
    constexpr int QA2 = BM / 2;
    static_assert(QA2 * 4 == 256, "one half dy micro-block per staging thread");
    WGather gb;
    const int bcq = ptid % QB, bpg = ptid / QB;
    const int acq = ptid % QA2, apg = ptid / QA2;
    const int a_coff = co0 + acq * 2;
    const bool a_cvalid = a_coff < p.Cout;
    {
      gb.src = p.x; gb.Hs = p.H; gb.Ws = p.W; gb.Cs = p.Cin; gb.ssh = p.sh; gb.ssw = p.sw;
      const int q = (k0 >> 2) + bcq;
      gb.cvalid = q * 4 < p.Ktot;
      gb.offy = gb.offx = gb.coff = 0;
      if (gb.cvalid) {
        const int tap = q / p.cpt;
        gb.coff = (q - tap * p.cpt) * 4;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        gb.offy = ky * p.dh - p.ph;
        gb.offx = kx * p.dw - p.pw;
      }
    }
    f32x2v ra[2][8];
    f32x4 rb[2][8];
    uint32_t oka[2] = {0, 0}, okb[2] = {0, 0};
    constexpr bool kBuf = NP == 2;   // buffer loads with out-of-range zero fill
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.N * p.H * p.W * p.Cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.M * p.Cout * 4, 0x00020000);
    float x_inv = 1.f, dy_inv = 1.f;   // f16x2: 1 / operand scales
    if constexpr (NP == 2) {
      x_inv = op_scale(act_absmax(p.x_scale)).inv;
      dy_inv = op_scale(act_absmax(p.dy_scale)).inv;
    }

    auto load = [&](auto SET, int kt) {
      constexpr int s = decltype(SET)::value;
      const int pix0 = pbeg + kt * BKP;
      if (kWgAbl & 1) return;   // ablation: no global loads
      if constexpr (kBuf) {
        gather8_buf<W8>(p, gb, rs_x, pix0 + bpg * 8, pend, rb[s]);
        gather8h_buf(rs_dy, p.Cout, a_coff, a_cvalid, pix0 + apg * 8, pend, ra[s]);
      } else {
        gather8<W8>(p, gb, pix0 + bpg * 8, pend, rb[s], okb[s]);
        gather8h(p.dy, p.Cout, a_coff, a_cvalid, pix0 + apg * 8, pend, ra[s], oka[s]);
      }
    };
    auto store = [&](auto SET, int stage) {
      constexpr int s = decltype(SET)::value;
      unsigned char* Ab = smem3 + stage * kStage;
      unsigned char* Bb = Ab + 3 * BM * kRowBytes;
      if (kWgAbl & 2) return;   // ablation: no split, no LDS writes
      if constexpr (kBuf) {
        split_store_buf<NP, 4, PKX>(rb[s], Bb, BN * kRowBytes, QB, bcq, bpg, x_inv);
        split_store_buf<NP, 2, PKD>(ra[s], Ab, BM * kRowBytes, QA2, acq, apg, dy_inv);
      } else {
        split_store8<NP>(rb[s], okb[s], Bb, BN * kRowBytes, QB, bcq, bpg, x_inv);
        split_store8h<NP>(ra[s], oka[s], Ab, BM * kRowBytes, QA2, acq, apg, dy_inv);
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    if (nk > 0) {
      load(S0{}, 0);
      if (1 < nk) load(S1{}, 1);
      store(S0{}, 0);
      if (2 < nk) load(S0{}, 2);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      if (kt + 1 < nk) {
        store(S1{}, 1);
        if (kt + 3 < nk) load(S1{}, kt + 3);
      }
      __syncthreads();
      if (kt + 1 < nk) {
        if (kt + 2 < nk) {
          store(S0{}, 0);
          if (kt + 4 < nk) load(S0{}, kt + 4);
        }
        __syncthreads();
      }
    }
    return;
  }

  // -------------------------------------------------------------------- matrix waves
  __builtin_amdgcn_s_setprio(3);
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  int fa_off[MB][2], fb_off[NB][2];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fa_off[a][kk] = wg_off(wm * WM + a * 32 + li, 2 * kk + lh);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    // im2col LDS row e * QB + q holds channel 4q + e of the k range (staging permutation).  Matrix wave wn takes q in
    // [32 wn, 32 wn + 32) of ALL four e-blocks as its NB = 4 column blocks: lane li then owns the four CONSECUTIVE
    // k columns 4 (32 wn + li) + {0..3} across its blocks, i.e. one 16-byte store per accumulator row.
    for (int kk = 0; kk < 2; ++kk) fb_off[b][kk] = 3 * BM * kRowBytes + wg_off(b * QB + wn * 32 + li, 2 * kk + lh);

  auto half_step = [&](const unsigned char* S, int kk) {
    bf16x8 fa[MB][3], fb[NB][3];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
        fa[a][pt] = *reinterpret_cast<const bf16x8*>(S + pt * BM * kRowBytes + fa_off[a][kk]);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
        fb[b][pt] = *reinterpret_cast<const bf16x8*>(S + pt * BN * kRowBytes + fb_off[b][kk]);
#pragma unroll
    for (int t = 0; t < X3Prod<NP>::N; ++t)
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a][b] = mfma_np<NP>(fa[a][x3_pa(NP, t)], fb[b][x3_pb(NP, t)], acc[a][b]);
  };

  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* S = smem3 + (kt & 1) * kStage;
    if (!(kWgAbl & 4)) {      // ablation: no fragment reads / MFMAs
      half_step(S, 0);
      half_step(S, 1);
    }
    __syncthreads();
  }
  if (kWgAbl & 8) {           // ablation: no stores
    if (acc[0][0][0] == 12345.f) p.out[0] = 0.f;
    return;
  }

  if constexpr (NP == 2) {   // f16x2: back to the operands' units
    const float sc = op_scale(act_absmax(p.x_scale)).s * op_scale(act_absmax(p.dy_scale)).s;
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[a][b] *= sc;
  }
  float* out = p.out + (size_t)z * p.Cout * p.Ktot;
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ra_ = wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int row = co0 + 2 * (ra_ % (BM / 2)) + ra_ / (BM / 2);   // inverse of the dy staging permutation
      if (row >= p.Cout) continue;
      static_assert(NB == 4 && QB == 64, "column blocks = the four channels of a staging micro-block");
      const int col = k0 + 4 * (wn * 32 + li);
      if (col + 3 < p.Ktot) {   // Ktot = taps * Cin with Cin % 4 == 0: rows are 16-byte aligned
        const f32x4 v = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
        *reinterpret_cast<f32x4*>(out + (size_t)row * p.Ktot + col) = v;
      } else {
#pragma unroll
        for (int b = 0; b < NB; ++b)
          if (col + b < p.Ktot) out[(size_t)row * p.Ktot + col + b] = acc[a][b][r];
      }
    }
}

template <int NP, bool W8, bool PKX = false, bool PKD = false>
static int launch_wgrad_x3ws_t(const WGradArgs& b, hipStream_t stream) {
  constexpr int BM = 128, BN = 256;
  const size_t lds = (size_t)2 * 3 * (BM + BN) * kRowBytes;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_x3ws_kernel<BM, BN, NP, W8, PKX, PKD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL((conv_wgrad_x3ws_kernel<BM, BN, NP, W8, PKX, PKD>), dim3(b.tiles_co * b.tiles_k * b.splitk), dim3(512), lds, stream, b);
  return check_launch("conv_wgrad_x3ws");
}

int launch_wgrad_x3ws(const WGradArgs& a, hipStream_t stream) {
  WGradArgs b = a;
  const bool w8 = (a.Wo & 7) == 0;
  if (a.planes == 1) return w8 ? launch_wgrad_x3ws_t<1, true>(b, stream) : launch_wgrad_x3ws_t<1, false>(b, stream);
  if (a.planes == 2) {
    const int pk = (a.x_packed ? 1 : 0) | (a.dy_packed ? 2 : 0);
    if (pk == 3) return w8 ? launch_wgrad_x3ws_t<2, true, true, true>(b, stream) : launch_wgrad_x3ws_t<2, false, true, true>(b, stream);
    if (pk == 2) return w8 ? launch_wgrad_x3ws_t<2, true, false, true>(b, stream) : launch_wgrad_x3ws_t<2, false, false, true>(b, stream);
    if (pk == 1) return w8 ? launch_wgrad_x3ws_t<2, true, true, false>(b, stream) : launch_wgrad_x3ws_t<2, false, true, false>(b, stream);
    return w8 ? launch_wgrad_x3ws_t<2, true>(b, stream) : launch_wgrad_x3ws_t<2, false>(b, stream);
  }
  return w8 ? launch_wgrad_x3ws_t<3, true>(b, stream) : launch_wgrad_x3ws_t<3, false>(b, stream);
}

}  // namespace evk
