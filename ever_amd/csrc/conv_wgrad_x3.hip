// Convolution weight gradient on the bf16 matrix pipe at fp32-grade accuracy (3-way bf16 operand split,
// six partial products per product; arithmetic argument in conv_igemm_x3.hip).
//
//   dw[co][k] = sum_m dy[m][co] * im2col(x)[m][k]        m = (n, oy, ox),  k = (ky, kx, ci)
//
// Same reference call sites as conv_wgrad.hip (aten::convolution_backward(weight) of every nn.Conv2d on
// the path).  GEMM view: rows = Cout, cols = kh*kw*Cin, reduction = pixels, 32 pixels per step.
//
// v_mfma_f32_32x32x16_bf16 wants, per lane, EIGHT reduction-consecutive bf16 values of one row, but both
// operands are pixel-major in HBM ([pixel][channel]).  The transpose happens in registers: a staging
// thread owns a micro-block of 8 consecutive pixels x 4 consecutive channels (eight 16-byte global loads,
// coalesced across the lanes along the channel axis), splits its 32 values, and writes, per channel and
// plane, one 16-byte run of 8 pixels into the LDS image [plane][channel row][32 pixels] bf16.  Channel
// 4*cq + e of the tile is stored at LDS row e*(B/4) + cq: the lanes of one ds_write_b128 (fixed e) then hit
// consecutive rows = distinct bank slots; the epilogue undoes the permutation.  Threads 0..BM-1 stage dy,
// threads BM..BM+BN-1 stage im2col(x): both are the same gather with different (wave-uniform) parameters.
// Split over pixel ranges (grid.y) into the caller's workspace + splitk_reduce, as the fp32 kernel.
#include "wgrad_common.hpp"
#include "x3_common.hpp"
#include <stdlib.h>

namespace evk {

template <int BM, int BN, int WAVES_M, int WAVES_N, int NPX>
__global__ __launch_bounds__(256) void conv_wgrad_x3_kernel(const WGradArgs p) {
  constexpr int NP = X3Mode<NPX>::NP;
  constexpr bool PK = X3Mode<NPX>::PK;   // one or both operands arrive packed (p.x_packed / p.dy_packed; x3_common.hpp)
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MB = WM / 32, NB = WN / 32;
  constexpr int QA = BM / 4, QB = BN / 4;  // channel quads per operand
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(BM % 64 == 0 && BN % 64 == 0 && BM + BN <= 256, "one micro-block per thread, wave-uniform roles");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  unsigned char* const Ap = smem3;                      // [3][BM][64 B]
  unsigned char* const Bp = smem3 + 3 * BM * kRowBytes;  // [3][BN][64 B]

  // XCD-aware order: all (co, k) tiles of one pixel chunk z run on the same XCD, so the chunk's dy / x
  // rows are fetched into ONE L2 instead of eight (PMC: 3.4x the algorithmic bytes with the default order)
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tiles_co * p.tiles_k;
  const int z = bid / ntile;
  const int tile = bid - z * ntile;
  const int tile_k = tile % p.tiles_k;
  const int tile_co = tile / p.tiles_k;
  const int co0 = tile_co * BM, k0 = tile_k * BN;
  const int pbeg = z * p.chunk;
  const int pend = min(p.M, pbeg + p.chunk);

  const int tid = threadIdx.x;
  // ---- staging role (wave-uniform): which tensor this thread gathers, and how
  const bool roleA = tid < BM;
  const bool staged = tid < BM + BN;
  const int idx = roleA ? tid : tid - BM;
  const int Q = roleA ? QA : QB;
  const int cq = idx % Q;  // channel quad
  const int pg = idx / Q;  // pixel group (8 pixels) 0..3
  const float* const src = roleA ? p.dy : p.x;
  const int Hs = roleA ? p.Ho : p.H, Ws = roleA ? p.Wo : p.W, Cs = roleA ? p.Cout : p.Cin;
  const int ssh = roleA ? 1 : p.sh, ssw = roleA ? 1 : p.sw;
  int offy = 0, offx = 0, coff = 0;
  bool cvalid = false;
  if (roleA) {
    coff = co0 + cq * 4;
    cvalid = coff < p.Cout;
  } else if (staged) {
    const int q = (k0 >> 2) + cq;
    cvalid = q * 4 < p.Ktot;
    if (cvalid) {
      const int tap = q / p.cpt;
      coff = (q - tap * p.cpt) * 4;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      offy = ky * p.dh - p.ph;
      offx = kx * p.dw - p.pw;
    }
  }
  unsigned char* const st_base = (roleA ? Ap : Bp);
  const int st_plane = (roleA ? BM : BN) * kRowBytes;

  f32x4 rv[8];
  uint32_t okmask = 0;
  float st_inv = 1.f, out_scale = 1.f;   // f16x2: 1 / scale of the operand this thread stages; product of both scales
  if constexpr (NP == 2) {
    const OpScale sx = op_scale(act_absmax(p.x_scale)), sd = op_scale(act_absmax(p.dy_scale));
    st_inv = roleA ? sd.inv : sx.inv;
    out_scale = sx.s * sd.s;
  }
  const bool st_packed = PK && (roleA ? p.dy_packed : p.x_packed) != 0;   // wave-uniform

  auto load_tiles = [&](int pix0) {
    okmask = 0;
    const int m0 = pix0 + pg * 8;
    if ((p.Wo & 7) == 0) {
      // the 8 pixels share an output row: one (n, oy, ox) decomposition per micro-block
      const uint32_t mm = (uint32_t)min(m0, p.M - 1);
      const uint32_t n = fdiv(mm, p.fd_hw);
      const uint32_t rem = mm - n * p.fd_hw.div;
      const uint32_t oy = fdiv(rem, p.fd_w);
      const int ox = (int)(rem - oy * p.fd_w.div);
      const int sy = (int)oy * ssh + offy;
      const bool rowok = cvalid && m0 < pend && (unsigned)sy < (unsigned)Hs;
      const int base = ((int)n * Hs + sy) * Ws * Cs + coff;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int sx = (ox + j) * ssw + offx;
        const bool ok = rowok && (unsigned)sx < (unsigned)Ws;
        okmask |= ok ? (1u << j) : 0u;
        rv[j] = *reinterpret_cast<const f32x4*>(src + (ok ? base + sx * Cs : 0));
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
        const int sy = (int)oy * ssh + offy, sx = ox * ssw + offx;
        const bool ok = cvalid && m < pend && (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws;
        okmask |= ok ? (1u << j) : 0u;
        rv[j] = *reinterpret_cast<const f32x4*>(src + (ok ? (((int)n * Hs + sy) * Ws + sx) * Cs + coff : 0));
      }
    }
  };

  auto store_tiles = [&]() {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = (okmask >> j) & 1u;
      rv[j].x = ok ? rv[j].x : 0.f; rv[j].y = ok ? rv[j].y : 0.f;
      rv[j].z = ok ? rv[j].z : 0.f; rv[j].w = ok ? rv[j].w : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = e * Q + cq;
      const int off = wg_off(row, pg);
      u32x4 H, M, L;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        uint32_t h, m = 0, l = 0;
        if constexpr (PK) {
          if (st_packed) split_op<NP, true>(rv[2 * t][e], rv[2 * t + 1][e], st_inv, h, m, l);
          else split_np<NP>(rv[2 * t][e], rv[2 * t + 1][e], st_inv, h, m, l);
        } else {
          split_np<NP>(rv[2 * t][e], rv[2 * t + 1][e], st_inv, h, m, l);
        }
        H[t] = h; M[t] = m; L[t] = l;
      }
      *reinterpret_cast<u32x4*>(st_base + off) = H;
      if (NP >= 2) *reinterpret_cast<u32x4*>(st_base + st_plane + off) = M;
      if (NP == 3) *reinterpret_cast<u32x4*>(st_base + 2 * st_plane + off) = L;
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

  int fa_off[MB][2], fb_off[NB][2];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fa_off[a][kk] = wg_off(wm * WM + a * 32 + li, 2 * kk + lh);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) fb_off[b][kk] = wg_off(wn * WN + b * 32 + li, 2 * kk + lh);

  auto half_step = [&](int kk) {
    bf16x8 fa[MB][3], fb[NB][3];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
        fa[a][pt] = *reinterpret_cast<const bf16x8*>(Ap + pt * BM * kRowBytes + fa_off[a][kk]);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int pt = 0; pt < NP; ++pt)
        fb[b][pt] = *reinterpret_cast<const bf16x8*>(Bp + pt * BN * kRowBytes + fb_off[b][kk]);
#pragma unroll
    for (int t = 0; t < X3Prod<NP>::N; ++t)
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a][b] = mfma_np<NP>(fa[a][x3_pa(NP, t)], fb[b][x3_pb(NP, t)], acc[a][b]);
  };

  const int nk = (pend - pbeg + BKP - 1) / BKP;
  if (nk > 0 && staged) {
    load_tiles(pbeg);
    store_tiles();
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more && staged) load_tiles(pbeg + (kt + 1) * BKP);
    half_step(0);
    half_step(1);
    __syncthreads();
    if (more) {
      if (staged) store_tiles();
      __syncthreads();
    }
  }

  // D rows = A rows (dy channels), D cols = B rows (im2col columns), both under the staging permutation
  float* out = p.out + (size_t)z * p.Cout * p.Ktot;
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ra = wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int row = co0 + 4 * (ra % QA) + ra / QA;
      if (row >= p.Cout) continue;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int rb = wn * WN + b * 32 + li;
        const int col = k0 + 4 * (rb % QB) + rb / QB;
        if (col < p.Ktot) out[(size_t)row * p.Ktot + col] = acc[a][b][r] * out_scale;
      }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_one(const WGradArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)3 * (BM + BN) * kRowBytes;
  if (a.planes == 1) {
    hipLaunchKernelGGL((conv_wgrad_x3_kernel<BM, BN, WAVES_M, WAVES_N, 1>), dim3(a.tiles_co * a.tiles_k * a.splitk),
                       dim3(256), lds, stream, a);
  } else if (a.planes == 2 && (a.x_packed || a.dy_packed)) {
    hipLaunchKernelGGL((conv_wgrad_x3_kernel<BM, BN, WAVES_M, WAVES_N, 4>), dim3(a.tiles_co * a.tiles_k * a.splitk),
                       dim3(256), lds, stream, a);
  } else if (a.planes == 2) {
    hipLaunchKernelGGL((conv_wgrad_x3_kernel<BM, BN, WAVES_M, WAVES_N, 2>), dim3(a.tiles_co * a.tiles_k * a.splitk),
                       dim3(256), lds, stream, a);
  } else {
    hipLaunchKernelGGL((conv_wgrad_x3_kernel<BM, BN, WAVES_M, WAVES_N, 3>), dim3(a.tiles_co * a.tiles_k * a.splitk),
                       dim3(256), lds, stream, a);
  }
  return check_launch("conv_wgrad_x3");
}

int launch_wgrad_x3(const WGradArgs& a, const WGradPlan& pl, hipStream_t stream) {
  if (pl.ws) return launch_wgrad_x3ws(a, stream);
  if (pl.bm == 128 && pl.bn == 128) return launch_one<128, 128, 2, 2>(a, stream);
  if (pl.bm == 64 && pl.bn == 128) return launch_one<64, 128, 2, 2>(a, stream);
  if (pl.bm == 128 && pl.bn == 64) return launch_one<128, 64, 2, 2>(a, stream);
  return launch_one<64, 64, 2, 2>(a, stream);
}

}  // namespace evk
