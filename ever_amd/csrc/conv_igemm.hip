// Implicit-GEMM convolution (forward and data-gradient) for gfx950 on v_mfma_f32_32x32x2_f32.
//
//   dst[m][co] = sum_k  gather(src)[m][k] * wgt[co][k]      m = (n, py, px),  k = (ky, kx, ci)
//
// Replaces aten::convolution / aten::convolution_backward(input) issued by nn.Conv2d at
// reference ever/module/_resnets.py:21-29,149 ; fpn.py:23-37,72-73,165,179 ; fs_relation.py:23-53.
//
// Layout: src NHWC fp32, wgt [Cd][Ktot] (K contiguous, = OHWI), dst NHWC fp32.
// One workgroup = 256 threads = 4 waves computes a BM x BN tile; K advances in steps of 32 floats.
// A (im2col rows) and B (weight rows) are staged global -> VGPR -> LDS as 16-byte chunks, double
// buffered, with the next step's global loads in flight under the current step's MFMAs.  The LDS
// image is [row][8 chunks] with chunk ^= (row>>1)&7 so the ds_read_b128 fragment reads of a
// 16-lane group hit 16 distinct 16-byte slots of the 256-byte bank row.
// Fragment mapping (guide §3): A operand lane l = A[i=l&31][k=l>>5], B lane l = B[k=l>>5][j=l&31];
// one ds_read_b128 per lane supplies k = 8g+4h .. 8g+4h+3, i.e. four consecutive MFMAs.
// C/D: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
//
// The gather is branch-free: every 16-byte load is issued unconditionally from a clamped (always
// valid) address and zeroed by a select, so hipcc emits the step's 8 global_load_dwordx4 back to
// back instead of one exec-masked branch + wait per load; all indices are 32-bit element offsets
// (tensors are < 2^31 elements, checked on the host).
#include "common.hpp"
#include <stdlib.h>

namespace evk {

struct IGemmArgs {
  const float* src;
  const float* wgt;
  const float* bias;
  float* dst;
  int N, Hs, Ws, Cs;   // gathered tensor
  int Hm, Wm;          // GEMM-row grid
  int Cd;              // GEMM N
  int kh, kw, cpt;     // cpt = Cs/4 (16-byte chunks per tap)
  int sh, sw, ph, pw, dh, dw;
  int M, Ktot;
  int Hd, Wd, dsh, dsw;  // destination row mapping: (n, py*dsh, px*dsw) in an [N,Hd,Wd,Cd] tensor
  int dense_dst;         // 1 => dst row offset = m*Cd
  int relu;
  int tiles_m, tiles_n;
};

constexpr int BK = 32;
constexpr int kInvalidRow = -(1 << 28);  // y0 of a row past M: every tap fails the bounds test

// MODE 0: forward gather       sy = py*sh - ph + ky*dh
// MODE 1: transposed, stride 1 sy = py + ph - ky*dh
// MODE 2: transposed, strided  t = py + ph - ky*dh ; sy = t/sh iff t >= 0 and t % sh == 0
template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE, int NBUF>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const IGemmArgs p) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MB = WM / 32, NB = WN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;  // rows per thread for the A / B staging passes
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                     // [NBUF][BM][32]
  float* Bs = smem + NBUF * BM * BK;    // [NBUF][BN][32]

  // XCD-aware tile order: consecutive tile ids (sharing A rows / weights) stay on one XCD's L2.
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int tile_n = bid % p.tiles_n;
  const int tile_m = bid / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int c8 = tid & 7;   // chunk column inside the K step
  const int rb = tid >> 3;  // base row 0..31

  // ---- per-thread gather state for its A rows
  int a_y0[AR], a_x0[AR], a_base[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    const int m = m0 + rb + 32 * j;
    if (m < p.M) {
      const int hw = p.Hm * p.Wm;
      const int n = m / hw;
      const int rem = m - n * hw;
      const int py = rem / p.Wm;
      const int px = rem - py * p.Wm;
      if (MODE == 0) {
        a_y0[j] = py * p.sh - p.ph;
        a_x0[j] = px * p.sw - p.pw;
      } else {
        a_y0[j] = py + p.ph;
        a_x0[j] = px + p.pw;
      }
      if (MODE == 2) a_base[j] = n * p.Hs;                                      // image row base
      else a_base[j] = ((n * p.Hs + a_y0[j]) * p.Ws + a_x0[j]) * p.Cs;          // element offset of tap (0,0)
    } else {
      a_y0[j] = kInvalidRow;
      a_x0[j] = 0;
      a_base[j] = 0;
    }
  }
  // B rows: element offset of this thread's chunk in step 0, or -1 past Cd
  int b_off[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int co = n0 + rb + 32 * j;
    b_off[j] = (co < p.Cd) ? co * p.Ktot + c8 * 4 : -1;
  }

  // K-chunk cursor of this thread: chunk q = kt*8 + c8  ->  (ky, kx, cc)
  int cc, kx, ky;
  {
    const int tap = c8 / p.cpt;
    cc = c8 - tap * p.cpt;
    ky = tap / p.kw;
    kx = tap - ky * p.kw;
  }

  f32x4 ra[AR], rbv[BR];
  uint32_t okmask = 0;  // bit j: A row j valid; bit 16+j: B row j valid (for the chunk held in ra / rbv)

  auto load_tiles = [&](int kt) {
    okmask = 0;
    const bool kvalid = ky < p.kh;
    const int oy = ky * p.dh, ox = kx * p.dw;
    const int tapoff = (oy * p.Ws + ox) * p.Cs;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      bool ok = kvalid;
      int idx;
      if (MODE == 0) {
        const int sy = a_y0[j] + oy, sx = a_x0[j] + ox;
        ok = ok && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
        idx = a_base[j] + tapoff + cc * 4;
      } else if (MODE == 1) {
        const int sy = a_y0[j] - oy, sx = a_x0[j] - ox;
        ok = ok && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
        idx = a_base[j] - tapoff + cc * 4;
      } else {
        const int ty = a_y0[j] - oy, tx = a_x0[j] - ox;
        int sy, sx;
        ok = ok && ty >= 0 && tx >= 0;
        if (p.sh == 2) { ok = ok && !(ty & 1); sy = ty >> 1; }
        else { sy = ty / p.sh; ok = ok && (sy * p.sh == ty); }
        if (p.sw == 2) { ok = ok && !(tx & 1); sx = tx >> 1; }
        else { sx = tx / p.sw; ok = ok && (sx * p.sw == tx); }
        ok = ok && sy < p.Hs && sx < p.Ws;
        idx = ((a_base[j] + sy) * p.Ws + sx) * p.Cs + cc * 4;
      }
      // unconditional load from a clamped (valid) offset; the zero-fill select is applied when the
      // chunk is written to LDS, AFTER the MFMA block, so the prefetch stays in flight under it
      okmask |= ok ? (1u << j) : 0u;
      ra[j] = *reinterpret_cast<const f32x4*>(p.src + (ok ? idx : 0));
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      const bool ok = kvalid && b_off[j] >= 0;
      okmask |= ok ? (1u << (16 + j)) : 0u;
      rbv[j] = *reinterpret_cast<const f32x4*>(p.wgt + (ok ? b_off[j] + kt * BK : 0));
    }
    // advance the cursor by 8 chunks (wave-uniform branch on the channel count)
    if (p.cpt >= 8) {
      cc += 8;
      const bool wrap = cc >= p.cpt;
      cc = wrap ? cc - p.cpt : cc;
      kx += wrap ? 1 : 0;
      const bool wrapx = kx == p.kw;
      kx = wrapx ? 0 : kx;
      ky += wrapx ? 1 : 0;
    } else {  // narrow inputs (4-band stem): a step spans several taps
      const int q = (kt + 1) * 8 + c8;
      const int tap = q / p.cpt;
      cc = q - tap * p.cpt;
      ky = tap / p.kw;
      kx = tap - ky * p.kw;
    }
  };

  auto store_tiles = [&](int buf) {
    float* Ab = As + buf * BM * BK;
    float* Bb = Bs + buf * BN * BK;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      const int row = rb + 32 * j;
      const int pc = c8 ^ ((row >> 1) & 7);
      const bool ok = (okmask >> j) & 1u;
      f32x4 v = ra[j];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      *reinterpret_cast<f32x4*>(Ab + row * BK + pc * 4) = v;
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      const int row = rb + 32 * j;
      const int pc = c8 ^ ((row >> 1) & 7);
      const bool ok = (okmask >> (16 + j)) & 1u;
      f32x4 v = rbv[j];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      *reinterpret_cast<f32x4*>(Bb + row * BK + pc * 4) = v;
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

  const int nk = (p.Ktot + BK - 1) / BK;

  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = (NBUF == 2) ? (kt & 1) : 0;
    if (kt + 1 < nk) load_tiles(kt + 1);

    const float* Ab = As + buf * BM * BK + (wm * WM + li) * BK;
    const float* Bb = Bs + buf * BN * BK + (wn * WN + li) * BK;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 fa[MB], fb[NB];
#pragma unroll
      for (int a = 0; a < MB; ++a) {
        const int row = wm * WM + a * 32 + li;
        const int pc = (2 * g + lh) ^ ((row >> 1) & 7);
        fa[a] = *reinterpret_cast<const f32x4*>(Ab + a * 32 * BK + pc * 4);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int row = wn * WN + b * 32 + li;
        const int pc = (2 * g + lh) ^ ((row >> 1) & 7);
        fb[b] = *reinterpret_cast<const f32x4*>(Bb + b * 32 * BK + pc * 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
    }

    if (NBUF == 1) __syncthreads();  // single buffer: everyone is done reading before it is overwritten
    if (kt + 1 < nk) store_tiles((NBUF == 2) ? (buf ^ 1) : 0);
    __syncthreads();
  }

  // ---- epilogue
#pragma unroll
  for (int a = 0; a < MB; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= p.M) continue;
      size_t roff;
      if (p.dense_dst) {
        roff = (size_t)row * p.Cd;
      } else {
        const int hw = p.Hm * p.Wm;
        const int n = row / hw;
        const int rem = row - n * hw;
        const int py = rem / p.Wm;
        const int px = rem - py * p.Wm;
        roff = (((size_t)n * p.Hd + (size_t)py * p.dsh) * p.Wd + (size_t)px * p.dsw) * p.Cd;
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int col = n0 + wn * WN + b * 32 + li;
        if (col < p.Cd) {
          float v = acc[a][b][r];
          if (p.bias) v += p.bias[col];
          if (p.relu) v = fmaxf(v, 0.f);
          p.dst[roff + col] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE, int NBUF = 2>
static int launch_cfg(IGemmArgs& a, hipStream_t stream) {
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.Cd, BN);
  const size_t lds = (size_t)NBUF * (BM + BN) * BK * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, MODE, NBUF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const long long nwg = (long long)a.tiles_m * a.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffLL) {
    set_error("conv_igemm: bad grid %lld", nwg);
    return EVK_E_INVALID;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, MODE, NBUF>), dim3((unsigned)nwg), dim3(256), lds,
                     stream, a);
  return check_launch("conv_igemm");
}

template <int MODE>
static int launch_mode(IGemmArgs& a, hipStream_t stream) {
  // Tile choice: N tile 64 for narrow outputs, else 128; M tile as large as keeps >= 2 workgroups
  // per CU (256 CUs) in flight.
  const int bn = (a.Cd <= 64) ? 64 : 128;
  const long long tn = ceil_div(a.Cd, bn);
  auto tiles = [&](int bm) { return (long long)ceil_div(a.M, bm) * tn; };
  if (bn == 64) {
    if (tiles(256) >= 512) return launch_cfg<256, 64, 4, 1, MODE>(a, stream);
    if (tiles(128) >= 512) return launch_cfg<128, 64, 2, 2, MODE>(a, stream);
    return launch_cfg<64, 64, 2, 2, MODE>(a, stream);
  }
  static const int exp_cfg = getenv("EVK_IGEMM_CFG") ? atoi(getenv("EVK_IGEMM_CFG")) : 0;
  if (tiles(128) >= 512) {
    if (exp_cfg == 1) return launch_cfg<128, 128, 2, 2, MODE, 1>(a, stream);
    if (exp_cfg == 2) return launch_cfg<128, 64, 2, 2, MODE, 2>(a, stream);
    if (exp_cfg == 3) return launch_cfg<256, 64, 4, 1, MODE, 2>(a, stream);
    if (exp_cfg == 4) return launch_cfg<256, 64, 4, 1, MODE, 1>(a, stream);
    return launch_cfg<128, 128, 2, 2, MODE>(a, stream);
  }
  return launch_cfg<64, 128, 2, 2, MODE>(a, stream);
}

int launch_igemm(IGemmArgs& a, int mode, hipStream_t stream) {
  const long long src_elems = (long long)a.N * a.Hs * a.Ws * a.Cs;
  const long long wgt_elems = (long long)a.Cd * a.Ktot;
  if (src_elems >= 0x7fffffffLL || wgt_elems >= 0x7fffffffLL) {
    set_error("conv_igemm: tensors of 2^31 or more elements are not supported (%lld / %lld)", src_elems, wgt_elems);
    return EVK_E_UNSUPPORTED;
  }
  if (mode == 0) return launch_mode<0>(a, stream);
  if (mode == 1) return launch_mode<1>(a, stream);
  return launch_mode<2>(a, stream);
}

static int check_desc(const evk_conv_desc* d) {
  EVK_REQUIRE(d, EVK_E_INVALID, "conv: null descriptor");
  EVK_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->kh > 0 && d->kw > 0,
              EVK_E_INVALID, "conv: non-positive dimension");
  EVK_REQUIRE(d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0 && d->pad_h >= 0 &&
                  d->pad_w >= 0,
              EVK_E_INVALID, "conv: bad stride/dilation/padding");
  EVK_REQUIRE(d->Cin % 4 == 0, EVK_E_UNSUPPORTED,
              "conv: Cin=%d must be a multiple of 4 (use evk_pad_channels)", d->Cin);
  const int ho = (d->H + 2 * d->pad_h - d->dil_h * (d->kh - 1) - 1) / d->stride_h + 1;
  const int wo = (d->W + 2 * d->pad_w - d->dil_w * (d->kw - 1) - 1) / d->stride_w + 1;
  EVK_REQUIRE(ho == d->Ho && wo == d->Wo, EVK_E_INVALID, "conv: Ho/Wo (%d,%d) inconsistent, expect (%d,%d)",
              d->Ho, d->Wo, ho, wo);
  EVK_REQUIRE((long long)d->N * d->Ho * d->Wo < 0x7fffffffLL && (long long)d->N * d->H * d->W < 0x7fffffffLL,
              EVK_E_UNSUPPORTED, "conv: more than 2^31 pixels");
  return EVK_OK;
}

}  // namespace evk

using namespace evk;

extern "C" int evk_conv2d_fwd(const evk_conv_desc* d, const float* x, const float* w, const float* bias,
                              float* y, uint32_t flags, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  EVK_REQUIRE(x && w && y, EVK_E_INVALID, "conv2d_fwd: null pointer");
  IGemmArgs a{};
  a.src = x; a.wgt = w; a.bias = bias; a.dst = y;
  a.N = d->N; a.Hs = d->H; a.Ws = d->W; a.Cs = d->Cin;
  a.Hm = d->Ho; a.Wm = d->Wo; a.Cd = d->Cout;
  a.kh = d->kh; a.kw = d->kw; a.cpt = d->Cin / 4;
  a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
  a.M = d->N * d->Ho * d->Wo;
  a.Ktot = d->kh * d->kw * d->Cin;
  a.Hd = d->Ho; a.Wd = d->Wo; a.dsh = 1; a.dsw = 1; a.dense_dst = 1;
  a.relu = (flags & EVK_CONV_RELU) ? 1 : 0;
  return launch_igemm(a, 0, (hipStream_t)stream);
}

extern "C" int evk_conv2d_dgrad(const evk_conv_desc* d, const float* dy, const float* wt, float* dx,
                                void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  EVK_REQUIRE(dy && wt && dx, EVK_E_INVALID, "conv2d_dgrad: null pointer");
  EVK_REQUIRE(d->Cout % 4 == 0, EVK_E_UNSUPPORTED, "conv2d_dgrad: Cout=%d must be a multiple of 4", d->Cout);
  hipStream_t st = (hipStream_t)stream;
  IGemmArgs a{};
  a.src = dy; a.wgt = wt; a.bias = nullptr; a.dst = dx;
  a.N = d->N; a.Hs = d->Ho; a.Ws = d->Wo; a.Cs = d->Cout;
  a.Cd = d->Cin;
  a.kh = d->kh; a.kw = d->kw; a.cpt = d->Cout / 4;
  a.Ktot = d->kh * d->kw * d->Cout;
  a.relu = 0;
  a.Hd = d->H; a.Wd = d->W;
  if (d->kh == 1 && d->kw == 1 && d->pad_h == 0 && d->pad_w == 0 && (d->stride_h > 1 || d->stride_w > 1)) {
    // 1x1 strided: only pixels (oy*s, ox*s) receive gradient.  GEMM over the output grid and a
    // scattered store into the zero-filled dx (no wasted MFMAs on the 3/4 empty rows).
    hipError_t e = hipMemsetAsync(dx, 0, (size_t)d->N * d->H * d->W * d->Cin * sizeof(float), st);
    if (e != hipSuccess) { set_error("conv2d_dgrad memset: %s", hipGetErrorString(e)); return EVK_E_LAUNCH; }
    a.Hm = d->Ho; a.Wm = d->Wo;
    a.sh = 1; a.sw = 1; a.ph = 0; a.pw = 0; a.dh = 1; a.dw = 1;
    a.M = d->N * d->Ho * d->Wo;
    a.dsh = d->stride_h; a.dsw = d->stride_w; a.dense_dst = 0;
    return launch_igemm(a, 0, st);
  }
  a.Hm = d->H; a.Wm = d->W;
  a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
  a.M = d->N * d->H * d->W;
  a.dsh = 1; a.dsw = 1; a.dense_dst = 1;
  return launch_igemm(a, (d->stride_h == 1 && d->stride_w == 1) ? 1 : 2, st);
}

// wt[ci][ky][kx][co] = w[co][ky][kx][ci]   (taps are NOT flipped: the transposed gather of
// conv_igemm walks ty = py + pad - ky*dil, which already pairs tap ky with its source row).
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout,
                                         int taps, int Cin) {
  const size_t total = (size_t)Cout * taps * Cin;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // i indexes wt: ((ci*taps + t)*Cout + co)
    const int co = (int)(i % Cout);
    const size_t r = i / Cout;
    const int t = (int)(r % taps);
    const int ci = (int)(r / taps);
    wt[i] = w[((size_t)co * taps + t) * Cin + ci];
  }
}

extern "C" int evk_conv2d_pack_dgrad_weight(const evk_conv_desc* d, const float* w, float* wt, void* stream) {
  EVK_REQUIRE(d && w && wt, EVK_E_INVALID, "pack_dgrad_weight: null pointer");
  const size_t total = (size_t)d->Cout * d->kh * d->kw * d->Cin;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_dgrad_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wt, d->Cout,
                     d->kh * d->kw, d->Cin);
  return check_launch("pack_dgrad_weight");
}
