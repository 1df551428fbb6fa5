// Implicit-GEMM convolution (forward and data-gradient) for gfx950 on v_mfma_f32_32x32x2_f32.
//
//   dst[m][co] = sum_k  gather(src)[m][k] * wgt[co][k]      m = (n, gy, gx),  k = (jy, jx, ci)
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
// C/D: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); the MFMA is issued as W_tile x X_tile^T so
// that rows = output channels (4 consecutive per lane and register quad => 16-byte stores).
//
// ONE gather form serves every use: source pixel of GEMM row (n, gy, gx) and tap (jy, jx) is
//     sy = gy*ash + oy0 + jy*oys ,   sx = gx*asw + ox0 + jx*oxs            (affine in the tap index)
//   forward            ash = stride, oy0 = -pad,  oys = dilation
//   dgrad, stride 1    ash = 1,      oy0 = +pad,  oys = -dilation          (taps are not flipped: the
//                                                                           sign of oys does it)
//   dgrad, stride s    one launch per residue class (cy, cx) of the input pixel mod s: only the taps
//                      with (cy + pad - ky*dil) % s == 0 reach that class, they are ky = ky0 + j*kstep
//                      and their source row is gy + (cy+pad-ky0*dil)/s - j*dil/g : again affine.  No
//                      MFMA is spent on the (s*s-1)/(s*s) structurally-zero products.
// The gather is branch-free: every 16-byte load is issued from a clamped (valid) 32-bit element
// offset and the zero fill is applied when the chunk is written to LDS, after the MFMA block, so the
// prefetch stays in flight under the MFMAs.
#include "igemm_common.hpp"

namespace evk {

constexpr int BK = 32;
constexpr int kInvalidRow = -(1 << 28);  // y0 of a row past M: every tap fails the bounds test

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const IGemmArgs p) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MB = WM / 32, NB = WN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;  // rows per thread for the A / B staging passes
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [2][BM][32]
  float* Bs = smem + 2 * BM * BK;    // [2][BN][32]

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
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
      const int gy = rem / p.Wm;
      const int gx = rem - gy * p.Wm;
      a_y0[j] = gy * p.ash + p.oy0;
      a_x0[j] = gx * p.asw + p.ox0;
      a_base[j] = ((n * p.Hs + a_y0[j]) * p.Ws + a_x0[j]) * p.Cs;  // element offset of tap (0,0)
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
    const int oy = ky * p.oys, ox = kx * p.oxs;
    const int tapoff = (oy * p.Ws + ox) * p.Cs + cc * 4;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      const int sy = a_y0[j] + oy, sx = a_x0[j] + ox;
      const bool ok = kvalid && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      okmask |= ok ? (1u << j) : 0u;
      ra[j] = *reinterpret_cast<const f32x4*>(p.src + (ok ? a_base[j] + tapoff : 0));
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

  // One K step.  The staging work of the NEXT tile is threaded through this tile's MFMA groups so
  // that it issues in the shadow of the matrix pipe (an in-order wave can issue VALU / VMEM / LDS
  // right behind an MFMA while that MFMA occupies the pipe for 64 cycles):
  //   g0 | address math + 8 global loads (tile kt+1) | g1 | g2 | vmcnt + zero-fill + 8 ds_write | g3 | barrier
  // Writing buffer buf^1 during the step is safe: its last readers passed the previous barrier.
  auto mfma_group = [&](const float* Ab, const float* Bb, int g) {
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
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[b][j], fa[a][j], acc[a][b], 0, 0, 0);
  };

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nk;
    const float* Ab = As + buf * BM * BK + (wm * WM + li) * BK;
    const float* Bb = Bs + buf * BN * BK + (wn * WN + li) * BK;
    mfma_group(Ab, Bb, 0);
    if (more) load_tiles(kt + 1);
    mfma_group(Ab, Bb, 1);
    mfma_group(Ab, Bb, 2);
    if (more) store_tiles(buf ^ 1);
    mfma_group(Ab, Bb, 3);
    __syncthreads();
  }

  igemm_epilogue<MB, NB, WM, WN>(p, acc, m0, n0, wm, wn, li, lh);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_cfg(IGemmArgs& a, hipStream_t stream) {
  a.tiles_m = ceil_div(a.M, BM);
  a.tiles_n = ceil_div(a.Cd, BN);
  const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float);
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const long long nwg = (long long)a.tiles_m * a.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffLL) {
    set_error("conv_igemm: bad grid %lld", nwg);
    return EVK_E_INVALID;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N>), dim3((unsigned)nwg), dim3(256), lds, stream, a);
  return check_launch("conv_igemm");
}

// 1x1 convolution on a handful of rows (the FS-Relation scene MLP on 1x1 maps: M = batch, K up to
// 2048; reference fs_relation.py:23-29).  An MFMA tile would be >90 % padding and latency bound on a
// 64-step K loop in 2-4 workgroups; here one workgroup owns one output column and all rows, streams
// its weight row once with 16-byte loads and reduces across the block.
constexpr int kSmallM = 32;
__global__ __launch_bounds__(256) void conv1x1_smallm_kernel(const float* __restrict__ src,
                                                             const float* __restrict__ wgt,
                                                             const float* __restrict__ bias, float* __restrict__ dst,
                                                             int M, int K, int Cd, int relu) {
  __shared__ float red[4][kSmallM];
  const int n = blockIdx.x;
  const int k4 = K >> 2;
  float acc[kSmallM];
#pragma unroll
  for (int m = 0; m < kSmallM; ++m) acc[m] = 0.f;
  for (int kc = threadIdx.x; kc < k4; kc += 256) {
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wgt + (size_t)n * K + kc * 4);
#pragma unroll
    for (int m = 0; m < kSmallM; ++m)
      if (m < M) {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(src + (size_t)m * K + kc * 4);
        acc[m] += w4.x * x4.x + w4.y * x4.y + w4.z * x4.z + w4.w * x4.w;
      }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int m = 0; m < kSmallM; ++m) {
    const float s = wave_sum(acc[m]);
    if (lane == 0) red[wave][m] = s;
  }
  __syncthreads();
  if (threadIdx.x < M) {
    const int m = threadIdx.x;
    float v = (red[0][m] + red[1][m]) + (red[2][m] + red[3][m]);
    if (bias) v += bias[n];
    if (relu) v = fmaxf(v, 0.f);
    dst[(size_t)m * Cd + n] = v;
  }
}

int launch_igemm(IGemmArgs& a, hipStream_t stream) {
  const long long src_elems = (long long)a.N * a.Hs * a.Ws * a.Cs;
  const long long wgt_elems = (long long)a.Cd * a.Ktot;
  if (src_elems >= 0x7fffffffLL || wgt_elems >= 0x7fffffffLL) {
    set_error("conv_igemm: tensors of 2^31 or more elements are not supported (%lld / %lld)", src_elems, wgt_elems);
    return EVK_E_UNSUPPORTED;
  }
  if (a.kh == 1 && a.kw == 1 && a.M <= kSmallM && a.dense_dst && !a.accum && a.ash == 1 && a.asw == 1 && a.oy0 == 0 &&
      a.ox0 == 0 && a.Hm == a.Hs && a.Wm == a.Ws) {
    hipLaunchKernelGGL(conv1x1_smallm_kernel, dim3(a.Cd), dim3(256), 0, stream, a.src, a.wgt, a.bias, a.dst, a.M,
                       a.Ktot, a.Cd, a.relu);
    return check_launch("conv1x1_smallm");
  }
  // Tile choice: N tile 64 for narrow outputs, else 128; M tile as large as keeps >= 2 workgroups
  // per CU (256 CUs) in flight.
  const int bn = (a.Cd <= 64) ? 64 : 128;
  const long long tn = ceil_div(a.Cd, bn);
  auto tiles = [&](int bm) { return (long long)ceil_div(a.M, bm) * tn; };
  if (bn == 64) {
    if (tiles(256) >= 512) return launch_cfg<256, 64, 4, 1>(a, stream);
    if (tiles(128) >= 512) return launch_cfg<128, 64, 2, 2>(a, stream);
    return launch_cfg<64, 64, 2, 2>(a, stream);
  }
  if (tiles(128) >= 512) return launch_cfg<128, 128, 2, 2>(a, stream);
  return launch_cfg<64, 128, 2, 2>(a, stream);
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

static int gcd_i(int a, int b) { return b == 0 ? a : gcd_i(b, a % b); }
AxisPlan plan_axis(int c, int pad, int dil, int stride, int ksize) {
  AxisPlan ap{0, 1, 0, 0, 0};
  const int g = gcd_i(dil, stride);
  if ((c + pad) % g != 0) return ap;  // no tap reaches this class
  ap.kstep = stride / g;
  int k0 = -1;
  for (int k = 0; k < ap.kstep && k < ksize; ++k)
    if ((c + pad - k * dil) % stride == 0) { k0 = k; break; }
  if (k0 < 0) return ap;
  ap.k0 = k0;
  ap.nt = (ksize - 1 - k0) / ap.kstep + 1;
  ap.o0 = (c + pad - k0 * dil) / stride;  // exact; may be negative
  ap.ostep = -(dil / g);
  return ap;
}

}  // namespace evk

using namespace evk;

static inline int kpad32(int k) { return (k + 31) & ~31; }

// w3 != nullptr selects the bf16-split kernel (conv_igemm_x3.hip) on pre-split weight planes
static int conv_fwd_any(const evk_conv_desc* d, const float* x, const float* w, const uint16_t* w3, const float* bias,
                        float* y, uint32_t flags, void* stream, const float* residual = nullptr,
                        float* bn_parts = nullptr, int32_t bn_cap = 0, int32_t* nparts = nullptr, int planes = 3,
                        const uint32_t* a_scale = nullptr, const uint32_t* w_scale = nullptr, uint32_t* out_amax = nullptr) {
  int rc = check_desc(d);
  if (rc) return rc;
  EVK_REQUIRE(x && (w || w3) && y, EVK_E_INVALID, "conv2d_fwd: null pointer");
  IGemmArgs a{};
  EVK_REQUIRE(residual != y, EVK_E_INVALID, "conv2d_fwd: residual must not alias y");
  a.src = x; a.wgt = w; a.wgt3 = w3; a.bias = bias; a.accum = residual; a.dst = y;
  a.N = d->N; a.Hs = d->H; a.Ws = d->W; a.Cs = d->Cin;
  a.Hm = d->Ho; a.Wm = d->Wo; a.Cd = d->Cout;
  a.kh = d->kh; a.kw = d->kw; a.cpt = d->Cin / 4;
  a.ash = d->stride_h; a.asw = d->stride_w;
  a.oy0 = -d->pad_h; a.oys = d->dil_h; a.ox0 = -d->pad_w; a.oxs = d->dil_w;
  a.M = d->N * d->Ho * d->Wo;
  a.Ktot = d->kh * d->kw * d->Cin;
  a.Hd = d->Ho; a.Wd = d->Wo; a.dsh = 1; a.dsw = 1; a.dense_dst = 1;
  a.relu = (flags & EVK_CONV_RELU) ? 1 : 0;
  a.Kpad = kpad32(a.Ktot);
  a.planes = planes;
  a.a_scale = a_scale; a.w_scale = w_scale;
  a.a_packed = (planes == 2 && (flags & EVK_CONV_X_PACKED)) ? 1 : 0;
  a.out_amax = out_amax;
  a.bn_want = (bn_parts && w3 && d->Cout % 4 == 0) ? 1 : 0;
  a.bn_buf = bn_parts;
  a.bn_cap = bn_cap;
  if (nparts) *nparts = 0;
  if (w3) {
    int hr = launch_conv3x3_wino(a, (hipStream_t)stream);         // 3x3 'same', f16x2, large maps: Winograd F(2,3) along x
    if (hr == 1) hr = launch_conv3x3_halo(a, (hipStream_t)stream);   // 3x3 'same' convolutions: LDS-halo kernel
    if (hr != 1) {
      if (nparts && hr == EVK_OK) *nparts = a.bn_parts;
      return hr;
    }
  }
  const int rc2 = w3 ? launch_igemm_x3(a, (hipStream_t)stream) : launch_igemm(a, (hipStream_t)stream);
  if (nparts && rc2 == EVK_OK) *nparts = a.bn_parts;
  return rc2;
}

extern "C" int evk_conv2d_fwd(const evk_conv_desc* d, const float* x, const float* w, const float* bias,
                              float* y, uint32_t flags, void* stream) {
  return conv_fwd_any(d, x, w, nullptr, bias, y, flags, stream);
}

extern "C" int evk_conv2d_fwd_x3(const evk_conv_desc* d, const float* x, const void* wsplit, const float* bias,
                                 float* y, uint32_t flags, void* stream) {
  EVK_REQUIRE(wsplit, EVK_E_INVALID, "conv2d_fwd_x3: null weight planes");
  return conv_fwd_any(d, x, nullptr, reinterpret_cast<const uint16_t*>(wsplit), bias, y, flags, stream);
}

extern "C" int32_t evk_conv2d_stats_max_parts(const evk_conv_desc* d) {
  if (!d) return 0;
  // one record per tile of at least 64 rows
  return (int32_t)(((int64_t)d->N * d->Ho * d->Wo + 63) / 64 + 1);
}
extern "C" int evk_conv2d_fwd_x3_stats(const evk_conv_desc* d, const float* x, const void* wsplit, const float* bias,
                                       float* y, uint32_t flags, float* bn_parts, int32_t bn_capacity, int32_t* nparts,
                                       void* stream) {
  EVK_REQUIRE(wsplit && bn_parts && nparts, EVK_E_INVALID, "conv2d_fwd_x3_stats: null pointer");
  return conv_fwd_any(d, x, nullptr, reinterpret_cast<const uint16_t*>(wsplit), bias, y, flags, stream, nullptr, bn_parts,
                      bn_capacity, nparts);
}

// Plain bf16 operands (ONE product per operand pair, fp32 accumulate): the counterpart of the reference's
// `--mixed_precision bf16`.  Same arguments and the same weight-plane buffers as the x3 forms (only plane 0 is read).
extern "C" int evk_conv2d_fwd_bf16(const evk_conv_desc* d, const float* x, const void* wsplit, const float* bias, float* y,
                                   uint32_t flags, float* bn_parts, int32_t bn_capacity, int32_t* nparts, void* stream) {
  EVK_REQUIRE(wsplit, EVK_E_INVALID, "conv2d_fwd_bf16: null weight planes");
  return conv_fwd_any(d, x, nullptr, reinterpret_cast<const uint16_t*>(wsplit), bias, y, flags, stream, nullptr, bn_parts,
                      bn_capacity, nparts, 1);
}

// 2-term fp16 split of operands scaled by a power of two (x3_common.hpp: NP = 2): three MFMA products per operand
// pair, 22-bit operands, fp32 accumulate — fp32-grade at half the matrix work of the x3 forms.  x_absmax / w_absmax:
// device words holding the bit image of max|x| (evk_absmax) and of max|w| (what the planes were produced with:
// evk_conv2d_split_weight_f16x2 / evk_conv2d_split_multi_f16x2).  residual, bn_parts, nparts may be null.
extern "C" int evk_conv2d_fwd_f16x2(const evk_conv_desc* d, const float* x, const uint32_t* x_absmax, const void* wsplit,
                                    const uint32_t* w_absmax, const float* bias, const float* residual, float* y,
                                    uint32_t flags, float* bn_parts, int32_t bn_capacity, int32_t* nparts,
                                    uint32_t* y_absmax, void* stream) {
  EVK_REQUIRE(wsplit && x_absmax && w_absmax, EVK_E_INVALID, "conv2d_fwd_f16x2: null weight planes / scales");
  return conv_fwd_any(d, x, nullptr, reinterpret_cast<const uint16_t*>(wsplit), bias, y, flags, stream, residual, bn_parts,
                      bn_capacity, nparts, 2, x_absmax, w_absmax, y_absmax);
}

// y = act(conv(x, w) + bias + residual): the inference form of a ResNet block's last convolution once its
// BatchNorm is folded into (w, bias) — reference _resnets.py:95-112 (`out += identity; relu`)
extern "C" int evk_conv2d_fwd_res(const evk_conv_desc* d, const float* x, const float* w, const float* bias,
                                  const float* residual, float* y, uint32_t flags, void* stream) {
  return conv_fwd_any(d, x, w, nullptr, bias, y, flags, stream, residual);
}
extern "C" int evk_conv2d_fwd_x3_res(const evk_conv_desc* d, const float* x, const void* wsplit, const float* bias,
                                     const float* residual, float* y, uint32_t flags, void* stream) {
  EVK_REQUIRE(wsplit, EVK_E_INVALID, "conv2d_fwd_x3_res: null weight planes");
  return conv_fwd_any(d, x, nullptr, reinterpret_cast<const uint16_t*>(wsplit), bias, y, flags, stream, residual);
}

static int conv_dgrad_any(const evk_conv_desc* d, const float* dy, const float* wt, const uint16_t* wt3,
                          const float* accum, float* dx, void* stream, int planes = 3,
                          const uint32_t* a_scale = nullptr, const uint32_t* w_scale = nullptr, uint32_t* out_amax = nullptr,
                          int dy_packed = 0, const uint32_t* accum_bits = nullptr) {
  int rc = check_desc(d);
  if (rc) return rc;
  EVK_REQUIRE(dy && (wt || wt3) && dx, EVK_E_INVALID, "conv2d_dgrad: null pointer");
  EVK_REQUIRE(d->Cout % 4 == 0, EVK_E_UNSUPPORTED, "conv2d_dgrad: Cout=%d must be a multiple of 4", d->Cout);
  hipStream_t st = (hipStream_t)stream;
  const int sh = d->stride_h, sw = d->stride_w;
  // does every residue class receive at least one tap?  if not, those pixels of dx are plain zeros
  bool all_covered = true;
  for (int cy = 0; cy < sh; ++cy) all_covered = all_covered && plan_axis(cy, d->pad_h, d->dil_h, sh, d->kh).nt > 0;
  for (int cx = 0; cx < sw; ++cx) all_covered = all_covered && plan_axis(cx, d->pad_w, d->dil_w, sw, d->kw).nt > 0;
  // accum == dx (split kernels only): accumulate in place — every epilogue thread reads the accum quads of its row block
  // before it stores the same addresses (igemm_store_rows), and pixels no tap reaches already hold their value.  The
  // strided 1x1 shortcut of a residual block uses this: its data gradient touches one pixel in four of a tensor the
  // main branch's data gradient has just written, so the 268 MB fill / copy below disappears.
  EVK_REQUIRE(accum != dx || wt3, EVK_E_INVALID, "conv2d_dgrad: accum must not alias dx in the fp32 kernels");
  if (!all_covered && accum != dx) {  // pixels no tap reaches: dx = 0 (+ accum)
    const size_t bytes = (size_t)d->N * d->H * d->W * d->Cin * sizeof(float);
    hipError_t e = accum ? hipMemcpyAsync(dx, accum, bytes, hipMemcpyDeviceToDevice, st) : hipMemsetAsync(dx, 0, bytes, st);
    if (e != hipSuccess) { set_error("conv2d_dgrad fill: %s", hipGetErrorString(e)); return EVK_E_LAUNCH; }
  }
  size_t woff = 0;   // class weights are packed back to back by evk_conv2d_pack_dgrad_weight
  size_t woff3 = 0;  // ... and the class planes by evk_conv2d_split_weight(for_dgrad = 1)
  for (int cy = 0; cy < sh; ++cy)
    for (int cx = 0; cx < sw; ++cx) {
      const AxisPlan py = plan_axis(cy, d->pad_h, d->dil_h, sh, d->kh);
      const AxisPlan px = plan_axis(cx, d->pad_w, d->dil_w, sw, d->kw);
      const int Hm = d->H > cy ? (d->H - cy + sh - 1) / sh : 0;
      const int Wm = d->W > cx ? (d->W - cx + sw - 1) / sw : 0;
      const size_t wsize = (size_t)d->Cin * py.nt * px.nt * d->Cout;
      if (py.nt > 0 && px.nt > 0 && Hm > 0 && Wm > 0) {
        IGemmArgs a{};
        a.planes = planes;
        a.a_scale = a_scale; a.w_scale = w_scale;
        a.a_packed = dy_packed;
        a.out_amax = out_amax;
        a.src = dy; a.wgt = wt ? wt + woff : nullptr; a.wgt3 = wt3 ? wt3 + woff3 : nullptr;
        a.bias = nullptr; a.accum = accum; a.dst = dx;
        a.accum_bits = accum_bits;
        a.N = d->N; a.Hs = d->Ho; a.Ws = d->Wo; a.Cs = d->Cout;
        a.Hm = Hm; a.Wm = Wm; a.Cd = d->Cin;
        a.kh = py.nt; a.kw = px.nt; a.cpt = d->Cout / 4;
        a.ash = 1; a.asw = 1;
        a.oy0 = py.o0; a.oys = py.ostep; a.ox0 = px.o0; a.oxs = px.ostep;
        a.M = d->N * Hm * Wm;
        a.Ktot = py.nt * px.nt * d->Cout;
        a.Hd = d->H; a.Wd = d->W; a.dsh = sh; a.dsw = sw; a.doy = cy; a.dox = cx;
        a.dense_dst = (sh == 1 && sw == 1) ? 1 : 0;
        a.relu = 0;
        a.Kpad = kpad32(a.Ktot);
        rc = wt3 ? launch_conv3x3_wino(a, st) : 1;
        if (rc == 1) rc = wt3 ? launch_conv3x3_halo(a, st) : 1;
        if (rc == 1) rc = wt3 ? launch_igemm_x3(a, st) : launch_igemm(a, st);
        if (rc) return rc;
      }
      woff += wsize;
      woff3 += (size_t)3 * d->Cin * kpad32(py.nt * px.nt * d->Cout);
    }
  return EVK_OK;
}

extern "C" int evk_conv2d_dgrad(const evk_conv_desc* d, const float* dy, const float* wt, const float* accum,
                                float* dx, void* stream) {
  return conv_dgrad_any(d, dy, wt, nullptr, accum, dx, stream);
}

extern "C" int evk_conv2d_dgrad_bf16(const evk_conv_desc* d, const float* dy, const void* wsplit_t, const float* accum,
                                     float* dx, void* stream) {
  EVK_REQUIRE(wsplit_t, EVK_E_INVALID, "conv2d_dgrad_bf16: null weight planes");
  return conv_dgrad_any(d, dy, nullptr, reinterpret_cast<const uint16_t*>(wsplit_t), accum, dx, stream, 1);
}

extern "C" int evk_conv2d_dgrad_f16x2(const evk_conv_desc* d, const float* dy, const uint32_t* dy_absmax,
                                      const void* wsplit_t, const uint32_t* w_absmax, const float* accum, float* dx,
                                      uint32_t* dx_absmax, void* stream) {
  EVK_REQUIRE(wsplit_t && dy_absmax && w_absmax, EVK_E_INVALID, "conv2d_dgrad_f16x2: null weight planes / scales");
  return conv_dgrad_any(d, dy, nullptr, reinterpret_cast<const uint16_t*>(wsplit_t), accum, dx, stream, 2, dy_absmax,
                        w_absmax, dx_absmax);
}

extern "C" int evk_conv2d_dgrad_f16x2_ex(const evk_conv_desc* d, const void* dy, const uint32_t* dy_absmax,
                                         const void* wsplit_t, const uint32_t* w_absmax, const float* accum, float* dx,
                                         uint32_t* dx_absmax, uint32_t flags, void* stream) {
  EVK_REQUIRE(wsplit_t && dy_absmax && w_absmax, EVK_E_INVALID, "conv2d_dgrad_f16x2_ex: null weight planes / scales");
  EVK_REQUIRE((flags & ~EVK_CONV_DY_PACKED) == 0, EVK_E_INVALID, "conv2d_dgrad_f16x2_ex: unknown flag 0x%x", flags);
  return conv_dgrad_any(d, reinterpret_cast<const float*>(dy), nullptr, reinterpret_cast<const uint16_t*>(wsplit_t), accum, dx,
                        stream, 2, dy_absmax, w_absmax, dx_absmax, (flags & EVK_CONV_DY_PACKED) ? 1 : 0);
}

// accum_bits: ReLU bits of `accum` (evk_bn_fwd_train_parts_bits): dx = dgrad + (accum where its bit is set).  Stride 1 only.
extern "C" int evk_conv2d_dgrad_f16x2_masked(const evk_conv_desc* d, const void* dy, const uint32_t* dy_absmax,
                                             const void* wsplit_t, const uint32_t* w_absmax, const float* accum,
                                             const uint32_t* accum_bits, float* dx, uint32_t* dx_absmax, uint32_t flags,
                                             void* stream) {
  EVK_REQUIRE(wsplit_t && dy_absmax && w_absmax && accum && accum_bits, EVK_E_INVALID, "conv2d_dgrad_f16x2_masked: null pointer");
  EVK_REQUIRE((flags & ~EVK_CONV_DY_PACKED) == 0, EVK_E_INVALID, "conv2d_dgrad_f16x2_masked: unknown flag 0x%x", flags);
  EVK_REQUIRE(d && d->stride_h == 1 && d->stride_w == 1 && d->Cin % 4 == 0 && accum != dx, EVK_E_UNSUPPORTED,
              "conv2d_dgrad_f16x2_masked: stride-1 convolutions with Cin %% 4 == 0, accum distinct from dx");
  return conv_dgrad_any(d, reinterpret_cast<const float*>(dy), nullptr, reinterpret_cast<const uint16_t*>(wsplit_t), accum, dx,
                        stream, 2, dy_absmax, w_absmax, dx_absmax, (flags & EVK_CONV_DY_PACKED) ? 1 : 0, accum_bits);
}

extern "C" int evk_conv2d_dgrad_x3(const evk_conv_desc* d, const float* dy, const void* wsplit_t, const float* accum,
                                   float* dx, void* stream) {
  EVK_REQUIRE(wsplit_t, EVK_E_INVALID, "conv2d_dgrad_x3: null weight planes");
  return conv_dgrad_any(d, dy, nullptr, reinterpret_cast<const uint16_t*>(wsplit_t), accum, dx, stream);
}

// Class-ordered data-gradient weights: for each residue class (cy, cx) in row-major order a block
// wt_c[ci][jy][jx][co] = w[co][ky0 + jy*kstep_y][kx0 + jx*kstep_x][ci].  Every tap belongs to exactly
// one class, so the total size is Cin*kh*kw*Cout; for stride 1 it is the plain [Cin][kh][kw][Cout].
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int kh, int kw,
                                         int Cin, int ky0, int ksy, int nty, int kx0, int ksx, int ntx) {
  const size_t total = (size_t)Cin * nty * ntx * Cout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    size_t r = i / Cout;
    const int jx = (int)(r % ntx);
    r /= ntx;
    const int jy = (int)(r % nty);
    const int ci = (int)(r / nty);
    const int ky = ky0 + jy * ksy, kx = kx0 + jx * ksx;
    wt[i] = w[(((size_t)co * kh + ky) * kw + kx) * Cin + ci];
  }
}

extern "C" int evk_conv2d_pack_dgrad_weight(const evk_conv_desc* d, const float* w, float* wt, void* stream) {
  EVK_REQUIRE(d && w && wt, EVK_E_INVALID, "pack_dgrad_weight: null pointer");
  EVK_REQUIRE(d->stride_h > 0 && d->stride_w > 0 && d->dil_h > 0 && d->dil_w > 0, EVK_E_INVALID,
              "pack_dgrad_weight: bad stride/dilation");
  size_t woff = 0;
  for (int cy = 0; cy < d->stride_h; ++cy)
    for (int cx = 0; cx < d->stride_w; ++cx) {
      const AxisPlan py = plan_axis(cy, d->pad_h, d->dil_h, d->stride_h, d->kh);
      const AxisPlan px = plan_axis(cx, d->pad_w, d->dil_w, d->stride_w, d->kw);
      const size_t total = (size_t)d->Cin * py.nt * px.nt * d->Cout;
      if (total > 0) {
        const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        hipLaunchKernelGGL(pack_dgrad_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wt + woff,
                           d->Cout, d->kh, d->kw, d->Cin, py.k0, py.kstep, py.nt, px.k0, px.kstep, px.nt);
        int rc = check_launch("pack_dgrad_weight");
        if (rc) return rc;
      }
      woff += total;
    }
  return EVK_OK;
}
