// Convolution weight gradient for gfx950 on v_mfma_f32_32x32x2_f32.
//
//   dw[co][k] = sum_m dy[m][co] * im2col(x)[m][k]        m = (n, oy, ox),  k = (ky, kx, ci)
//
// Replaces aten::convolution_backward(weight, bias) of the nn.Conv2d sites listed in
// conv_igemm.hip (reference ever/module/_resnets.py:21-29,149 ; fpn.py ; fs_relation.py).
//
// GEMM view: rows = Cout, cols = kh*kw*Cin, reduction = pixels.  Both operands are stored
// pixel-major in HBM ([pixel][channel]), which is exactly the k-major LDS image the f32 MFMA
// fragments want (lane l reads row k=l>>5, column i=l&31: 32 consecutive words, conflict free),
// so tiles are staged [32 pixels][BM or BN channels] without any transpose.  The pixel range is
// split across workgroups (grid.z); partial tiles go to the caller's workspace and are summed in a
// fixed order by a second kernel => bitwise reproducible, no atomics.
#include "wgrad_common.hpp"
#include <stdlib.h>

namespace evk {

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WGradArgs p) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MB = WM / 32, NB = WN / 32;
  constexpr int ACH = BM / 4, BCH = BN / 4;          // 16-byte chunks per staged row
  constexpr int APASS = BKP * ACH / 256, BPASS = BKP * BCH / 256;
  constexpr int AROWS = 256 / ACH, BROWS = 256 / BCH;  // rows covered per pass
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  static_assert(MB <= 2 && NB <= 2, "fragment vectors are float or float2");
  struct alignas(4 * MB) FragA { float v[MB]; };
  struct alignas(4 * NB) FragB { float v[NB]; };

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                   // [2][32][BM]
  float* Bs = smem + 2 * BKP * BM;    // [2][32][BN]

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
  // A staging: dy rows
  const int a_c = tid % ACH, a_r = tid / ACH;
  const bool a_cvalid = (co0 + a_c * 4) < p.Cout;
  // B staging: im2col rows; this thread's K chunk is fixed for the whole reduction
  const int b_c = tid % BCH, b_r = tid / BCH;
  const int q = (k0 >> 2) + b_c;
  const bool b_cvalid = q * 4 < p.Ktot;
  int b_dy = 0, b_dx = 0, b_cc = 0;
  if (b_cvalid) {
    const int tap = q / p.cpt;
    b_cc = (q - tap * p.cpt) * 4;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    b_dy = ky * p.dh - p.ph;
    b_dx = kx * p.dw - p.pw;
  }

  f32x4 ra[APASS], rb[BPASS];
  uint32_t okmask = 0;  // bit j: A pass j valid; bit 16+j: B pass j valid

  // Branch-free gathers: every load is issued from a clamped (valid) 32-bit element offset; the
  // zero-fill select is applied when the chunk is written to LDS (after the MFMA block), so the
  // prefetch stays in flight under the MFMAs.
  auto load_tiles = [&](int pix0) {
    okmask = 0;
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const int m = pix0 + a_r + j * AROWS;
      const bool ok = a_cvalid && m < pend;
      okmask |= ok ? (1u << j) : 0u;
      ra[j] = *reinterpret_cast<const f32x4*>(p.dy + (ok ? m * p.Cout + co0 + a_c * 4 : 0));
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      const int m = pix0 + b_r + j * BROWS;
      const uint32_t mm = (uint32_t)min(m, p.M - 1);
      const uint32_t n = fdiv(mm, p.fd_hw);
      const uint32_t rem = mm - n * p.fd_hw.div;
      const uint32_t oy = fdiv(rem, p.fd_w);
      const uint32_t ox = rem - oy * p.fd_w.div;
      const int sy = (int)oy * p.sh + b_dy;
      const int sx = (int)ox * p.sw + b_dx;
      const bool ok = b_cvalid && m < pend && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
      okmask |= ok ? (1u << (16 + j)) : 0u;
      rb[j] = *reinterpret_cast<const f32x4*>(p.x + (ok ? (((int)n * p.H + sy) * p.W + sx) * p.Cin + b_cc : 0));
    }
  };
  auto store_tiles = [&](int buf) {
    float* Ab = As + buf * BKP * BM;
    float* Bb = Bs + buf * BKP * BN;
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const bool ok = (okmask >> j) & 1u;
      f32x4 v = ra[j];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      *reinterpret_cast<f32x4*>(Ab + (a_r + j * AROWS) * BM + a_c * 4) = v;
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      const bool ok = (okmask >> (16 + j)) & 1u;
      f32x4 v = rb[j];
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
      *reinterpret_cast<f32x4*>(Bb + (b_r + j * BROWS) * BN + b_c * 4) = v;
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

  const int nk = (pend - pbeg + BKP - 1) / BKP;
  if (nk > 0) {
    load_tiles(pbeg);
    store_tiles(0);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles(pbeg + (kt + 1) * BKP);
    // Fragment reads: the wave's 64 (or 32) tile rows are dealt to (MFMA block a, lane row i) as
    // row = MB*i + a, so ONE ds_read_b64 per operand feeds both blocks of a k-step, and the reads of
    // step s+1 are issued before the MFMAs of step s (register double buffer) to hide LDS latency.
    const float* Ab = As + buf * BKP * BM + wm * WM + MB * li;
    const float* Bb = Bs + buf * BKP * BN + wn * WN + NB * li;
    FragA fa_cur = *reinterpret_cast<const FragA*>(Ab + lh * BM);
    FragB fb_cur = *reinterpret_cast<const FragB*>(Bb + lh * BN);
#pragma unroll
    for (int s = 0; s < BKP / 2; ++s) {
      FragA fa_nxt = fa_cur;
      FragB fb_nxt = fb_cur;
      if (s + 1 < BKP / 2) {
        fa_nxt = *reinterpret_cast<const FragA*>(Ab + (2 * s + 2 + lh) * BM);
        fb_nxt = *reinterpret_cast<const FragB*>(Bb + (2 * s + 2 + lh) * BN);
      }
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_cur.v[a], fb_cur.v[b], acc[a][b], 0, 0, 0);
      fa_cur = fa_nxt;
      fb_cur = fb_nxt;
      // pin the issue order: this step's (prefetch) LDS reads first, then its MFMAs -> the reads'
      // latency is covered by MB*NB MFMAs (>= 256 cycles) instead of being waited for at once
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MB * NB, 0);
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  float* out = p.out + (size_t)z * p.Cout * p.Ktot;
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = co0 + wm * WM + MB * ((r & 3) + 8 * (r >> 2) + 4 * lh) + a;
      if (row >= p.Cout) continue;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int col = k0 + wn * WN + NB * li + b;
        if (col < p.Ktot) out[(size_t)row * p.Ktot + col] = acc[a][b][r];
      }
    }
}

#ifndef EVK_SK_NT
#define EVK_SK_NT 1   // (528.2 -> 528.6 tiles/s over three interleaved rounds: inside the noise, kept as the right hint)
#endif
#if EVK_SK_NT
#define SK_LD(p) __builtin_nontemporal_load(p)   // the partials' last reader
#else
#define SK_LD(p) (*(p))
#endif
// out[i] = sum_z ws[z][i]  (fixed order)
// `lanes` threads share one 16-byte element: lane l sums the splits z = l, l + lanes, ... (fixed order), the lanes are
// folded through LDS in index order => the result depends on (splitk, lanes) only, never on timing.  With one thread
// per element a small weight tensor with many splits (1x1 convolutions on 128^2 maps: 4096 floats x 1024 splits) ran on
// four workgroups and took 78 us — three times the weight-gradient kernel it follows.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, size_t n4,
                                                            int splitk, size_t stride4, int lanes) {
  __shared__ f32x4 red[256];
  const f32x4* w4 = reinterpret_cast<const f32x4*>(ws);
  f32x4* o4 = reinterpret_cast<f32x4*>(out);
  const int epb = 256 / lanes;                     // elements per workgroup
  const int e = threadIdx.x % epb, l = threadIdx.x / epb;
  for (size_t i0 = (size_t)blockIdx.x * epb; i0 < n4; i0 += (size_t)gridDim.x * epb) {
    const size_t i = i0 + e;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
      int z = l;
      for (; z + 3 * lanes < splitk; z += 4 * lanes) {  // four independent 16-byte loads in flight
        const f32x4 a = SK_LD(w4 + (size_t)z * stride4 + i), b = SK_LD(w4 + (size_t)(z + lanes) * stride4 + i);
        const f32x4 c = SK_LD(w4 + (size_t)(z + 2 * lanes) * stride4 + i), d = SK_LD(w4 + (size_t)(z + 3 * lanes) * stride4 + i);
        s += (a + b) + (c + d);
      }
      for (; z < splitk; z += lanes) s += SK_LD(w4 + (size_t)z * stride4 + i);
    }
    if (lanes > 1) {
      red[threadIdx.x] = s;
      __syncthreads();
      if (l == 0 && i < n4) {
        for (int k = 1; k < lanes; ++k) s += red[k * epb + e];
        o4[i] = s;
      }
      __syncthreads();
    } else if (i < n4) {
      o4[i] = s;
    }
  }
}

// Column sums of a [rows][C] matrix: stage 1 -> partial[blk][C], stage 2 -> out[C].  C % 4 == 0.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ src, float* __restrict__ partial,
                                                             int64_t rows, int C, int64_t rows_per_blk) {
  __shared__ f32x4 red[256];
  const int c4 = C >> 2;
  const int tpc = min(c4, 256);       // threads across channels
  const int rl = 256 / tpc;           // row lanes
  const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r1 = min(rows, r0 + rows_per_blk);
  for (int cb = tc; cb < c4; cb += tpc) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (tr < rl)
      for (int64_t r = r0 + tr; r < r1; r += rl) s += *reinterpret_cast<const f32x4*>(src + r * C + cb * 4);
    red[threadIdx.x] = s;
    __syncthreads();
    if (tr == 0) {
      for (int k = 1; k < rl; ++k) s += red[k * tpc + tc];
      *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.x * C + cb * 4) = s;
    }
    __syncthreads();
  }
}
// 32 channels x 8 partial-lanes per workgroup: coalesced 128-byte rows, fp64, LDS fold
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                           int nblk, int C) {
  // 8 channels x 32 partial-lanes per workgroup, four loads in flight per lane (as the BatchNorm finalisation)
  __shared__ double red[32][8];
  const int tc = threadIdx.x & 7, tl = threadIdx.x >> 3;
  const int c = xcd_remap((int)blockIdx.x, (int)gridDim.x) * 8 + tc;   // (neighbouring channel groups on ONE XCD: bn.hip, bn_fin_group)
  double s = 0.0;
  if (c < C) {
    int b = tl;
    for (; b + 96 < nblk; b += 128) {
      const float a0 = partial[(size_t)b * C + c], a1 = partial[(size_t)(b + 32) * C + c];
      const float a2 = partial[(size_t)(b + 64) * C + c], a3 = partial[(size_t)(b + 96) * C + c];
      s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    }
    for (; b < nblk; b += 32) s += (double)partial[(size_t)b * C + c];
  }
  red[tl][tc] = s;
  __syncthreads();
  if (tl == 0 && c < C) {
    for (int k = 1; k < 32; ++k) s += red[k][tc];
    out[c] = (float)s;
  }
}

WGradPlan plan_wgrad(const evk_conv_desc* d, int x3, int planes, int tr, int shared) {
  WGradPlan pl;
  const int Ktot = d->kh * d->kw * d->Cin;
  const int M = d->N * d->Ho * d->Wo;
  pl.bm = d->Cout <= 64 ? 64 : 128;
  pl.bn = Ktot <= 64 ? 64 : 128;
  // wide wave-specialised tile (128 x 256, one 8-wave workgroup per CU) when both dimensions are there
  // (planner constants; ONLY under EVK_TUNE=1 — the survey tool tools/autotune_wgrad.py — are they read from the environment,
  // on every call, and clamped to values the plan below can divide by)
  static const bool tune = getenv("EVK_TUNE") != nullptr;
  auto knob = [&](const char* name, int dflt, int lo) {
    const char* v = tune ? getenv(name) : nullptr;
    const int k = (v && *v) ? atoi(v) : dflt;
    return k < lo ? lo : k;
  };
  const int ws_mode = knob("EVK_WG_WS", 1, 0);
  // (its gathers are raw buffer loads: 32-bit byte offsets, so both tensors must stay below 2 GiB)
  const bool fits32 = (long long)d->N * d->H * d->W * d->Cin * 4 < 0x7fffffffLL &&
                      (long long)d->N * d->Ho * d->Wo * d->Cout * 4 < 0x7fffffffLL;
  // (Cout below 128 leaves rows of the 128-row tile empty: 96 / 64 measured level on C5)
  const int ws_min = knob("EVK_WG_WS_MINCOUT", 128, 1);
  pl.ws = (x3 && ws_mode && d->Cout >= ws_min && Ktot >= 256 && fits32) ? 1 : 0;
  if (tr) { pl.ws = 1; pl.bm = 128; }
  if (pl.ws) pl.bn = 256;
  pl.tiles_co = ceil_div(d->Cout, pl.bm);
  pl.tiles_k = tr == 2 ? d->Cin / 64 : ceil_div(Ktot, pl.bn);
  const int tiles = pl.tiles_co * pl.tiles_k;
  // Split the pixel reduction so that the grid fills WHOLE rounds of the machine: slots = 256 CUs x
  // resident workgroups per CU (LDS-limited: 64 KB tiles -> 2, 48 KB -> 3, 32 KB -> 4).  A grid of
  // 2.04 rounds costs 3 (measured: 1044 workgroups on 512 slots ran at 63 % MFMA utilisation).
  const int rounds = knob("EVK_WG_ROUNDS", 1, 1), min_chunk = knob("EVK_WG_MINCHUNK", 256, 32);
  const int lds_kb = 2 * BKP * (pl.bm + pl.bn) * 4 / 1024;
  // split kernel: single-buffered 3-plane bf16 stage (48 KB at 128x128), residency set by its VGPRs
  const int per_cu = pl.ws ? 1 : x3 ? (pl.bm + pl.bn >= 256 ? 3 : 4)
                        : (lds_kb >= 64 ? 2 : (lds_kb >= 48 ? 3 : 4));
  // A wide-tile kernel (one 8-wave workgroup per CU, up to 700 us long) that runs BESIDE the backward chain — the caller says
  // so with EVK_CONV_WGRAD_SHARED — is split for HALF of the CUs: sixteen workgroups per XCD, the other sixteen CUs of every XCD
  // stay with the main stream's kernels instead of queueing behind it.  Round 5, three interleaved rounds per box, tiles/s:
  // 100 % 545.9 / 541.8 / 560.0, 75 % 564.7, 62 % 545.6, 56 % 544.6, 50 % 557.4 / 552.2 / 565.9 (+1.1 .. +2.1 %), 44 % 545.8,
  // 37 % 531.2, 25 % 469.7; the nine-tap planar kernel alone at 50 %: +1.1 %, the x3ws kernels alone: +0.3 %; the 128 x 128
  // single-role tiles (3-4 workgroups per CU) do not care (percent; 100 = as if alone).
  const int fill = (shared && pl.ws) ? knob("EVK_WG_SHARED_FILL", 50, 1) : 100;
  const int slots = 256 * per_cu * (fill > 0 && fill <= 100 ? fill : 100) / 100;
  int maxsplit = ceil_div(M, min_chunk);  // at least min_chunk pixels per split
  int sk = (rounds * slots) / tiles;      // floor: never spill into an extra, nearly empty round
  if (sk > maxsplit) sk = maxsplit;
  if (sk < 1) sk = 1;
  int chunk = ceil_div(M, sk);
  chunk = ((chunk + BKP - 1) / BKP) * BKP;
  pl.chunk = chunk;
  pl.splitk = ceil_div(M, chunk);
  return pl;
}
static int colsum_blocks(int64_t rows) {
  int64_t b = (rows + 255) / 256;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_wgrad(const WGradArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)2 * BKP * (BM + BN) * sizeof(float);
  hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WAVES_M, WAVES_N>), dim3(a.tiles_co * a.tiles_k * a.splitk),
                     dim3(256), lds, stream, a);
  return check_launch("conv_wgrad");
}

// column sums of a [rows][c] fp32 matrix (bias gradients): also used by conv_transpose.hip
size_t colsum_workspace_bytes(int64_t rows, int c) { return (size_t)colsum_blocks(rows) * c * sizeof(float) + 256; }
int launch_colsum(const float* src, float* out, int64_t rows, int c, float* workspace, hipStream_t st) {
  const int nblk = colsum_blocks(rows);
  const int64_t rpb = (rows + nblk - 1) / nblk;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, src, workspace, rows, c, rpb);
  int rc = check_launch("colsum_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3((c + 7) / 8), dim3(256), 0, st, (const float*)workspace, out, nblk, c);
  return check_launch("colsum_final");
}

}  // namespace evk

using namespace evk;

static size_t wgrad_ws_bytes(const evk_conv_desc* d, int x3) {
  if (!d) return 0;
  const WGradPlan pl = plan_wgrad(d, x3), pl2 = plan_wgrad(d, x3, 2);   // the f16x2 plan may split differently
  const size_t Ktot = (size_t)d->kh * d->kw * d->Cin;
  int sk = pl.splitk > pl2.splitk ? pl.splitk : pl2.splitk;
  if (x3 && d->Cin % 64 == 0) {   // ... and the planar kernels'
    for (int tr = 1; tr <= 2; ++tr) { const int sk3 = plan_wgrad(d, x3, 2, tr).splitk; sk = sk > sk3 ? sk : sk3; }
  }
  size_t a = sk > 1 ? (size_t)sk * d->Cout * Ktot * sizeof(float) : 0;
  size_t b = (size_t)colsum_blocks((int64_t)d->N * d->Ho * d->Wo) * d->Cout * sizeof(float);
  return (a > b ? a : b) + 256;
}

extern "C" size_t evk_conv2d_wgrad_workspace_bytes(const evk_conv_desc* d) { return wgrad_ws_bytes(d, 0); }
extern "C" size_t evk_conv2d_wgrad_x3_workspace_bytes(const evk_conv_desc* d) { return wgrad_ws_bytes(d, 1); }

static int conv_wgrad_any(const evk_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                          void* workspace, size_t workspace_bytes, void* stream, int x3, int planes = 3,
                          const uint32_t* x_scale = nullptr, const uint32_t* dy_scale = nullptr, uint32_t pk_flags = 0) {
  EVK_REQUIRE(d && x && dy && dw, EVK_E_INVALID, "conv2d_wgrad: null pointer");
  EVK_REQUIRE(d->Cin % 4 == 0 && d->Cout % 4 == 0, EVK_E_UNSUPPORTED,
              "conv2d_wgrad: Cin=%d and Cout=%d must be multiples of 4", d->Cin, d->Cout);
  EVK_REQUIRE(workspace_bytes >= wgrad_ws_bytes(d, x3) && (workspace || workspace_bytes == 0), EVK_E_WORKSPACE,
              "conv2d_wgrad: workspace %zu < %zu", workspace_bytes, wgrad_ws_bytes(d, x3));
  EVK_REQUIRE((long long)d->N * d->H * d->W * d->Cin < 0x7fffffffLL &&
                  (long long)d->N * d->Ho * d->Wo * d->Cout < 0x7fffffffLL,
              EVK_E_UNSUPPORTED, "conv2d_wgrad: tensors of 2^31 or more elements are not supported");
  hipStream_t st = (hipStream_t)stream;
  const int planar = (pk_flags & (EVK_CONV_X_PLANAR | EVK_CONV_DY_PLANAR)) ? 1 : 0;   // (workspace: sized for the unshared plan, which splits more)
  EVK_REQUIRE(!planar || ((pk_flags & EVK_CONV_X_PLANAR) && (pk_flags & EVK_CONV_DY_PLANAR) && planes == 2 && !dbias),
              EVK_E_UNSUPPORTED, "conv2d_wgrad: planar operands come in pairs (x and dy), f16x2 only, no bias gradient");
  const int nine = planar && wgrad_tr_nine_tap(d) ? 1 : 0;
  const WGradPlan pl = plan_wgrad(d, x3, planes, planar ? 1 + nine : 0, (pk_flags & EVK_CONV_WGRAD_SHARED) ? 1 : 0);
  WGradArgs a{};
  a.planes = planes;
  a.planar = planar;
  a.x_scale = x_scale; a.dy_scale = dy_scale;
  a.x_packed = (pk_flags & EVK_CONV_X_PACKED) ? 1 : 0;
  a.dy_packed = (pk_flags & EVK_CONV_DY_PACKED) ? 1 : 0;
  EVK_REQUIRE(!(a.dy_packed && dbias), EVK_E_UNSUPPORTED, "conv2d_wgrad: the bias gradient needs dy as fp32");
  a.x = x; a.dy = dy;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.kh = d->kh; a.kw = d->kw; a.cpt = d->Cin / 4;
  a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
  a.M = d->N * d->Ho * d->Wo;
  a.Ktot = d->kh * d->kw * d->Cin;
  a.chunk = pl.chunk; a.tiles_co = pl.tiles_co; a.tiles_k = pl.tiles_k; a.splitk = pl.splitk;
  a.fd_hw = make_fastdiv((uint32_t)(d->Ho * d->Wo));
  a.fd_w = make_fastdiv((uint32_t)d->Wo);
  a.out = pl.splitk > 1 ? (float*)workspace : dw;
  int rc;
  if (planar) {
    EVK_REQUIRE(wgrad_tr_applicable(a), EVK_E_UNSUPPORTED, "conv2d_wgrad: planar operands need Cin %% 64 == 0 and Cout %% 64 == 0 "
                "(Cin=%d Cout=%d) and tensors below 2 GiB", d->Cin, d->Cout);
    rc = launch_wgrad_tr(a, nine, st);
  } else if (x3) rc = launch_wgrad_x3(a, pl, st);
  else if (pl.bm == 128 && pl.bn == 128) rc = launch_wgrad<128, 128, 2, 2>(a, st);
  else if (pl.bm == 64 && pl.bn == 128) rc = launch_wgrad<64, 128, 2, 2>(a, st);
  else if (pl.bm == 128 && pl.bn == 64) rc = launch_wgrad<128, 64, 2, 2>(a, st);
  else rc = launch_wgrad<64, 64, 2, 2>(a, st);
  if (rc) return rc;
  if (pl.splitk > 1) {
    const size_t n = (size_t)d->Cout * a.Ktot;  // multiple of 4 since Cin % 4 == 0
    const size_t n4 = n / 4;
    // enough threads to fill the chip: lanes per element = smallest power of two with n4 * lanes >= 128 K (<= 64)
    int lanes = 1;
    while (lanes < 64 && lanes * 2 <= pl.splitk && n4 * (size_t)lanes < 131072) lanes *= 2;
    const size_t epb = 256 / lanes;
    const int blocks = (int)((n4 + epb - 1) / epb > 4096 ? 4096 : (n4 + epb - 1) / epb);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, n4,
                       pl.splitk, n4, lanes);
    rc = check_launch("splitk_reduce");
    if (rc) return rc;
  }
  if (dbias) {
    const int64_t rows = a.M;
    const int nblk = colsum_blocks(rows);
    const int64_t rpb = (rows + nblk - 1) / nblk;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, dy, (float*)workspace, rows, d->Cout, rpb);
    rc = check_launch("colsum_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((d->Cout + 7) / 8), dim3(256), 0, st, (const float*)workspace,
                       dbias, nblk, d->Cout);
    rc = check_launch("colsum_final");
  }
  return rc;
}

extern "C" int evk_conv2d_wgrad(const evk_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                void* workspace, size_t workspace_bytes, void* stream) {
  return conv_wgrad_any(d, x, dy, dw, dbias, workspace, workspace_bytes, stream, 0);
}

extern "C" int evk_conv2d_wgrad_bf16(const evk_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(d && d->Cin % 4 == 0 && d->Cout % 4 == 0, EVK_E_UNSUPPORTED, "conv2d_wgrad_bf16: channels must be multiples of 4");
  return conv_wgrad_any(d, x, dy, dw, dbias, workspace, workspace_bytes, stream, 1, 1);
}

// f16x2 arithmetic (conv_igemm.hip: evk_conv2d_fwd_f16x2): both operands are scaled activations
extern "C" int evk_conv2d_wgrad_f16x2(const evk_conv_desc* d, const float* x, const uint32_t* x_absmax, const float* dy,
                                      const uint32_t* dy_absmax, float* dw, float* dbias, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(d && d->Cin % 4 == 0 && d->Cout % 4 == 0, EVK_E_UNSUPPORTED, "conv2d_wgrad_f16x2: channels must be multiples of 4");
  EVK_REQUIRE(x_absmax && dy_absmax, EVK_E_INVALID, "conv2d_wgrad_f16x2: null scales");
  return conv_wgrad_any(d, x, dy, dw, dbias, workspace, workspace_bytes, stream, 1, 2, x_absmax, dy_absmax);
}

// flags: EVK_CONV_X_PACKED / EVK_CONV_DY_PACKED — that operand holds packed words (evk_pack_f16x2) instead of fp32
extern "C" int evk_conv2d_wgrad_f16x2_ex(const evk_conv_desc* d, const void* x, const uint32_t* x_absmax, const void* dy,
                                         const uint32_t* dy_absmax, float* dw, float* dbias, void* workspace,
                                         size_t workspace_bytes, uint32_t flags, void* stream) {
  EVK_REQUIRE(d && d->Cin % 4 == 0 && d->Cout % 4 == 0, EVK_E_UNSUPPORTED, "conv2d_wgrad_f16x2_ex: channels must be multiples of 4");
  EVK_REQUIRE(x_absmax && dy_absmax, EVK_E_INVALID, "conv2d_wgrad_f16x2_ex: null scales");
  EVK_REQUIRE((flags & ~(EVK_CONV_X_PACKED | EVK_CONV_DY_PACKED | EVK_CONV_X_PLANAR | EVK_CONV_DY_PLANAR | EVK_CONV_WGRAD_SHARED)) == 0, EVK_E_INVALID,
              "conv2d_wgrad_f16x2_ex: unknown flag 0x%x", flags);
  return conv_wgrad_any(d, reinterpret_cast<const float*>(x), reinterpret_cast<const float*>(dy), dw, dbias, workspace,
                        workspace_bytes, stream, 1, 2, x_absmax, dy_absmax, flags);
}

extern "C" int evk_conv2d_wgrad_x3(const evk_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(d && d->Cin % 4 == 0 && d->Cout % 4 == 0, EVK_E_UNSUPPORTED, "conv2d_wgrad_x3: channels must be multiples of 4");
  return conv_wgrad_any(d, x, dy, dw, dbias, workspace, workspace_bytes, stream, 1);
}
