// 3x3 stride-1 "same" convolution (forward and data gradient) in the f16x2 split arithmetic with the Winograd F(2,3)
// minimal filtering along the image's x axis — 4 multiplications per PAIR of output pixels and kernel row where the direct
// form (conv3x3_halo_x3.hip) needs 6: 1.5x fewer MFMA passes on the layers that run against the chip's power cap
// (the three products of the split arithmetic are 9.6 of the 22.5 convolution ms).  Same reference
// call sites as the halo kernel (3x3 convolutions of _resnets.py:21-29, fpn.py:72-73,165).
//
//   out[y][2j]   = sum_ky sum_ci  m0 + m1 + m2          m_xi = V_xi[y + ky][j][ci] * U_xi[ky][co][ci]
//   out[y][2j+1] = sum_ky sum_ci  m1 - m2 - m3
//   V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3) of the four input pixels x = 2j - 1 .. 2j + 2      (coefficients +-1: one fp32
//   U = (g0, (g0 + g1 + g2)/2, (g0 - g1 + g2)/2, g2) of the three taps of kernel row ky            rounding per element)
//
// Why one axis only: the transform-domain accumulators are 4 per output pair (2x the direct form) — in two dimensions
// 16 per 2x2 outputs (4x), which leaves a 64-tile x 64-channel workgroup tile whose weight fragments (16 xi x 64 co x 16 ci
// x 4 B per 1536 MFMA cycles = 43 B/clk/CU from L2) and four-fold redundant input transform (5 VALU per transformed
// element) would sit beside the matrix pipe as co-bottlenecks; along y the kernel rows stay what they are in the halo
// kernel, whole-row shifts of one LDS image.  |V| <= 2 max|d| and |U| <= 1.5 max|g|: both fit the two bits of headroom the
// operand scale leaves below fp16's largest number (x3_common.hpp: x / s < 2^14), so scales, split and products are the
// direct kernel's.  K order: chunk of 16 channels, kernel row, xi.  Values differ from the direct kernels' by rounding
// (tests/test_wino_gpu.py: both ~2e-6 from fp64).
//
// Workgroup: PH x 16 output pixels (= PH x 8 pairs = GEMM rows) x BN channels, eight SYMMETRIC waves (4 along M x 2 along N; the
// 128 accumulator registers of a 32-pair x 64-channel x 4-xi wave tile leave room for two waves per SIMD and no more, so
// there are no staging waves): every wave issues a quarter of an iteration's weight DMA (pre-transformed, pre-split
// planes [plane][ky][xi][chunk][Cout][16], `buffer_load ... lds`, two stages), and once per chunk transforms one item of
// the NEXT chunk's halo — 4 pixels x 4 channels from global memory into the four xi images — between the MFMAs of the
// current one.  LDS: image [2 stages][2 planes][4 xi][PH + 2 rows][8 pairs][32 B] + weights [2 stages][4 xi][2 planes][BN][32 B].
#include "igemm_common.hpp"
#include "split_weight.hpp"
#include "lds_dma.hpp"
#include <stdlib.h>
#include <string.h>

namespace evk {

#ifndef EVK_WINO_ABL
#define EVK_WINO_ABL 0   // timing ablations (tools/build_variant.sh -DEVK_WINO_ABL=n; wrong results): 1 no weight DMA, 2 no halo
#endif                   // loads / transform, 4 no MFMA, 8 no barrier, 16 no output stores
constexpr int kAbl = EVK_WINO_ABL;
constexpr int kWPW = 16, kWJ = kWPW / 2;       // output patch width, pairs per patch row
constexpr int kWCh = 16, kWRB = kWCh * 2;      // channels per chunk; bytes per LDS row and plane

__device__ __forceinline__ int wino_half(int parity, int c16) { return (c16 ^ (parity & 1)) << 4; }

// MW = 8 matrix waves (4 along M x 2 along N, two per SIMD) + 4 staging waves, as in the halo kernel's widest form: at 128
// accumulator registers + one fragment set a matrix wave stays under the 168 registers that three waves per SIMD allow,
// and it issues no vector-memory instruction — a wave that issues DMA blocks on the CU's in-flight limit and multiplies
// nothing meanwhile (the first form of this kernel, eight symmetric waves that each issued a quarter of
// the ring, spent a quarter of every iteration that way: 819 us on 3x3x256 @128^2 against 745 with the DMA compiled out).
template <int BN, int PH, bool PK>
__global__ __launch_bounds__(768) void conv3x3_wino_x3_kernel(const IGemmArgs p, int tiles_y, int tiles_x) {
  constexpr int kHR = PH + 2;                          // halo rows
  constexpr int kXi = kHR * kWJ * kWRB;                // one xi image of one plane
  constexpr int kAPlane = 4 * kXi;
  constexpr int kAStage = 2 * kAPlane;                 // 36864 B at PH 16
  constexpr int kBStage = 4 * 2 * BN * kWRB;           // 32768 B at BN 128
  constexpr int WMR = PH * kWJ / 4;                    // GEMM rows (pairs) per matrix wave
  constexpr int MB = WMR / 32, WN = BN / 2, NB = WN / 32;
  static_assert(MB == 1, "wave tile: 32 pairs");
  constexpr int kItems = kHR * kWJ * 4;                // (halo row, pair, channel quad): 576 at PH 16
  constexpr int AI = (kItems + 255) / 256;             // items per staging thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smemw[];
  unsigned char* const Abase = smemw;                  // [2][kAStage]
  unsigned char* const Bbase = smemw + 2 * kAStage;    // [2][kBStage]

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % p.tiles_n;
  int t = bid / p.tiles_n;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  const int n = t / tiles_y;
  const int Y0 = ty * PH, X0 = tx * kWPW, n0 = tile_n * BN;
  const int nchunk = (p.Cs + kWCh - 1) / kWCh;
  const int niter = nchunk * 3;
  const int tid = threadIdx.x;

  if (tid >= 512) {
    // ------------------------------------------------------------------ staging waves
    const int ptid = tid - 512;
    const int swave = __builtin_amdgcn_readfirstlane(ptid >> 6), lane = ptid & 63;
    float a_inv = 1.f;
    if constexpr (!PK) a_inv = op_scale(act_absmax(p.a_scale)).inv;
    // the halo through a buffer descriptor: an offset beyond it reads zeros — the image border and a partial last chunk's
    // missing channel quads cost no select
    const __amdgpu_buffer_rsrc_t rs_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, (int)((size_t)p.N * p.Hs * p.Ws * p.Cs * 4), 0x00020000);
    // item e = ptid + 256 i: channel quad c4, pair j, halo row hy; its four pixels x = 2j - 1 .. 2j + 2
    uint32_t a_off[AI][4];
    int a_lds[AI], a_c4[AI];
    bool a_has[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int e = ptid + 256 * i;
      a_has[i] = e < kItems;
      const int c4 = e & 3, j = (e >> 2) & 7, hy = a_has[i] ? (e >> 5) : 0;
      a_c4[i] = c4;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int sy = Y0 - 1 + hy, sx = X0 - 1 + 2 * j + tt;
        const bool ok = a_has[i] && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
        a_off[i][tt] = ok ? (uint32_t)(((n * p.Hs + sy) * p.Ws + sx) * p.Cs + c4 * 4) * 4u : kDmaOOB;
      }
      a_lds[i] = (hy * kWJ + j) * kWRB + wino_half(hy, c4 >> 1) + ((c4 & 1) << 3);
    }
    u32x4 ra[AI][4];
    auto load_item = [&](int i, int c) {
      const bool cut = c * kWCh + a_c4[i] * 4 >= p.Cs;   // (Cs % 16 == 8: quads 2 and 3 of the last chunk)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
        ra[i][tt] = (kAbl & 32) ? u32x4{0x3f800000u + (uint32_t)c, 0x40000000u, 0x3f000000u + a_off[i][tt], 0x3fc00000u}   // (no loads)
                                : __builtin_amdgcn_raw_buffer_load_b128(rs_a, cut ? kDmaOOB : a_off[i][tt], c * kWCh * 4, 0);
    };
    // V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3), split into its two fp16 planes, one 8-byte store per xi and plane
    auto store_item = [&](int i, int stage) {
      if (!a_has[i]) return;
      if (kAbl & 16) {   // (loads only: their registers are named, nothing is transformed or stored)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) asm volatile("" :: "v"(ra[i][tt]));
        return;
      }
      unsigned char* A = Abase + stage * kAStage;
      f32x4 d[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const uint32_t w[4] = {ra[i][tt].x, ra[i][tt].y, ra[i][tt].z, ra[i][tt].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) d[tt][e] = PK ? unpack_hl(w[e]) : __builtin_bit_cast(float, w[e]);
      }
      const f32x4 v[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        uint32_t h0, l0, h1, l1, unused = 0;
        split_np<2>(v[xi].x, v[xi].y, a_inv, h0, l0, unused);
        split_np<2>(v[xi].z, v[xi].w, a_inv, h1, l1, unused);
        *reinterpret_cast<uint2*>(A + xi * kXi + a_lds[i]) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(A + kAPlane + xi * kXi + a_lds[i]) = make_uint2(l0, l1);
      }
    };
    // ---- weights by DMA: an iteration's stage is 4 xi x 2 planes x BN rows x 32 B; one instruction moves 32 rows (1 KB),
    // the 16-byte halves of a row swapped on the source side where the fragment reads swap them (odd 8-row groups).  Two
    // stages: iteration it + 1 is issued at the start of iteration it and has the whole iteration to land.
    constexpr int NQ = BN / 32, NBI = 4 * 2 * NQ, PERB = NBI / 4;
    static_assert(NBI % 4 == 0, "weight DMA instructions split over four staging waves");
    const uint32_t tap_bytes = (uint32_t)nchunk * (uint32_t)p.Cd * kWRB, plane_bytes = 12u * tap_bytes;
    const i32x4 rs_b = make_rsrc(p.wgt3, 2u * plane_bytes);
    const uint32_t ldsB = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)Bbase;
    uint32_t bd_voff[PERB], bd_lds[PERB];
#pragma unroll
    for (int i = 0; i < PERB; ++i) {
      const int id = swave * PERB + i;
      const int xi = id / (2 * NQ), pt = (id / NQ) % 2, q = id % NQ;
      const int row = 32 * q + (lane >> 1), half = (lane & 1) ^ ((row >> 3) & 1);
      const int co = n0 + row;
      bd_voff[i] = co < p.Cd ? (uint32_t)pt * plane_bytes + (uint32_t)xi * tap_bytes + (uint32_t)co * kWRB + (uint32_t)half * 16u : kDmaOOB;
      bd_lds[i] = (uint32_t)((xi * 2 + pt) * BN * kWRB + q * 1024);
    }
    auto issue_b = [&](int it, int stage) {
      const int c = it / 3, jy = it - 3 * c;
      const uint32_t soff = (uint32_t)(jy * 4) * tap_bytes + (uint32_t)c * (uint32_t)p.Cd * kWRB;
      const uint32_t S = ldsB + stage * kBStage;
#pragma unroll
      for (int i = 0; i < PERB; ++i) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(bd_voff[i]), "s"(S + bd_lds[i]), "s"(rs_b), "s"(soff) : "memory");
      }
    };
    static_assert(AI <= 3, "one item per iteration of a chunk");
    if (!(kAbl & 1)) issue_b(0, 0);
    if (!(kAbl & 2)) {
#pragma unroll
      for (int i = 0; i < AI; ++i) load_item(i, 0);
#pragma unroll
      for (int i = 0; i < AI; ++i) store_item(i, 0);
      if (nchunk > 1) {
#pragma unroll
        for (int i = 0; i < AI; ++i) load_item(i, 1);
      }
    }
    wait_vmcnt<0>();
    ring_barrier();
    // Iteration (c, jy) carries item jy of the image of chunk c + 1 (all of it in one iteration made that iteration the staging
    // waves' — the matrix waves waited at its barrier: 463 us for the kernel with the MFMAs compiled out) and sends the same
    // item's loads for chunk c + 2 out behind the iteration's DMA, so that the wait for the DMA leaves them in flight.
    for (int c = 0; c < nchunk; ++c) {
#pragma unroll
      for (int jy = 0; jy < 3; ++jy) {
        const int it = 3 * c + jy;
        // (the compiler's own wait in front of the item's first use counts loads only: everything older than the four loads
        // of the previous iteration completed with that iteration's wait)
        if (!(kAbl & 2) && jy < AI && c + 1 < nchunk) store_item(jy, (c + 1) & 1);   // (that image stage: last read in chunk c - 1)
        if (!(kAbl & 1) && it + 1 < niter) issue_b(it + 1, (it + 1) & 1);           // (that weight stage: in iteration it - 1)
        if (!(kAbl & 2) && jy < AI && c + 2 < nchunk) {
          load_item(jy, c + 2);
          wait_vmcnt<4>();
        } else {
          wait_vmcnt<0>();
        }
        ring_barrier();
      }
    }
    return;
  }

  // -------------------------------------------------------------------- matrix waves
  __builtin_amdgcn_s_setprio(3);
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  // A: rows (xi*kHR + wm*4 + jy) * 8 + li of the image; the 16-byte halves swap with the parity of (row >> 3)
  const int a_row0 = (wm * (WMR / kWJ)) * kWJ + li;             // + (xi*kHR + jy) * 8
  int fb[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int row = wn * WN + b * 32 + li;
    fb[b] = row * kWRB + wino_half(row >> 3, lh);
  }
  f32x16 acc[4][NB];
#pragma unroll
  for (int xi = 0; xi < 4; ++xi)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[xi][b][r] = 0.f;

  ring_barrier();
  for (int it = 0; it < niter; ++it) {
    const int c = it / 3, jy = it - 3 * c;
    const unsigned char* A = Abase + (c & 1) * kAStage;
    const unsigned char* B = Bbase + (it & 1) * kBStage;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      bf16x8 fa[2], fbv[NB][2];
      const int arow = a_row0 + (xi * kHR + jy) * kWJ;
      const int aoff = arow * kWRB + wino_half(jy + (li >> 3), lh);
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) fa[pt] = *reinterpret_cast<const bf16x8*>(A + pt * kAPlane + aoff);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) fbv[b][pt] = *reinterpret_cast<const bf16x8*>(B + (xi * 2 + pt) * BN * kWRB + fb[b]);
      if (kAbl & 4) {   // (no matrix work: the fragments are still read)
#pragma unroll
        for (int b = 0; b < NB; ++b) asm volatile("" :: "v"(fa[0]), "v"(fa[1]), "v"(fbv[b][0]), "v"(fbv[b][1]));
        continue;
      }
#pragma unroll
      for (int t6 = 0; t6 < 3; ++t6)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[xi][b] = mfma_np<2>(fbv[b][x3_pb(2, t6)], fa[x3_pa(2, t6)], acc[xi][b]);
    }
    ring_barrier();
  }

  // ---- output transform (in registers: a lane holds one pair, all four xi) and epilogue
  const float sc = op_scale(act_absmax(p.a_scale)).s * op_scale(*p.w_scale).s;
  f32x16 ev[NB], od[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    ev[b] = (acc[0][b] + acc[1][b] + acc[2][b]) * sc;
    od[b] = (acc[1][b] - acc[2][b] - acc[3][b]) * sc;
  }
  const int m = wm * WMR + li;
  const int gy = Y0 + (m >> 3), gx = X0 + 2 * (m & 7);
  const bool in_e = gy < p.Hm && gx < p.Wm, in_o = gy < p.Hm && gx + 1 < p.Wm;
  const size_t roff_e = (((size_t)n * p.Hd + (size_t)(gy * p.dsh + p.doy)) * p.Wd + (size_t)(gx * p.dsw + p.dox)) * p.Cd;
  const size_t roff_o = roff_e + (size_t)p.dsw * p.Cd;
  if (p.bn_part) {
    BnLaneStat st;
    bn_stat_init(st);
    float* scratch = reinterpret_cast<float*>(smemw) + wave * 32 * (WN + 4);
    float* xch = reinterpret_cast<float*>(smemw) + 8 * 32 * (WN + 4);
    igemm_store_rows_stats<NB, WN>(p, ev, in_e ? roff_e : ~(size_t)0, n0, wn, li, lh, scratch, st);
    igemm_store_rows_stats<NB, WN>(p, od, in_o ? roff_o : ~(size_t)0, n0, wn, li, lh, scratch, st);
    bn_part_write<WN, 4, 2>(p, st, (n * tiles_y + ty) * tiles_x + tx, n0, wm, wn, lane, xch);
    return;
  }
  if (kAbl & 16) return;
  AmaxAcc amax_l{0u, p.out_amax != nullptr};
  if (in_e) igemm_store_rows<NB, WN>(p, ev, roff_e, n0, wn, lh, amax_l);
  if (in_o) igemm_store_rows<NB, WN>(p, od, roff_o, n0, wn, lh, amax_l);
  if (p.out_amax) amax_commit(p.out_amax, amax_l.m);
}

// EVK_WINO: 0 = never (the direct halo kernel everywhere), 1 = where measured ahead (default), 2 = wherever it applies (tests)
static int wino_mode() {
  static const int m = getenv("EVK_WINO") ? atoi(getenv("EVK_WINO")) : 1;
  return m;
}

// geometry part of the decision — a pure function of the launch shape, shared with the weight-plane producers
static bool wino_geometry(int N, int Hm, int Wm, int Cs, int Cd) {
  const int mode = wino_mode();
  if (!mode) return false;
  if (Hm < 8 || Wm < 8 || (Cs % 8) != 0 || Cd < 64) return false;
  if (4 * Cs < 3 * ceil_div(Cs, kWCh) * kWCh) return false;
  const long long cover = (long long)ceil_div(Hm, 16) * 16 * ceil_div(Wm, kWPW) * kWPW;
  if (4LL * Hm * Wm < 3 * cover) return false;
  if (mode >= 2) return true;
  // one 16 x 16 patch x 128 channels per workgroup: whole 128-wide column tiles and at least one workgroup per CU (measured,
  // tools/wino_probe.py, us direct -> this: 3x3x256 @128^2 926 -> 815 forward / 802 -> 707 data gradient, 256 -> 128 477 -> 430,
  // 256 @64^2 212 -> 190, 128 @64^2 57 -> 52; behind on 64 -> 64 @128^2 (half of the column tile is padding: 73 -> 103) and on
  // the 32^2 / 16^2 maps, which are not bound by the matrix pipe)
  return (Cd % 128) == 0 && (long long)N * ceil_div(Hm, 16) * ceil_div(Wm, kWPW) * (Cd / 128) >= 256;
}

bool conv3x3_wino_applies(const IGemmArgs& a) {
  if (a.planes != 2 || a.kh != 3 || a.kw != 3 || a.ash != 1 || a.asw != 1) return false;
  if (!((a.oys == 1 || a.oys == -1) && a.oy0 == -a.oys && (a.oxs == 1 || a.oxs == -1) && a.ox0 == -a.oxs)) return false;
  if (a.Hm != a.Hs || a.Wm != a.Ws || (!a.dense_dst && (a.dsh != 1 || a.dsw != 1))) return false;
  if ((long long)a.N * a.Hs * a.Ws * a.Cs * 4 >= 0x7fffffffLL) return false;   // (the halo is loaded through a buffer descriptor)
  // (only where the halo kernel applies as well: the weight-plane producers of the other arithmetics lay those shapes out
  // for it, and a multi-tensor split job is built before the arithmetic is known)
  return conv3x3_halo_applies(a) && wino_geometry(a.N, a.Hm, a.Wm, a.Cs, a.Cd);
}

// the same decision from a convolution descriptor (forward, or stride-1 data gradient) — f16x2 arithmetic only, which the
// caller knows (the weight-plane producers: a scale word is present)
bool conv_desc_uses_wino(const evk_conv_desc* d, int for_dgrad) {
  if (d->kh != 3 || d->kw != 3) return false;
  if (d->stride_h != 1 || d->stride_w != 1 || d->dil_h != 1 || d->dil_w != 1 || d->pad_h != 1 || d->pad_w != 1) return false;
  if (!conv_desc_uses_halo(d, for_dgrad)) return false;
  return for_dgrad ? wino_geometry(d->N, d->H, d->W, d->Cout, d->Cin) : wino_geometry(d->N, d->Ho, d->Wo, d->Cin, d->Cout);
}

template <int BN, int PH, bool PK>
static int launch_wino_t(IGemmArgs& a, hipStream_t stream) {
  a.tiles_n = ceil_div(a.Cd, BN);
  const int tiles_y = ceil_div(a.Hm, PH), tiles_x = ceil_div(a.Wm, kWPW);
  a.tiles_m = a.N * tiles_y * tiles_x;
  bn_stats_setup(a, PH * kWPW, BN, 4, a.tiles_m);
  const size_t lds = (size_t)2 * (2 * 4 * (PH + 2) * kWJ * kWRB) + (size_t)2 * (4 * 2 * BN * kWRB);
  static PerDeviceOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_x3_kernel<BN, PH, PK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const long long nwg = (long long)a.tiles_m * a.tiles_n;
  hipLaunchKernelGGL((conv3x3_wino_x3_kernel<BN, PH, PK>), dim3((unsigned)nwg), dim3(768), lds, stream, a, tiles_y, tiles_x);
  return check_launch("conv3x3_wino_x3");
}

int launch_conv3x3_wino(IGemmArgs& a, hipStream_t stream) {
  if (!conv3x3_wino_applies(a)) return 1;
  return a.a_packed ? launch_wino_t<128, 16, true>(a, stream) : launch_wino_t<128, 16, false>(a, stream);
}

// planes for the Winograd kernel: out[pt][ky*4 + xi][chunk][row][16] fp16 of U / s.  The kernel always correlates
// (source row = gy - 1 + ky, source x = 2j - 1 + t): the data gradient's planes hold the transposed, FLIPPED filter.
__global__ void split_weight_wino_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int Cout, int Cin,
                                         int for_dgrad, const uint32_t* __restrict__ wscale) {
  split_wino_body(w, out, Cout, Cin, for_dgrad, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
                  (size_t)gridDim.x * blockDim.x, wscale);
}

int launch_split_weight_wino(const float* w, uint16_t* out, int Cout, int Cin, int for_dgrad, hipStream_t st,
                             const uint32_t* wscale) {
  const int rows = for_dgrad ? Cin : Cout, K = for_dgrad ? Cout : Cin;
  const size_t total = (size_t)3 * ((K + kWCh - 1) / kWCh) * rows * (kWCh / 2);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(split_weight_wino_kernel, dim3(blocks), dim3(256), 0, st, w, out, Cout, Cin, for_dgrad, wscale);
  return check_launch("split_weight_wino");
}

}  // namespace evk
