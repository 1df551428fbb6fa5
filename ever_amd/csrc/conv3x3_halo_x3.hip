// 3x3 stride-1 "same" convolution (forward and data gradient) in the split-MFMA arithmetic with an LDS-resident
// input halo — the im2col gather of conv_igemm_x3*.hip re-reads every input pixel nine times from L2 (once per
// tap) and splits it nine times; here a workgroup owns an 8 x 16 patch of output pixels (= 128 GEMM rows), loads
// the 10 x 18 halo of one 16-channel chunk ONCE, splits it once into the three bf16 planes, and forms the nine taps'
// A fragments by shifted ds_read_b128 from the same LDS image: 6.4x fewer activation bytes from L2 and 6.4x fewer
// split VALU per MFMA (the two costs that hold the generic kernel at ~195 of the ~290 TFLOP/s its matrix waves
// reach alone).  Replaces the same reference call sites (3x3 convolutions of _resnets.py:21-29,
// fpn.py:72-73,165).
//
// K order: for each 16-channel chunk c, for each kernel row jy: the three taps jx (one barrier per 3 taps = 72
// MFMAs per matrix wave).  Weights come pre-split in a layout made for this order, [plane][tap][chunk][Cout][16]
// (split_weight_halo_kernel; 4 KB contiguous per (plane, tap, chunk) for a 128-row tile).
// Wave roles as in conv_igemm_x3ws.hip: waves 0-3 MFMA + fragment reads, waves 4-7 staging.
// LDS: halo [2 stages][3 planes][10 x 32 slots][32 B] (pitch 32 >= 18 so that the 16-lane groups of a ds_read_b128
// always cover 16 distinct row residues), weights [2 stages][3 taps][3 planes][128 rows][32 B]; 16-byte halves of a
// 32-byte row are swapped on odd 8-row groups (conflict-free b128 reads for any tap shift).
#include "igemm_common.hpp"
#include "split_weight.hpp"
#include "lds_dma.hpp"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// (Round 5 negative result, removed from this file in round 6 — commit 6ed01e3 has the code: explicitly software-pipelined matrix
// loops, fragment reads of tap u+1 interleaved with the MFMAs of tap u across the iteration's barrier, no exposed LDS wait left in
// the ISA: 3x3x256 @128^2 762-781 -> 797-810 us.  The loop is not latency-bound, it runs against the chip's power cap — PMC:
// matrix pipe 0.69 busy at 1.76 GHz; the same launch 1023 us on random activations, 743 on zeros — and a schedule with fewer
// stalls returns its gain as a lower clock)

namespace evk {

constexpr int kPW = 16;                        // output patch width; height PH = 8 or 16 (template)
// halo row pitch in slots: 32 (>= 18, multiple of 16: every 16-lane group of a ds_read_b128 covers 16 distinct row
// residues) for the 8-row patch; 18 for the 16-row patch, whose halo would not fit twice otherwise (2 of 16 lanes
// of a read group then share a slot: +1 LDS cycle on a path that is not the bottleneck)
// PL = LDS planes per operand: 2 under the f16x2 arithmetic (its third plane does not exist), else 3.  With two planes
// the 16-row patch affords pitch 32 as well: at pitch 18 EVERY fragment read is 2-way bank-conflicted for the lane
// groups ds_read_b128 is really served in ({0-3,12-15,20-27} / {4-11,16-19,28-31}, not 16 contiguous lanes; PMC: a third
// of the kernel's LDS cycles; exhaustive check in the round-2 notes), at pitch 32 none is.
template <int PH, int PL = 3> struct HaloGeom {
  static constexpr int kHP = (PH == 8 || PL == 2) ? 32 : 18;
  static constexpr int kHSlots = (PH + 2) * kHP;
  static constexpr int kHaloPix = (PH + 2) * (kPW + 2);
};
constexpr int kCh = 16;                        // channels per chunk = one 32x32x16 k-block
constexpr int kRB = kCh * 2;                   // bytes per row and plane

__device__ __forceinline__ int half_off(int row, int c16) { return row * kRB + ((c16 ^ ((row >> 3) & 1)) << 4); }

// MW = matrix waves: 4 (2 x 2, one per SIMD) or 8 (4 x 2, two per SIMD: one's fragment reads under the other's MFMAs)
template <int BN, int PH, int NPX, bool WDMA, int MW>
__global__ __launch_bounds__(256 + 64 * MW) void conv3x3_halo_x3_kernel(const IGemmArgs p, int tiles_y, int tiles_x) {
  constexpr int NP = X3Mode<NPX>::NP;
  constexpr bool PK = X3Mode<NPX>::PK;   // the activation operand arrives packed (x3_common.hpp)
  constexpr int PL = NP == 2 ? 2 : 3;
  using Geo = HaloGeom<PH, PL>;
  constexpr int kPH = PH, kHP = Geo::kHP, kHSlots = Geo::kHSlots, kHaloPix = Geo::kHaloPix;
  constexpr int kAStage = PL * kHSlots * kRB;         // three planes: 30720 B (PH 8) / 31104 B (PH 16); two: 20480 / 36864
  constexpr int kBStage = 3 * PL * BN * kRB;          // 36864 B at BN = 128 (24576 with two planes)
  constexpr int MWM = MW / 2;                         // matrix waves along M
  constexpr int WMR = PH * kPW / MWM;                 // rows per matrix wave
  constexpr int WN = BN / 2, NB = WN / 32, MB = WMR / 32;
  constexpr int AI = (kHaloPix * 4 + 255) / 256;      // halo items per staging thread
  // f16x2: the weight tiles go global -> LDS by DMA (they are pre-split planes in this kernel's own order: nothing to
  // convert), three stages deep; the other arithmetics stage them through registers, two stages
  constexpr bool BDMA = WDMA && NP == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  unsigned char* const Abase = smem3;                 // [2][kAStage]
  unsigned char* const Bbase = smem3 + 2 * kAStage;   // [BDMA ? 3 : 2][kBStage]

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % p.tiles_n;
  int t = bid / p.tiles_n;
  const int tx = t % tiles_x;
  t /= tiles_x;
  const int ty = t % tiles_y;
  const int n = t / tiles_y;
  const int Y0 = ty * kPH, X0 = tx * kPW, n0 = tile_n * BN;
  const int nchunk = (p.Cs + kCh - 1) / kCh;   // (a partial last chunk: its missing channels are zero in both operands)
  const int niter = nchunk * 3;
  const int tid = threadIdx.x;

  if (tid >= 64 * MW) {
    // ------------------------------------------------------------------ staging waves
    const int ptid = tid - 64 * MW;
    // halo items: (pixel 0..179, float4 q 0..3); three per thread, the last pass partially filled
    int a_src[AI], a_lds[AI];
    bool a_ok[AI], a_has[AI], a_okc[AI];   // a_okc: a_ok for the chunk last loaded (channels past Cs in a partial last chunk)
    const int tail_ch = p.Cs % kCh;        // 0: every chunk is whole
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int e = ptid + 256 * i;
      a_has[i] = e < kHaloPix * 4;
      const int pix = a_has[i] ? (e >> 2) : 0, q = e & 3;
      const int hy = pix / (kPW + 2), hx = pix - hy * (kPW + 2);
      const int sy = Y0 - 1 + hy, sx = X0 - 1 + hx;
      a_ok[i] = a_has[i] && (unsigned)sy < (unsigned)p.Hs && (unsigned)sx < (unsigned)p.Ws;
      a_okc[i] = a_ok[i];
      a_src[i] = a_ok[i] ? ((n * p.Hs + sy) * p.Ws + sx) * p.Cs + q * 4 : 0;
      const int hr = hy * kHP + hx;
      a_lds[i] = hr * kRB + (((q >> 1) ^ ((hr >> 3) & 1)) << 4) + ((q & 1) << 3);
    }
    // weight items: 3 taps x 3 planes x BN rows x 2 halves of 16 B
    constexpr int kBItems = 3 * 3 * BN * 2;
    constexpr int BI = (kBItems + 255) / 256;
    int b_src[BI], b_lds[BI];
    bool b_has[BI];
    const size_t tap_stride = (size_t)nchunk * p.Cd * kCh;   // bf16 elements between taps of one plane
    const size_t plane_stride = 9 * tap_stride;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int e0 = ptid + 256 * i;
      b_has[i] = e0 < kBItems;
      const int e = b_has[i] ? e0 : 0;
      const int half = e & 1, row = (e >> 1) % BN, rest = e / (2 * BN);  // rest = jx * 3 + pt
      const int pt = rest % 3, jx = rest / 3;
      b_has[i] = b_has[i] && pt < NP;   // plain bf16 / f16x2: only the h / the h and l planes of the weights are staged
      int co = n0 + row;
      co = co < p.Cd ? co : p.Cd - 1;
      b_src[i] = (int)(pt * plane_stride + (size_t)jx * tap_stride + (size_t)co * kCh + half * 8);
      b_lds[i] = (jx * PL + (pt < PL ? pt : 0)) * BN * kRB + half_off(row, half);
    }
    f32x4 ra[AI];
    u32x4 rbv[BI];
    float a_inv = 1.f;   // f16x2: 1 / activation scale
    if constexpr (NP == 2) a_inv = op_scale(act_absmax(p.a_scale)).inv;
    auto load_a = [&](int c) {
      const bool partial = tail_ch != 0 && c == nchunk - 1;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        a_okc[i] = a_ok[i] && (!partial || ((ptid + 256 * i) & 3) * 4 < tail_ch);
        ra[i] = *reinterpret_cast<const f32x4*>(p.src + (a_okc[i] ? a_src[i] + c * kCh : 0));
      }
    };
    auto store_a = [&](int stage) {
      unsigned char* A = Abase + stage * kAStage;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        if (!a_has[i]) continue;
        const f32x4 v = ra[i];
        const bool ok = a_okc[i];
        uint32_t h0, m0 = 0, l0 = 0, h1, m1 = 0, l1 = 0;
        split_op<NP, PK>(ok ? v.x : 0.f, ok ? v.y : 0.f, a_inv, h0, m0, l0);
        split_op<NP, PK>(ok ? v.z : 0.f, ok ? v.w : 0.f, a_inv, h1, m1, l1);
        *reinterpret_cast<uint2*>(A + a_lds[i]) = make_uint2(h0, h1);
        if (NP >= 2) *reinterpret_cast<uint2*>(A + kHSlots * kRB + a_lds[i]) = make_uint2(m0, m1);
        if (NP == 3) *reinterpret_cast<uint2*>(A + 2 * kHSlots * kRB + a_lds[i]) = make_uint2(l0, l1);
      }
    };
    // iteration it = 3*c + jy reads taps (jy, 0..2) of chunk c
    auto load_b = [&](int it) {
      const int c = it / 3, jy = it - 3 * c;
      const size_t off = (size_t)(jy * 3) * tap_stride + (size_t)c * p.Cd * kCh;
#pragma unroll
      for (int i = 0; i < BI; ++i) rbv[i] = *reinterpret_cast<const u32x4*>(p.wgt3 + off + b_src[i]);
    };
    auto store_b = [&](int stage) {
      unsigned char* B = Bbase + stage * kBStage;
#pragma unroll
      for (int i = 0; i < BI; ++i)
        if (b_has[i]) *reinterpret_cast<u32x4*>(B + b_lds[i]) = rbv[i];
    };

    if constexpr (BDMA) {
      // ---- weights by DMA.  A (tap, plane) tile is BN rows x 32 B, contiguous in HBM; one instruction moves 32 rows, the
      // 16-byte halves of a row swapped on the source side where half_off() swaps them in LDS.  Iteration it + 2 is issued
      // during iteration it (ring of three), so a tile has two iterations to land; completion is counted by hand and the
      // barriers are raw (a __syncthreads would drain the DMA: lds_dma.hpp).
      constexpr int NQ = BN / 32, NBI = 3 * PL * NQ, PERB = NBI / 4;
      static_assert(NBI % 4 == 0, "weight DMA instructions split over four staging waves");
      const int swave = __builtin_amdgcn_readfirstlane(ptid >> 6), lane = ptid & 63;
      const uint32_t tap_bytes = (uint32_t)tap_stride * 2u, plane_bytes = 9u * tap_bytes;
      const i32x4 rs_b = make_rsrc(p.wgt3, (uint32_t)PL * plane_bytes);
      const uint32_t ldsB = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)Bbase;
      uint32_t bd_voff[PERB], bd_lds[PERB];
#pragma unroll
      for (int i = 0; i < PERB; ++i) {
        const int id = swave * PERB + i;
        const int jx = id / (PL * NQ), pt = (id / NQ) % PL, q = id % NQ;
        const int row = 32 * q + (lane >> 1), half = (lane & 1) ^ ((row >> 3) & 1);
        const int co = n0 + row;
        bd_voff[i] = co < p.Cd ? (uint32_t)pt * plane_bytes + (uint32_t)jx * tap_bytes + (uint32_t)co * kRB + (uint32_t)half * 16u
                               : kDmaOOB;
        bd_lds[i] = (uint32_t)((jx * PL + pt) * BN * kRB + q * 1024);
      }
      auto issue_b = [&](int it, int stage) {
        const int c = it / 3, jy = it - 3 * c;
        const uint32_t soff = (uint32_t)(jy * 3) * tap_bytes + (uint32_t)c * (uint32_t)p.Cd * kRB;
        const uint32_t S = ldsB + stage * kBStage;
#pragma unroll
        for (int i = 0; i < PERB; ++i) {
          uint32_t keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(bd_voff[i]), "s"(S + bd_lds[i]), "s"(rs_b), "s"(soff) : "memory");
        }
      };
      issue_b(0, 0);
      if (niter > 1) issue_b(1, 1);
      load_a(0);
      store_a(0);
      if (nchunk > 1) load_a(1);
      wait_vmcnt<0>();
      ring_barrier();
      int st2 = 2;   // stage of iteration it + 2
      for (int it = 0; it < niter; ++it) {
        const int c = it / 3, jy = it - 3 * c;
        // (halo first: its loads were issued an iteration or more ago; the wait the compiler puts in front of their use
        // then finds the previous iteration's DMA landed too, not one issued a moment ago)
        if (jy == 0 && c + 1 < nchunk) store_a((c + 1) & 1);   // the other halo stage was last read in chunk c-1
        if (jy == 1 && c + 2 < nchunk) load_a(c + 2);
        if (it + 2 < niter) {
          issue_b(it + 2, st2);   // last read in iteration it - 1
          wait_vmcnt<PERB>();     // everything older than these PERB instructions: iteration it + 1's tiles, the halo loads
        } else {
          wait_vmcnt<0>();
        }
        st2 = st2 == 2 ? 0 : st2 + 1;
        ring_barrier();
      }
      return;
    }
    load_a(0);
    load_b(0);
    store_a(0);
    store_b(0);
    if (nchunk > 1) load_a(1);
    if (niter > 1) load_b(1);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
      const int c = it / 3, jy = it - 3 * c;
      if (it + 1 < niter) {
        store_b((it + 1) & 1);
        if (it + 2 < niter) load_b(it + 2);
      }
      if (jy == 0 && c + 1 < nchunk) store_a((c + 1) & 1);   // the other halo stage was last read in chunk c-1
      if (jy == 1 && c + 2 < nchunk) load_a(c + 2);
      __syncthreads();
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

  int hb[MB];   // halo slot of the lane's pixel for tap offset (0, 0)
#pragma unroll
  for (int a = 0; a < MB; ++a) {
    const int m = wm * WMR + a * 32 + li;
    hb[a] = ((m >> 4) + 1) * kHP + (m & 15) + 1;
  }
  int fb[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) fb[b] = half_off(wn * WN + b * 32 + li, lh);

  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    const int c = it / 3, jy = it - 3 * c;
    const unsigned char* A = Abase + (c & 1) * kAStage;
    const unsigned char* B = Bbase + (BDMA ? it % 3 : (it & 1)) * kBStage;
    const int dy = p.oy0 + jy * p.oys;
#pragma unroll
    for (int jx = 0; jx < 3; ++jx) {
      const int dx = p.ox0 + jx * p.oxs;
      bf16x8 fa[MB][3], fbv[NB][3];
#pragma unroll
      for (int a = 0; a < MB; ++a) {
        const int hr = hb[a] + dy * kHP + dx;
        const int off = half_off(hr, lh);
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) fa[a][pt] = *reinterpret_cast<const bf16x8*>(A + pt * kHSlots * kRB + off);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int pt = 0; pt < NP; ++pt)
          fbv[b][pt] = *reinterpret_cast<const bf16x8*>(B + (jx * PL + pt) * BN * kRB + fb[b]);
#pragma unroll
      for (int t6 = 0; t6 < X3Prod<NP>::N; ++t6)
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            acc[a][b] = mfma_np<NP>(fbv[b][x3_pb(NP, t6)], fa[a][x3_pa(NP, t6)], acc[a][b]);
    }
    __syncthreads();
  }

  if constexpr (NP == 2) igemm_scale_acc<MB, NB>(acc, op_scale(act_absmax(p.a_scale)).s * op_scale(*p.w_scale).s);
  if (p.bn_part) {
    // BatchNorm statistics of the output from the epilogue: the loop ended on a barrier of all eight waves and the
    // staging waves have nothing left to write, so the LDS is free to park the tile (igemm_common.hpp)
    BnLaneStat st;
    bn_stat_init(st);
    float* scratch = reinterpret_cast<float*>(smem3) + (wm * 2 + wn) * 32 * (WN + 4);
    float* xch = reinterpret_cast<float*>(smem3) + MW * 32 * (WN + 4);
#pragma unroll
    for (int a = 0; a < MB; ++a) {
      const int m = wm * WMR + a * 32 + li;
      const int gy = Y0 + (m >> 4), gx = X0 + (m & 15);
      // (a patch may hang over the bottom / right edge of the map: those rows are computed — from zero halo — and dropped)
      const size_t roff = (gy < p.Hm && gx < p.Wm)
          ? (((size_t)n * p.Hd + (size_t)(gy * p.dsh + p.doy)) * p.Wd + (size_t)(gx * p.dsw + p.dox)) * p.Cd : ~(size_t)0;
      igemm_store_rows_stats<NB, WN>(p, acc[a], roff, n0, wn, li, lh, scratch, st);
    }
    bn_part_write<WN, MWM, 2>(p, st, (n * tiles_y + ty) * tiles_x + tx, n0, wm, wn, lh * 32 + li, xch);
    return;
  }
  AmaxAcc amax_l{0u, p.out_amax != nullptr};
#pragma unroll
  for (int a = 0; a < MB; ++a) {
    const int m = wm * WMR + a * 32 + li;
    const int gy = Y0 + (m >> 4), gx = X0 + (m & 15);
    if (gy >= p.Hm || gx >= p.Wm) continue;   // outside the map (ragged last patch row / column)
    const size_t roff = (((size_t)n * p.Hd + (size_t)(gy * p.dsh + p.doy)) * p.Wd + (size_t)(gx * p.dsw + p.dox)) * p.Cd;
    igemm_store_rows<NB, WN>(p, acc[a], roff, n0, wn, lh, amax_l);
  }
  if (p.out_amax) amax_commit(p.out_amax, amax_l.m);
}

// does the halo kernel take this launch?  (also decides the layout evk_conv2d_split_weight produces)
bool conv3x3_halo_applies(const IGemmArgs& a) {
  static const int on = getenv("EVK_X3_HALO") ? atoi(getenv("EVK_X3_HALO")) : 1;
  if (!on) return false;
  if (a.kh != 3 || a.kw != 3 || a.ash != 1 || a.asw != 1) return false;
  if (!((a.oys == 1 || a.oys == -1) && a.oy0 == -a.oys && (a.oxs == 1 || a.oxs == -1) && a.ox0 == -a.oxs)) return false;
  if (a.Hm != a.Hs || a.Wm != a.Ws) return false;
  // patches of 8 (or 16) x 16 output pixels; the last row / column of patches may hang over the edge of the map (the halo
  // loads zeros there, the epilogue drops those rows) as long as at least 3/4 of the patch grid is map
  // (round 4: H % 8 == 0 and W % 16 == 0 were required until then — a 616 x 344 scene, or the stride-4 map of a 416-wide
  // tile, fell back to the implicit-GEMM kernels)
  {
    const long long cover = (long long)ceil_div(a.Hm, 8) * 8 * ceil_div(a.Wm, kPW) * kPW;
    if (a.Hm < 4 || a.Wm < 8 || 4LL * a.Hm * a.Wm < 3 * cover) return false;
  }
  // (reduction channels: whole 16-channel chunks, or a partial last one as long as three quarters of the chunks' slots are
  // channels — Cin = 200 = 12.5 chunks; pairs of channels are split together, so Cs must be even: % 8 keeps the 16-byte loads)
  if ((a.Cs % 8) != 0 || 4 * a.Cs < 3 * ceil_div(a.Cs, kCh) * kCh || a.Cd < 64 || (!a.dense_dst && (a.dsh != 1 || a.dsw != 1))) return false;
  // enough workgroups for the 256 CUs, if necessary with the 64-wide N tile
  // (EVK_X3_HALO_MIN_WG=0 makes the choice independent of the batch size: tests/test_linearity_pinned_gpu.py pins the
  // accumulation order — chunk-major here, tap-major in the implicit-GEMM kernels — for a batch and its halves)
  static const long long min_wg = getenv("EVK_X3_HALO_MIN_WG") ? atoll(getenv("EVK_X3_HALO_MIN_WG")) : 256;
  const long long patches = (long long)a.N * ceil_div(a.Hm, 8) * ceil_div(a.Wm, kPW);
  return patches * ceil_div(a.Cd, 64) >= min_wg;
}

// the same decision from a convolution descriptor (forward, or stride-1 data gradient)
bool conv_desc_uses_halo(const evk_conv_desc* d, int for_dgrad) {
  if (d->stride_h != 1 || d->stride_w != 1 || d->dil_h != 1 || d->dil_w != 1 || d->pad_h != 1 || d->pad_w != 1) return false;
  IGemmArgs a{};
  a.N = d->N; a.kh = d->kh; a.kw = d->kw; a.ash = 1; a.asw = 1; a.dense_dst = 1; a.dsh = 1; a.dsw = 1;
  if (!for_dgrad) {
    a.Hs = d->H; a.Ws = d->W; a.Cs = d->Cin; a.Hm = d->Ho; a.Wm = d->Wo; a.Cd = d->Cout;
    a.oy0 = -1; a.oys = 1; a.ox0 = -1; a.oxs = 1;
  } else {
    a.Hs = d->Ho; a.Ws = d->Wo; a.Cs = d->Cout; a.Hm = d->H; a.Wm = d->W; a.Cd = d->Cin;
    a.oy0 = 1; a.oys = -1; a.ox0 = 1; a.oxs = -1;
  }
  return conv3x3_halo_applies(a);
}

template <int BN, int PH, int NPX, bool WDMA = false, int MW = 4>
static int launch_halo_np(IGemmArgs& a, hipStream_t stream) {
  constexpr int NP = X3Mode<NPX>::NP;
  a.tiles_n = ceil_div(a.Cd, BN);
  const int tiles_y = ceil_div(a.Hm, PH), tiles_x = ceil_div(a.Wm, kPW);
  a.tiles_m = a.N * tiles_y * tiles_x;
  bn_stats_setup(a, PH * kPW, BN, 2, a.tiles_m);   // two row waves per patch; the ring (>= 100 KB) is the scratch
  constexpr int PL = NP == 2 ? 2 : 3;
  const size_t lds = (size_t)2 * (PL * HaloGeom<PH, PL>::kHSlots * kRB) + (size_t)(WDMA ? 3 : 2) * (3 * PL * BN * kRB);
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_x3_kernel<BN, PH, NPX, WDMA, MW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const long long nwg = (long long)a.tiles_m * a.tiles_n;
  hipLaunchKernelGGL((conv3x3_halo_x3_kernel<BN, PH, NPX, WDMA, MW>), dim3((unsigned)nwg), dim3(256 + 64 * MW), lds, stream, a, tiles_y, tiles_x);
  return check_launch("conv3x3_halo_x3");
}

template <int BN, int PH, int MW = 4>
static int launch_halo(IGemmArgs& a, hipStream_t stream) {
  if constexpr (MW == 8) {   // (the eight-matrix-wave form exists for the f16x2 arithmetic with DMA-fed weights)
    if (a.planes == 2)
      return a.a_packed ? launch_halo_np<BN, PH, 4, true, 8>(a, stream) : launch_halo_np<BN, PH, 2, true, 8>(a, stream);
  }
  if (a.planes == 1) return launch_halo_np<BN, PH, 1>(a, stream);
  if (a.planes == 2) {
    // (the weight tiles by DMA; through registers as in the other arithmetics: 452.4 vs 465.0 tiles/s)
    return a.a_packed ? launch_halo_np<BN, PH, 4, true>(a, stream) : launch_halo_np<BN, PH, 2, true>(a, stream);
  }
  return launch_halo_np<BN, PH, 3>(a, stream);
}

int launch_conv3x3_halo(IGemmArgs& a, hipStream_t stream) {
  if (!conv3x3_halo_applies(a)) return 1;
  static const bool tune = getenv("EVK_TUNE") != nullptr;
  if (tune) {   // tools/autotune_convs.py
    const char* f = getenv("EVK_X3_HALO_FORCE");
    if (f && *f) {
      if (!strcmp(f, "h64x8")) return launch_halo<64, 8>(a, stream);
      if (!strcmp(f, "m64x8")) return launch_halo<64, 8, 8>(a, stream);
      if (!strcmp(f, "h64x16")) return launch_halo<64, 16>(a, stream);
      if (!strcmp(f, "m64x16")) return launch_halo<64, 16, 8>(a, stream);
      if (!strcmp(f, "h128x8") && a.Cd > 64) return launch_halo<128, 8>(a, stream);
      if (!strcmp(f, "h128x16") && a.Cd > 64) return launch_halo<128, 16>(a, stream);
      if (!strcmp(f, "m128x8") && a.Cd > 64) return launch_halo<128, 8, 8>(a, stream);
      if (!strcmp(f, "m128x16") && a.Cd > 64) return launch_halo<128, 16, 8>(a, stream);
    }
  }
  // workgroup counts from which the 128-wide tile / the 16-row patch is taken (swept in the step in round 5, where the chip is
  // shared with the side stream: 128 / 384 / 768 all within 0.1 % of 256)
  constexpr long long min128 = 256, mintall = 256;
  if (a.Cd <= 64 || (long long)a.N * ceil_div(a.Hm, 8) * ceil_div(a.Wm, kPW) * ceil_div(a.Cd, 128) < min128)
    return launch_halo<64, 8>(a, stream);   // small maps (16^2 .. 32^2): 64-wide tiles keep every CU busy
  // 16 x 16 patches (256 GEMM rows) halve the weight bytes per MFMA, the larger share of the staging traffic now;
  // taken when they still fill the chip.  With the weights fed by DMA (f16x2) the staging waves no longer hold the matrix
  // waves back, and eight matrix waves (two per SIMD) are 1-7 % ahead of four on every 128-wide shape
  // (tools/autotune_convs.py: 777 -> 763 us on 3x3x256 @128^2, 61 -> 57 on 3x3x128 @64^2, 59-62 -> 58 on 3x3x256 @32^2).
  const bool wide8 = a.planes == 2;
  // (16-row patches unless they would add a mostly empty last patch row: H % 16 in 1..8 is served better by 8-row patches)
  const bool tall_fits = (a.Hm % 16) == 0 || (a.Hm % 16) > 8;
  if (tall_fits && (long long)a.N * ceil_div(a.Hm, 16) * ceil_div(a.Wm, kPW) * ceil_div(a.Cd, 128) >= mintall)
    return wide8 ? launch_halo<128, 16, 8>(a, stream) : launch_halo<128, 16>(a, stream);
  return wide8 ? launch_halo<128, 8, 8>(a, stream) : launch_halo<128, 8>(a, stream);
}

// planes for the halo kernel: out[pt][tap][chunk][row][16] bf16; tap = jy*3 + jx in the kernel's (affine) tap
// order, which is (ky, kx) for the forward and for the stride-1 data gradient alike (the sign of oys flips the
// direction, not the index).  rows = Cout (forward: w[row][ky][kx][ci]) or Cin (data gradient: w[co][ky][kx][row]).
__global__ void split_weight_halo_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int Cout, int Cin,
                                         int for_dgrad, const uint32_t* __restrict__ wscale) {
  static_assert(kHaloCh == kCh, "split_weight.hpp chunk width");
  split_halo_body(w, out, Cout, Cin, for_dgrad, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
                  (size_t)gridDim.x * blockDim.x, wscale);
}

int launch_split_weight_halo(const float* w, uint16_t* out, int Cout, int Cin, int for_dgrad, hipStream_t st,
                             const uint32_t* wscale) {
  const int rows = for_dgrad ? Cin : Cout, K = for_dgrad ? Cout : Cin;
  const size_t total = (size_t)9 * ((K + kCh - 1) / kCh) * rows * (kCh / 2);
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(split_weight_halo_kernel, dim3(blocks), dim3(256), 0, st, w, out, Cout, Cin, for_dgrad, wscale);
  return check_launch("split_weight_halo");
}

}  // namespace evk
