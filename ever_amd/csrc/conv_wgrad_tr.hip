// Weight gradient of the f16x2 arithmetic from PLANAR operands: global -> LDS by DMA, fragments by transposing LDS reads.
//
//   dw[co][k] = sum_m dy[m][co] * im2col(x)[m][k]        m = (n, oy, ox),  k = (ky, kx, ci)
//
// The register-staged kernels (conv_wgrad_x3 / x3ws) spend their staging waves on what the two operands' layout forces:
// both are pixel-major in HBM while the MFMA wants, per lane, eight reduction(pixel)-consecutive values of one channel, so
// every element is loaded to a VGPR, (split,) transposed with byte permutes and written as pixel runs (~400 VALU and 18
// ds_write_b128 per staging thread and 32-pixel step; matrix pipe 31-58 % busy).  Here
//  * the operands arrive as two fp16 PLANES of value / s each (evk_pack_planar_f16x2 or a producer's planar mode):
//    H[M][C] then L[M][C] in one allocation of the fp32 tensor's size — the same 22-bit (h, l) pair as the packed word;
//  * a step's tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (16 B per lane, lane-linear LDS side): one
//    instruction moves 8 pixel rows x 128 B (64 channels) — whole cache lines (64-byte rows, the first form of this
//    kernel, fetched every line twice: DMA alone 495 us on 3x3x256 @128^2 against 371 us of MFMA issue).  Pixels
//    outside the image / the chunk get an out-of-range offset: the DMA writes zeros;
//  * fragments come from ds_read_b64_tr_b16: within a 16-lane group, lane 4j+q supplies the address of channels
//    4q..4q+3 of pixel j and lane i receives pixels 0..3 of channel i (tools/probes/tr_read.hip) — the transpose the
//    MFMA operand needs, for nothing.  Rows are 128 B, so the four pixel rows a 32-lane service group touches would
//    fall on two 64-byte windows of the 256-byte bank row twice each; the 64-byte halves of rows with pixel bit 1 set
//    are therefore swapped — on the SOURCE side of the DMA (lane -> global chunk), the LDS side being lane-linear — and
//    the reads apply the same XOR (a per-lane constant): conflict-free for any whole-pixel shift;
//  * no wave specialisation: 8 waves, two per SIMD, each issues its sixth of the step's DMA and its MFMAs, so a SIMD's
//    two waves cover each other's LDS latencies (one matrix wave per SIMD ran at 57 % of the MFMA rate with the loads
//    ablated).  Ring of three stages, DMA two steps ahead, counted vmcnt, raw s_barrier, one barrier per 32 pixels.
//  * NT = 9 (3x3, stride 1, padding 1, W % 32 == 0): ONE halo image (3 rows x 34 pixels x 64 channels) serves all nine
//    taps — a tap is a whole-row offset of the transposing read — so the im2col operand is fetched 3.2 rows instead
//    of 9 per step, and a step carries 27 MFMAs per k-half and wave instead of 12 (tile 128 co x 9 taps x 64 ci).
//    NT = 1: tile 128 co x 256 k columns as four independent 64-channel segments (any kernel / stride / dilation).
#include "wgrad_common.hpp"
#include "x3_common.hpp"
#include "lds_dma.hpp"
#include <stdlib.h>

#ifndef EVK_WG_ABL
#define EVK_WG_ABL 0   // timing ablations (tools/build_variant.sh -DEVK_WG_ABL=n; wrong results): 1 no loads, 2 no split / LDS writes,
#endif                 // 4 no fragment reads / MFMAs, 8 no stores
namespace evk {
constexpr int kWgAbl = EVK_WG_ABL;


namespace {

constexpr int kRow = 128;                 // bytes of one pixel row of a segment: 64 channels fp16
constexpr int kOct = 8 * kRow;            // one DMA instruction: 8 pixel rows
constexpr int kSeg = 32 * kRow;           // one segment of a step: [32 pixels][64 channels]
constexpr int kHaloPx = 40;               // halo slots per image row (34 used: ox0 - 1 .. ox0 + 32)
constexpr int kHaloRow = kHaloPx * kRow;
constexpr uint32_t kOOBtr = kDmaOOB;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s16x4 tr_read(uint32_t lds_byte) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)lds_byte);
}
// pixels k..k+3 and k+4..k+7 of this lane's channel; `base` is a per-step VGPR, `off` a compile-time constant that goes
// into the instruction's offset field
__device__ __forceinline__ f16x8 frag8(uint32_t base, int off) {
  const s16x4 lo = tr_read(base + off), hi = tr_read(base + off + 4 * kRow);
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, v);
}
}  // namespace

template <int NT>
struct TrGeom {
  static constexpr int BM = 128;
  static constexpr int kAPlane = 2 * kSeg;                              // dy: two 64-channel segments
  static constexpr int kBPlane = NT == 9 ? 3 * kHaloRow : 4 * kSeg;     // halo image / four segments
  static constexpr int kStage = 2 * (kAPlane + kBPlane);
  static constexpr int NB = NT == 9 ? 9 : 4;                           // 32-column blocks per wave
  static constexpr int kADma = 2 * 2 * 4;                               // planes x segments x octets
  static constexpr int kBDma = NT == 9 ? 2 * 3 * 5 : 2 * 4 * 4;
  static constexpr int PER = 6;                                         // DMA instructions per wave and step
  static_assert(kADma + kBDma <= 8 * PER, "eight waves issue the step's DMA");
};

template <int NT>
__global__ __launch_bounds__(512) void conv_wgrad_tr_kernel(const WGradArgs p) {
  using G = TrGeom<NT>;
  constexpr int NST = 3, NB = G::NB, PER = G::PER;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_tr[];

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tiles_co * p.tiles_k;
  const int z = bid / ntile;
  const int tile = bid - z * ntile;
  const int tile_k = tile % p.tiles_k;
  const int tile_co = tile / p.tiles_k;
  const int co0 = tile_co * G::BM;
  const int k0 = tile_k * (NT == 9 ? 64 : 256);   // NT = 9: first input channel of the tile; NT = 1: first k column
  const int pbeg = z * p.chunk;
  const int pend = min(p.M, pbeg + p.chunk);
  const int nk = (pend - pbeg + BKP - 1) / BKP;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

  // ------------------------------------------------------------------ this wave's share of a step's DMA
  // Everything that does not change from step to step is decoded here, once: instruction `id` of the step (wave-uniform)
  // -> what it loads and where it lands; per step only the pixel position moves, and it is wave-uniform too (a step's 32
  // pixels start at a multiple of 8, W % 8 == 0: an octet lies in one image row), so the per-lane work of an instruction
  // is an add, two compares and a select.
  const int drow = lane >> 3;                                    // pixel row inside the octet
  const int dch = ((lane & 7) ^ (((drow >> 1) & 1) << 2)) * 16;  // source chunk landing on LDS chunk (lane & 7)
  const uint32_t x_plane = (uint32_t)p.N * p.H * p.W * p.Cin * 2u, dy_plane = (uint32_t)p.M * p.Cout * 2u;
  const i32x4 rs_x = make_rsrc(p.x, 2u * x_plane), rs_dy = make_rsrc(p.dy, 2u * dy_plane);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_tr;
  // slots 0, 1 of every wave are dy instructions, slots 2..5 im2col ones (16 + 32 over the eight waves; the halo form has
  // 30: the last wave repeats two) — the kind of a slot is a compile-time fact, no branch per instruction
  int d_dst[PER], d_oc[PER], d_a[PER], d_b[PER];   // LDS offset in the stage; octet; two kind-specific scalars
  uint32_t d_base[PER];                             // scalar part of the byte offset that does not move
  bool d_ok[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if (i < 2) {                                                 // dy: id = (plane, segment, octet)
      const int id = wave * 2 + i;
      const int pl = id >> 3, sg = (id >> 2) & 1;
      d_oc[i] = id & 3;
      const int co = co0 + 64 * sg;
      d_ok[i] = co < p.Cout;
      d_base[i] = (uint32_t)co * 2u + (pl ? dy_plane : 0u);
      d_dst[i] = pl * G::kAPlane + sg * kSeg + d_oc[i] * kOct;
      d_a[i] = d_b[i] = 0;
    } else if (NT == 9) {                                        // halo: (plane, image row, octet of slots)
      int j = wave * 4 + i - 2;
      if (j >= 30) j -= 2;
      const int pl = j / 15, r3 = (j - pl * 15) / 5;
      d_oc[i] = j - pl * 15 - r3 * 5;
      d_ok[i] = true;
      d_a[i] = r3 - 1;                                           // image row offset
      d_b[i] = 0;
      d_base[i] = (uint32_t)k0 * 2u + (pl ? x_plane : 0u);
      d_dst[i] = 2 * G::kAPlane + pl * G::kBPlane + r3 * kHaloRow + d_oc[i] * kOct;
    } else {                                                     // im2col segment: (plane, segment, octet)
      const int j = wave * 4 + i - 2;
      const int pl = j >> 4, sg = (j >> 2) & 3;
      d_oc[i] = j & 3;
      const int kcol = k0 + 64 * sg;
      d_ok[i] = kcol < p.Ktot;
      const int tap = d_ok[i] ? kcol / p.Cin : 0;
      const int ci = d_ok[i] ? kcol - tap * p.Cin : 0;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      d_a[i] = ky * p.dh - p.ph;
      d_b[i] = kx * p.dw - p.pw;
      d_base[i] = (uint32_t)ci * 2u + (pl ? x_plane : 0u);
      d_dst[i] = 2 * G::kAPlane + pl * G::kBPlane + sg * kSeg + d_oc[i] * kOct;
    }
  }
  const uint32_t a_lane = (uint32_t)drow * (uint32_t)p.Cout * 2u + (uint32_t)dch;
  const uint32_t b_lane = (uint32_t)drow * (uint32_t)(p.sw * p.Cin) * 2u + (uint32_t)dch;   // (NT = 9: sw = 1)

  auto issue = [&](int kt) {
    const uint32_t S = lds0 + (uint32_t)((kt % NST) * G::kStage);
    if (kWgAbl & 1) return;
    const int m0 = pbeg + kt * BKP;
    // the step's first pixel (wave-uniform).  NT = 9: the whole step lies in this image row; NT = 1: an octet does
    // (a step starts at a multiple of 32 pixels, Wo % 8 == 0 is the launcher's condition)
    const uint32_t mm0 = (uint32_t)min(m0, p.M - 1);
    const uint32_t n0 = fdiv(mm0, p.fd_hw);
    const uint32_t rem0 = mm0 - n0 * p.fd_hw.div;
    const uint32_t oy0 = fdiv(rem0, p.fd_w);
    const int ox0 = (int)(rem0 - oy0 * p.fd_w.div);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int mo = m0 + 8 * d_oc[i];
      if (i < 2) {
        const uint32_t off = (d_ok[i] && mo + drow < pend) ? (uint32_t)mo * (uint32_t)p.Cout * 2u + d_base[i] + a_lane : kOOBtr;
        dma16(rs_dy, S + (uint32_t)d_dst[i], off);
      } else if (NT == 9) {
        const int sy = (int)oy0 + d_a[i], hs = 8 * d_oc[i] + drow, sx = ox0 - 1 + hs;
        const bool rowok = m0 < pend && (unsigned)sy < (unsigned)p.H;                     // scalar
        const uint32_t sb = (uint32_t)((((int)n0 * p.H + sy) * p.W + ox0 - 1 + 8 * d_oc[i]) * p.Cin) * 2u + d_base[i];   // scalar
        const uint32_t off = (rowok && hs < 34 && (unsigned)sx < (unsigned)p.W) ? sb + b_lane : kOOBtr;
        dma16(rs_x, S + (uint32_t)d_dst[i], off);
      } else {
        uint32_t n = n0, oy = oy0;
        int ox = ox0 + 8 * d_oc[i];
        if (ox >= p.Wo) {                                        // the step wrapped into the next image row(s): scalar branch
          const uint32_t mm = (uint32_t)min(mo, p.M - 1);
          n = fdiv(mm, p.fd_hw);
          const uint32_t rem = mm - n * p.fd_hw.div;
          oy = fdiv(rem, p.fd_w);
          ox = (int)(rem - oy * p.fd_w.div);
        }
        const int sy = (int)oy * p.sh + d_a[i], sxb = ox * p.sw + d_b[i];
        const bool rowok = d_ok[i] && (unsigned)sy < (unsigned)p.H;                       // scalar
        const uint32_t sb = (uint32_t)((((int)n * p.H + sy) * p.W + sxb) * p.Cin) * 2u + d_base[i];   // scalar
        const int sx = sxb + drow * p.sw;
        const uint32_t off = (rowok && mo + drow < pend && (unsigned)sx < (unsigned)p.W) ? sb + b_lane : kOOBtr;
        dma16(rs_x, S + (uint32_t)d_dst[i], off);
      }
    }
  };

  // ------------------------------------------------------------------ fragment addressing
  // wave (wm, wn): 32 output channels (dy segment wm >> 1, half wm & 1) x NB column blocks
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int g = lane >> 4, jj = (lane & 15) >> 2, q = lane & 3;
  const int lane_lo = (g & 1) * 32 + q * 8;                      // inside the 64-byte half: channel group and quad
  auto half_off = [&](int hb, int shift) {                       // byte offset of half hb for pixel rows jj + shift (+ 4u)
    return (8 * (g >> 1) + jj + shift) * kRow + ((hb ^ (((jj + shift) >> 1) & 1)) * 64) + lane_lo;
  };
  const int fa_off = (wm >> 1) * kSeg + half_off(wm & 1, 0);
  int fb_off[NT == 9 ? 3 : 4];
  if constexpr (NT == 9) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) fb_off[kx] = 2 * G::kAPlane + half_off(wn, kx);
  } else {
#pragma unroll
    for (int b = 0; b < 4; ++b) fb_off[b] = 2 * G::kAPlane + (2 * wn + (b >> 1)) * kSeg + half_off(b & 1, 0);
  }
  const float out_scale = op_scale(act_absmax(p.x_scale)).s * op_scale(act_absmax(p.dy_scale)).s;

  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  // ------------------------------------------------------------------ ring: step kt lives in slot kt % 3
  if (nk > 0) issue(0);
  if (nk > 1) issue(1);
  if (nk > 1) wait_vmcnt<PER>(); else wait_vmcnt<0>();
  ring_barrier();                                  // step 0 has landed (everybody's share)
  for (int kt = 0; kt < nk; ++kt) {
    // slot (kt + 2) % 3 was read in step kt - 1, which every wave finished before the last barrier.  Waves w and w + 4
    // share a SIMD: the first four issue their DMA share (addresses + 6 instructions, ~1 k cycles of issue) at the top of
    // the step, the other four between the k-halves, so that a SIMD's matrix pipe has one wave's MFMAs meanwhile
    if (kt + 2 < nk && wave < 4) issue(kt + 2);
    const uint32_t S = lds0 + (uint32_t)((kt % NST) * G::kStage);
    if (!(kWgAbl & 4)) {
      // units u = (k-half, group of GS column blocks); the fragments of unit u + 1 are read before the MFMAs of unit u are
      // issued and nothing else moves across (sched_barrier): left alone, the scheduler hoists every read of the step to
      // its top (9 taps x 2 planes x 4 registers beside 144 accumulators: spills)
      constexpr int GS = NT == 9 ? 3 : 4, NG = NB / GS, U = 2 * NG;
      const uint32_t ba = opaque(S + (uint32_t)fa_off);
      uint32_t bb[NT == 9 ? 3 : 4];
#pragma unroll
      for (int e = 0; e < (NT == 9 ? 3 : 4); ++e) bb[e] = opaque(S + (uint32_t)fb_off[e]);
      f16x8 fa[2][2], fb[2][GS][2];
      auto load_unit = [&](int u, int s_) {
        const int kk = u / NG, gq = u % NG;
        const int ko = kk * 16 * kRow;
        if (gq == 0) {
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) fa[kk][pt] = frag8(ba, ko + pt * G::kAPlane);
        }
#pragma unroll
        for (int e = 0; e < GS; ++e)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
            if constexpr (NT == 9) fb[s_][e][pt] = frag8(bb[e], ko + pt * G::kBPlane + gq * kHaloRow);
            else fb[s_][e][pt] = frag8(bb[e], ko + pt * G::kBPlane);
          }
      };
      load_unit(0, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (u == NG && kt + 2 < nk && wave >= 4) issue(kt + 2);
        if (u + 1 < U) load_unit(u + 1, (u + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const int kk = u / NG, gq = u % NG;
        // the three products of a block are dependent MFMAs on one accumulator: a unit's blocks are interleaved product
        // by product (reuse distance = GS)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int e = 0; e < GS; ++e)
            acc[gq * GS + e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk][kHA[t]], fb[u & 1][e][kHB[t]], acc[gq * GS + e], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if ((kWgAbl & 4) && kt + 2 < nk && wave >= 4) issue(kt + 2);
    if (kt + 2 < nk) wait_vmcnt<PER>(); else wait_vmcnt<0>();   // all but the newest step's: step kt + 1 has landed
    ring_barrier();
  }
  if (kWgAbl & 8) {
    if (acc[0][0] == 12345.f) p.out[0] = 0.f;
    return;
  }

  float* out = p.out + (size_t)z * p.Cout * p.Ktot;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (row >= p.Cout) continue;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int col = NT == 9 ? b * p.Cin + k0 + wn * 32 + li : k0 + wn * 128 + b * 32 + li;
      if (col < p.Ktot) out[(size_t)row * p.Ktot + col] = acc[b][r] * out_scale;
    }
  }
}

bool wgrad_tr_applicable(const WGradArgs& a) {
  return a.planes == 2 && a.Cin % 64 == 0 && a.Cout % 64 == 0 && a.Wo % 8 == 0 &&
         (long long)a.N * a.H * a.W * a.Cin * 4 < 0x7fffffffLL && (long long)a.M * a.Cout * 4 < 0x7fffffffLL;
}
// the nine-tap form: 3x3, stride 1, padding 1, no dilation, image rows that are whole steps
bool wgrad_tr_nine_tap(const evk_conv_desc* d) {
  static const int on = getenv("EVK_WG_TR9") ? atoi(getenv("EVK_WG_TR9")) : 1;
  return on && d->kh == 3 && d->kw == 3 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 1 && d->pad_w == 1 &&
         d->dil_h == 1 && d->dil_w == 1 && d->W % 32 == 0 && d->Wo == d->W && d->Ho == d->H;
}

template <int NT>
static int launch_tr(const WGradArgs& b, hipStream_t stream) {
  const size_t lds = (size_t)3 * TrGeom<NT>::kStage;
  static PerDeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_tr_kernel<NT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL((conv_wgrad_tr_kernel<NT>), dim3(b.tiles_co * b.tiles_k * b.splitk), dim3(512), lds, stream, b);
  return check_launch("conv_wgrad_tr");
}

int launch_wgrad_tr(const WGradArgs& a, int nine_tap, hipStream_t stream) {
  WGradArgs b = a;
  return nine_tap ? launch_tr<9>(b, stream) : launch_tr<1>(b, stream);
}

// ---- stand-alone producers of the planar form (the fused ones are the BatchNorm passes' planar modes)
__global__ __launch_bounds__(256) void pack_planar_f16x2_kernel(const float* __restrict__ x, size_t n8,
                                                                const uint32_t* __restrict__ amax, uint32_t* __restrict__ H,
                                                                uint32_t* __restrict__ L) {
  const float inv = op_scale(act_absmax(amax)).inv;
  // (grid-stride: one trip with the one-shot grid; EVK_PACK_PLANAR_WG caps the grid — the pass runs on the side stream beside
  // HBM-bound kernels of the backward chain)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + i * 8), v1 = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
    u32x4 h, l;
    uint32_t a, b;
    split2h(v0.x * inv, v0.y * inv, a, b); h.x = a; l.x = b;
    split2h(v0.z * inv, v0.w * inv, a, b); h.y = a; l.y = b;
    split2h(v1.x * inv, v1.y * inv, a, b); h.z = a; l.z = b;
    split2h(v1.z * inv, v1.w * inv, a, b); h.w = a; l.w = b;
    *reinterpret_cast<u32x4*>(H + i * 4) = h;
    *reinterpret_cast<u32x4*>(L + i * 4) = l;
  }
}
__global__ __launch_bounds__(256) void unpack_planar_f16x2_kernel(const uint32_t* __restrict__ H, const uint32_t* __restrict__ L,
                                                                  size_t n8, const uint32_t* __restrict__ amax,
                                                                  float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const float s = op_scale(act_absmax(amax)).s;
  if (i >= n8) return;
  const u32x4 h = *reinterpret_cast<const u32x4*>(H + i * 4), l = *reinterpret_cast<const u32x4*>(L + i * 4);
  const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
  float o[8];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    o[2 * t] = unpack_hl((hw[t] & 0xffffu) | (lw[t] << 16)) * s;
    o[2 * t + 1] = unpack_hl((hw[t] >> 16) | (lw[t] & 0xffff0000u)) * s;
  }
  *reinterpret_cast<f32x4*>(out + i * 8) = f32x4{o[0], o[1], o[2], o[3]};
  *reinterpret_cast<f32x4*>(out + i * 8 + 4) = f32x4{o[4], o[5], o[6], o[7]};
}

}  // namespace evk

using namespace evk;

extern "C" int evk_pack_planar_f16x2(const float* x, int64_t n, const uint32_t* x_absmax, void* out, void* stream) {
  EVK_REQUIRE(x && x_absmax && out && n > 0 && n % 8 == 0, EVK_E_INVALID, "pack_planar_f16x2: null pointer or n %% 8 != 0");
  const size_t n8 = (size_t)n / 8;
  uint32_t* H = (uint32_t*)out;
  static const long long cap = getenv("EVK_PACK_PLANAR_WG") ? atoll(getenv("EVK_PACK_PLANAR_WG")) : 0;
  size_t grid = (n8 + 255) / 256;
  if (cap > 0 && grid > (size_t)cap) grid = (size_t)cap;
  hipLaunchKernelGGL(pack_planar_f16x2_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, n8,
                     x_absmax, H, H + n / 2);
  return check_launch("pack_planar_f16x2");
}
extern "C" int evk_unpack_planar_f16x2(const void* planar, int64_t n, const uint32_t* x_absmax, float* out, void* stream) {
  EVK_REQUIRE(planar && x_absmax && out && n > 0 && n % 8 == 0, EVK_E_INVALID, "unpack_planar_f16x2: null pointer or n %% 8 != 0");
  const size_t n8 = (size_t)n / 8;
  const uint32_t* H = (const uint32_t*)planar;
  hipLaunchKernelGGL(unpack_planar_f16x2_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, H,
                     H + n / 2, n8, x_absmax, out);
  return check_launch("unpack_planar_f16x2");
}
