// Weight gradient of the f16x2 arithmetic from PLANAR operands: global -> LDS by DMA, fragments by transposing LDS reads.
//
//   dw[co][k] = sum_m dy[m][co] * im2col(x)[m][k]        m = (n, oy, ox),  k = (ky, kx, ci)
//
// The register-staged kernels (conv_wgrad_x3 / x3ws) spend their staging waves on what the two operands' layout forces:
// both are pixel-major in HBM while the MFMA wants, per lane, eight reduction(pixel)-consecutive values of one channel, so
// every element is loaded to a VGPR, (split,) transposed with byte permutes and written as pixel runs (~400 VALU and 18
// ds_write_b128 per staging thread and 32-pixel step; matrix pipe 31-58 % busy).  Here
//  * the operands arrive as two fp16 PLANES of value / s each (evk_pack_planar_f16x2 or a producer's EVK_*_PLANAR mode):
//    H[M][C] then L[M][C] in one allocation of the fp32 tensor's size — the same 22-bit (h, l) pair as the packed word;
//  * a step's tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (16 B per lane, lane-linear): the LDS image is
//    [plane][32-channel segment][32 pixels][64 B], i.e. exactly the memory order of a 64-byte piece of 16 consecutive
//    pixel rows per instruction; pixels outside the image / the chunk get an out-of-range offset (the DMA writes zeros);
//    four loader waves issue 12 instructions each per step and do nothing else (~80 VALU for the addresses);
//  * the matrix waves read their fragments with ds_read_b64_tr_b16: within a 16-lane group, lane 4j+q supplies the
//    address of channels 4q..4q+3 of pixel j and lane i receives pixels 0..3 of channel i (tools/probes/tr_read.hip) —
//    the transpose the MFMA operand needs, for nothing; a group's four pixel rows are 256 contiguous bytes: no conflict.
// Ring of three 48 KB stages, DMA two steps ahead (counted vmcnt, raw s_barrier: a __syncthreads would drain the DMA),
// one barrier per 32-pixel step.  Tile 128 (Cout) x 256 (k), split over pixel chunks like the other kernels.
#include "wgrad_common.hpp"
#include "x3_common.hpp"
#include <stdlib.h>

namespace evk {

namespace {

constexpr int kSeg = 32 * 64;            // one segment of a step: [32 pixels][32 channels fp16]
constexpr uint32_t kOOBtr = 0x80000000u;  // beyond every buffer's num_records: the DMA writes zeros

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, 0, 0, 0);
}
__device__ __forceinline__ s16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
__device__ __forceinline__ f16x8 frag8(const unsigned char* p) {   // pixels k..k+3 and k+4..k+7 of this lane's channel
  const s16x4 lo = tr_read(p), hi = tr_read(p + 4 * 64);
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, v);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// the loader waves' barrier: no fence (a __syncthreads would wait for vmcnt(0), i.e. drain the DMA that is meant to stay in
// flight across it); the "memory" clobber keeps the compiler from moving the DMA issue across it
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

}  // namespace

template <int BM, int BN>
__global__ __launch_bounds__(512) void conv_wgrad_tr_kernel(const WGradArgs p) {
  constexpr int SA = BM / 32, SB = BN / 32;          // 32-channel segments per operand tile
  constexpr int kAPlane = SA * kSeg, kBPlane = SB * kSeg;
  constexpr int kStage = 2 * (kAPlane + kBPlane);
  constexpr int NST = 3;
  constexpr int WM = BM / 2, WN = BN / 2, MB = WM / 32, NB = WN / 32;
  constexpr int XS = SB / 4;                         // im2col segments per loader wave
  static_assert(SA == 4 && SB % 4 == 0, "four loader waves: one dy segment and SB / 4 im2col segments each");
  constexpr int PER = 2 * 2 * (1 + XS);              // DMA instructions per loader wave and step

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_tr[];

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
    // ------------------------------------------------------------------ loader waves
    const int w = (tid - 256) >> 6, lane = tid & 63;
    const int pxl = lane >> 2, ch16 = (lane & 3) * 16;
    const uint32_t x_plane = (uint32_t)p.N * p.H * p.W * p.Cin * 2u, dy_plane = (uint32_t)p.M * p.Cout * 2u;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(2u * x_plane), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)(2u * dy_plane), 0x00020000);
    // this wave's dy segment and im2col segments (wave-uniform, constant over the steps)
    const int a_co = co0 + 32 * w;
    const bool a_ok = a_co < p.Cout;
    int b_offy[XS], b_offx[XS], b_ci[XS];
    bool b_ok[XS];
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      const int kcol = k0 + 32 * (XS * w + s);
      b_ok[s] = kcol < p.Ktot;
      const int tap = b_ok[s] ? kcol / p.Cin : 0;
      b_ci[s] = b_ok[s] ? kcol - tap * p.Cin : 0;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      b_offy[s] = ky * p.dh - p.ph;
      b_offx[s] = kx * p.dw - p.pw;
    }
    auto issue = [&](int kt) {
      unsigned char* S = smem_tr + (kt % NST) * kStage;
      if (p.dbg & 1) return;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = pbeg + kt * BKP + 16 * h + pxl;
        const bool mok = m < pend;
        const uint32_t mm = (uint32_t)min(m, p.M - 1);
        // dy: the plain [M][Cout] matrix
        const uint32_t aoff = (mok && a_ok) ? (mm * (uint32_t)p.Cout + (uint32_t)a_co) * 2u + (uint32_t)ch16 : kOOBtr;
        unsigned char* Ad = S + w * kSeg + h * 1024;
        dma16(rs_dy, Ad, aoff);
        dma16(rs_dy, Ad + kAPlane, aoff == kOOBtr ? kOOBtr : aoff + dy_plane);
        const uint32_t n = fdiv(mm, p.fd_hw);
        const uint32_t rem = mm - n * p.fd_hw.div;
        const uint32_t oy = fdiv(rem, p.fd_w);
        const int ox = (int)(rem - oy * p.fd_w.div);
#pragma unroll
        for (int s = 0; s < XS; ++s) {
          const int sy = (int)oy * p.sh + b_offy[s], sx = ox * p.sw + b_offx[s];
          const bool ok = mok && b_ok[s] && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
          const uint32_t boff = ok ? (uint32_t)((((int)n * p.H + sy) * p.W + sx) * p.Cin + b_ci[s]) * 2u + (uint32_t)ch16 : kOOBtr;
          unsigned char* Bd = S + 2 * kAPlane + (XS * w + s) * kSeg + h * 1024;
          dma16(rs_x, Bd, boff);
          dma16(rs_x, Bd + kBPlane, ok ? boff + x_plane : kOOBtr);
        }
      }
    };
    // step kt lives in ring slot kt % 3; the DMA runs two steps ahead of the matrix waves
    if (nk > 0) issue(0);
    if (nk > 1) issue(1);
    if (nk > 1) wait_vmcnt<PER>(); else wait_vmcnt<0>();
    raw_barrier();              // step 0 has landed
    for (int kt = 0; kt < nk; ++kt) {
      // slot (kt + 2) % 3 was read in step kt - 1, which every matrix wave finished before the last barrier
      if (kt + 2 < nk) {
        issue(kt + 2);
        wait_vmcnt<PER>();                      // all but the newest step's: step kt + 1 has landed
      } else {
        wait_vmcnt<0>();
      }
      raw_barrier();
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

  // transposing read: lane l of 16-lane group g = l >> 4 addresses pixel row (8 * (g >> 1) + ((l & 15) >> 2)) of its
  // k-half, channels 16 * (g & 1) + 4 * (l & 3) ... + 3, and receives pixels 8 * (g >> 1) + 0..3 of channel
  // 16 * (g & 1) + (l & 15) = MFMA row / column (l & 31), k = 8 * (l >> 5) + 0..3 (second read: + 4 pixel rows)
  const int g = lane >> 4;
  const int fr_off = (8 * (g >> 1) + ((lane & 15) >> 2)) * 64 + (g & 1) * 32 + (lane & 3) * 8;
  const float out_scale = op_scale(act_absmax(p.x_scale)).s * op_scale(act_absmax(p.dy_scale)).s;

  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* S = smem_tr + (kt % NST) * kStage;
    if (!(p.dbg & 4)) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        f16x8 fa[MB][2], fb[NB][2];
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt)
            fa[a][pt] = frag8(S + pt * kAPlane + (wm * MB + a) * kSeg + kk * 1024 + fr_off);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt)
            fb[b][pt] = frag8(S + 2 * kAPlane + pt * kBPlane + (wn * NB + b) * kSeg + kk * 1024 + fr_off);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a][kHA[t]], fb[b][kHB[t]], acc[a][b], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (p.dbg & 8) {
    if (acc[0][0][0] == 12345.f) p.out[0] = 0.f;
    return;
  }

  float* out = p.out + (size_t)z * p.Cout * p.Ktot;
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = co0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= p.Cout) continue;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int col = k0 + wn * WN + b * 32 + li;
        if (col < p.Ktot) out[(size_t)row * p.Ktot + col] = acc[a][b][r] * out_scale;
      }
    }
}

bool wgrad_tr_applicable(const WGradArgs& a) {
  return a.planes == 2 && a.Cin % 32 == 0 && a.Cout % 32 == 0 &&
         (long long)a.N * a.H * a.W * a.Cin * 4 < 0x7fffffffLL && (long long)a.M * a.Cout * 4 < 0x7fffffffLL;
}

int launch_wgrad_tr(const WGradArgs& a, hipStream_t stream) {
  constexpr int BM = 128, BN = 256;
  WGradArgs b = a;
  static const int dbg = getenv("EVK_WG_DBG") ? atoi(getenv("EVK_WG_DBG")) : 0;
  b.dbg = dbg;
  const size_t lds = (size_t)3 * 2 * ((BM + BN) / 32) * kSeg;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_tr_kernel<BM, BN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_wgrad_tr_kernel<BM, BN>), dim3(b.tiles_co * b.tiles_k * b.splitk), dim3(512), lds, stream, b);
  return check_launch("conv_wgrad_tr");
}

// ---- stand-alone producers of the planar form (the fused ones are the BatchNorm passes' EVK_BN_PLANAR_* modes)
__global__ __launch_bounds__(256) void pack_planar_f16x2_kernel(const float* __restrict__ x, size_t n8,
                                                                const uint32_t* __restrict__ amax, uint32_t* __restrict__ H,
                                                                uint32_t* __restrict__ L) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const float inv = op_scale(act_absmax(amax)).inv;
  if (i >= n8) return;
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
  hipLaunchKernelGGL(pack_planar_f16x2_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n8,
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
