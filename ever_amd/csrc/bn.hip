// BatchNorm2d (+ residual add, + ReLU) forward / backward on NHWC fp32 for gfx950.
// Replaces aten::batch_norm / native_batch_norm_backward, aten::add_, aten::relu_ at reference
// ever/module/_resnets.py:95-112 (bn1..bn3, `out += identity`, relu), fs_relation.py:39-53,
// fpn.py:163-167.  HBM-bound: every kernel streams [rows][C] with 16-byte accesses, a workgroup's
// row range is one contiguous span.  Statistics are reduced in two stages (fp32 partials per
// workgroup, fp64 finalisation) so results do not depend on the launch grid.
#include "x3_common.hpp"
#include <stdlib.h>


namespace evk {

// Traversal direction of the streaming passes (round 5).  The apply passes, forward and backward, walk the map from
// its END: the pass in front of them (the convolution that wrote z; the reduce pass that has just read g and z) finished there,
// so the lines most recently touched come first, and what the pass writes is in turn met head-first by the next convolution.
// Same arithmetic, same bits.  Six interleaved rounds on one box: 538.49 -> 539.65 tiles/s (+0.22 %, ahead in every round);
// the forward alone +0.13 %, the reduce pass reversed as well +0.19 % (tools/ab_libs.sh, profiles/r05_experiments/ab_bn_rev*.txt).
// Bits: 1 bn_apply, 2 bn_bwd_apply, 4 bn_bwd_partial.
__device__ __forceinline__ unsigned bn_blk() { return gridDim.x - 1u - blockIdx.x; }

// Channel group of a FINALISATION workgroup (round 6).  Consecutive channel groups read neighbouring 4-byte .. 32-byte pieces of
// the same 64-byte lines of the partial records, and the hardware deals consecutive workgroups to the eight XCDs round-robin:
// with group = blockIdx every line of the records was fetched by up to eight L2s (bn_parts_final_kernel<1, 256>: 53.8 MB of HBM
// fetches for 6.3 MB of records, profiles/r06_experiments/traffic_by_kernel.txt).  xcd_remap hands the workgroups of ONE XCD
// consecutive groups instead.  Which workgroup finalises a channel changes, what it computes does not: the same bits.
__device__ __forceinline__ int bn_fin_group() { return xcd_remap((int)blockIdx.x, (int)gridDim.x); }

// non-temporal loads of the stem's 268 MB map in its fused BatchNorm + pool passes (experiment: a plain read of more than 256 MB
// behind a plain-store writer streams at 4.1 TB/s, with the hint at 6.8 — tools/probes/mall_direction.hip).
// Bits: 1 backward reduce pass, 2 forward, 4 backward apply pass
// kernel times with the hint (us): backward reduce 123.4 -> 115.5, backward apply 98.1 -> 94.7, forward 82.8 -> 99.4 (plain there)
template <int BIT>
__device__ __forceinline__ f32x4 pool_ld(const float* p) {
  if constexpr ((5 & BIT) != 0) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return *reinterpret_cast<const f32x4*>(p);
}


// streaming loads of the one-element-per-thread apply passes with the non-temporal hint (round 5): they are the LAST reader of
// what they stream for a long while (the forward apply of z until the backward; the backward apply of g and z for good), so the
// lines need not stay in L2 / the memory-side cache: 536.2 -> 539.6 tiles/s, three interleaved rounds on one box
// (tools/ab_lib.sh).  The reduce pass keeps plain loads: the apply pass re-reads its data.
__device__ __forceinline__ f32x4 bn_ld(const float* p, size_t i) {
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i);
}

constexpr int kMaxStatBlocks = 2048;

// the slots of an output's operand-scale buffer start empty (block_absmax below fills them)
__device__ __forceinline__ void zero_amax(uint32_t* __restrict__ amax) {   // by workgroup 0 of the finalisation kernels
  if (amax && blockIdx.x == 0 && threadIdx.x < kAmaxSlots) amax[threadIdx.x * kAmaxStride] = 0;
}

struct BnPlan {
  int nblk;
  int64_t rows_per_blk;
  int tpc, rl;
};
static BnPlan bn_plan(int64_t rows, int C, int64_t per_override = 0, int64_t cap_override = 0) {
  BnPlan p;
  const int c4 = C / 4;
  p.tpc = c4 < 256 ? c4 : 256;
  p.rl = 256 / p.tpc;
  // ~32K elements per workgroup.  Round 5, three interleaved rounds on each of two boxes, tiles/s:
  // 65536 549.1 / 525.7, 49152 - / 526.6, 40960 - / 528.1, 32768 551.2 / 528.7 (+0.4 / +0.6 %), 24576 - / 526.8, 16384 545.9 / -:
  // twice the workgroups halve the latency-bound reduce passes on the small maps, four times cost more in the finalisation
  constexpr int64_t per = 32768;
  const int64_t per_e = per_override > 0 ? per_override : per;
  int64_t nb = (rows * (int64_t)C + per_e - 1) / per_e;
  // at most TWO workgroups per CU (<= kMaxStatBlocks): whole rounds of the chip and a quarter of the partials for
  // the finalisation of the large maps.  Three interleaved rounds, two boxes: 2048 542.9 / 546.4, 1024 543.1, 768 - / 547.4,
  // 640 - / 545.9, 512 546.1 / 550.0 (+0.6 / +0.65 %), 384 - / 547.4, 256 538.6
  constexpr int64_t cap = 512;
  if (nb > (cap_override > 0 ? cap_override : cap)) nb = cap_override > 0 ? cap_override : cap;
  if (nb > kMaxStatBlocks) nb = kMaxStatBlocks;
  if (nb < 1) nb = 1;
  int64_t rpb = (rows + nb - 1) / nb;
  rpb = ((rpb + p.rl - 1) / p.rl) * p.rl;
  p.rows_per_blk = rpb;
  p.nblk = (int)((rows + rpb - 1) / rpb);
  return p;
}

// partial[blk][0][C] = sum (x - pivot), partial[blk][1][C] = sum (x - pivot)^2 with pivot[c] = x[0][c].
// Shifting by a sample of the same channel keeps var = E[d^2] - E[d]^2 free of the catastrophic
// cancellation the raw moments suffer when |mean| >> std (8-sample statistics at 2x2 maps).
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                               int64_t rows, int C, int64_t rows_per_blk, int tpc,
                                                               int rl) {
  __shared__ f32x4 red[2][256];
  const int c4 = C >> 2;
  const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  const int64_t r1 = min(rows, r0 + rows_per_blk);
  for (int cb = tc; cb < c4; cb += tpc) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    const f32x4 pv = *reinterpret_cast<const f32x4*>(x + cb * 4);
    if (tr < rl) {
      // 4 independent 16-byte loads in flight per lane: the reduction is latency bound otherwise
      // (measured 2.7 TB/s with one load per lane per trip)
      int64_t r = r0 + tr;
      const int64_t st = rl;
      for (; r + 3 * st < r1; r += 4 * st) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + r * C + cb * 4) - pv;
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + (r + st) * C + cb * 4) - pv;
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(x + (r + 2 * st) * C + cb * 4) - pv;
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(x + (r + 3 * st) * C + cb * 4) - pv;
        s += (v0 + v1) + (v2 + v3);
        q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      }
      for (; r < r1; r += st) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + cb * 4) - pv;
        s += v;
        q += v * v;
      }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (tr == 0) {
      for (int k = 1; k < rl; ++k) {
        s += red[0][k * tpc + tc];
        q += red[1][k * tpc + tc];
      }
      float* o = partial + (size_t)blockIdx.x * 2 * C;
      *reinterpret_cast<f32x4*>(o + cb * 4) = s;
      *reinterpret_cast<f32x4*>(o + C + cb * 4) = q;
    }
    __syncthreads();
  }
}

// Sum the per-workgroup partials of 8 channels with 32 lanes per channel in fp64 (four independent loads
// in flight per lane: the chain of up to 2048 partials per channel is latency bound, and a grid of C/8
// workgroups instead of C/32 spreads it over more CUs), then fold the 32 lanes through LDS in a fixed
// order.  Returns true on the lane that holds the totals.
constexpr int kFinCh = 8;
// Fold one value per thread over the FL lanes of a channel (thread = lane * FC + channel, FC * FL = 256): xor-shuffles
// inside a wave (a channel's lanes sit FC apart), then the four waves' results through LDS — a fixed tree, so the result
// is reproducible, and 4 + log2 steps where a serial fold by one thread took FL dependent LDS round trips (32 / 128 of
// them: 4 / 16 us of the 6 / 16 us these finalisation launches took).  Every thread gets the total of its channel.
template <int FC, typename T, typename Op>
__device__ __forceinline__ T fold_channel_lanes(T v, T (*lds)[FC], Op op) {
#pragma unroll
  for (int o = FC; o < 64; o <<= 1) v = op(v, __shfl_xor(v, o, 64));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane < FC) lds[w][lane] = v;
  __syncthreads();
  const int tc = threadIdx.x % FC;
  const T r = op(op(lds[0][tc], lds[1][tc]), op(lds[2][tc], lds[3][tc]));
  __syncthreads();
  return r;
}
template <int FC = kFinCh>
__device__ __forceinline__ bool reduce_partials(const float* __restrict__ partial, int nblk, int C, int& c, double& s,
                                                double& q) {
  __shared__ double red[4][FC];
  const int tc = threadIdx.x % FC, tl = threadIdx.x / FC;
  c = bn_fin_group() * FC + tc;
  s = 0.0;
  q = 0.0;
  if (c < C) {
    const float* ps = partial + c;
    const size_t st = (size_t)2 * C;
    int b = tl;
    for (; b + 3 * (256 / FC) < nblk; b += 4 * (256 / FC)) {
      const float s0 = ps[b * st], s1 = ps[(b + (256 / FC)) * st], s2 = ps[(b + 2 * (256 / FC)) * st],
                  s3 = ps[(b + 3 * (256 / FC)) * st];
      const float q0 = ps[b * st + C], q1 = ps[(b + (256 / FC)) * st + C], q2 = ps[(b + 2 * (256 / FC)) * st + C],
                  q3 = ps[(b + 3 * (256 / FC)) * st + C];
      s += ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
      q += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
    }
    for (; b < nblk; b += (256 / FC)) {
      s += (double)ps[b * st];
      q += (double)ps[b * st + C];
    }
  }
  auto add = [](double a, double b) { return a + b; };
  s = fold_channel_lanes<FC>(s, red, add);
  q = fold_channel_lanes<FC>(q, red, add);
  return tl == 0 && c < C;
}

// mean / biased var -> save_mean, save_invstd, scale/shift for the apply pass; running stats update
// follows torch: running = (1-m)*running + m*stat, with the UNBIASED variance (n/(n-1)).
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ partial, int nblk, int C,
                                                             double inv_rows, double unbias,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float momentum,
                                                             float eps, float* __restrict__ save_mean,
                                                             float* __restrict__ save_invstd,
                                                             float* __restrict__ scale_shift,
                                                             uint32_t* __restrict__ amax) {
  zero_amax(amax);   // the apply pass accumulates max|y| into its slots
  int c;
  double s, q;
  if (!reduce_partials(partial, nblk, C, c, s, q)) return;
  const double dm = s * inv_rows;  // mean of (x - pivot)
  const double mean = (double)x[c] + dm;
  double var = q * inv_rows - dm * dm;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  save_mean[c] = meanf;
  save_invstd[c] = invstd;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = g * invstd;
  scale_shift[c] = sc;
  scale_shift[C + c] = b - meanf * sc;
}

__global__ void bn_eval_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                                    float* __restrict__ scale_shift, float* __restrict__ save_mean,
                                    float* __restrict__ save_invstd, uint32_t* __restrict__ amax) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  zero_amax(amax);
  if (c >= C) return;
  const float invstd = (float)(1.0 / sqrt((double)rv[c] + (double)eps));
  if (save_mean) save_mean[c] = rm[c];
  if (save_invstd) save_invstd[c] = invstd;
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = g * invstd;
  scale_shift[c] = sc;
  scale_shift[C + c] = b - rm[c] * sc;
}

// Channel chunk of flat 16-byte element i: (i mod c4) without a per-lane 64-bit division — the workgroup's base
// goes through the scalar unit, the lane offset (< 256 + c4 <= 768) through an exact float reciprocal.
__device__ __forceinline__ int chunk_of(size_t blk_base, int c4) {
  const uint32_t v = (uint32_t)(blk_base % (size_t)c4) + threadIdx.x;
  const uint32_t q = (uint32_t)((float)v * (1.0f / (float)c4));
  int r = (int)(v - q * (uint32_t)c4);
  r = r < 0 ? r + c4 : r;
  return r >= c4 ? r - c4 : r;
}

// max|v| over a workgroup's 16-byte elements into the output's operand-scale buffer (bit image; the f16x2 convolution
// arithmetic's scale of the tensor being written, x3_common.hpp act_absmax): ONE atomic max per workgroup on slot
// (workgroup & 63), the slots a cache line apart and zeroed by the finalisation kernel launched just before.  Measured:
// free (BatchNorm family 4.16 -> 4.13 TB/s).  What was not: an atomic or a guarded read of ONE word from every wave, and
// a last-arriver fold with a device-scope fence per workgroup (0.4 TB/s each); per-workgroup words + a fold launch worked
// but cost 136 launches of 5.7 us per step.
__device__ __forceinline__ void block_absmax(const f32x4 v, bool valid, uint32_t* __restrict__ amax) {
  __shared__ uint32_t red[4];
  uint32_t m = 0;
  if (valid) {
    m = __builtin_bit_cast(uint32_t, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    if (v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w) m = 0x7fc00000u;   // fmaxf drops NaNs: keep them visible
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = max(max(red[0], red[1]), max(red[2], red[3]));
    if (t) atomicMax(&amax[(blockIdx.x & (kAmaxSlots - 1)) * kAmaxStride], t);
  }
}

// y = act(x*scale + shift [+ residual]).
// ONE 16-byte element per thread and a grid of n4/256 workgroups: on the 268 MB maps a 1-read + 1-write stream
// runs at 6.1 TB/s this way against 4.3 TB/s for 2048 grid-striding workgroups with four loads in flight each
// (tools/probes/copy_patterns.hip) — the resident workgroups then sweep one contiguous window of HBM in dispatch
// order instead of 2048 x 4 scattered pages.  scale/shift (2C floats) come from L1/L2, not LDS: staging them
// per workgroup would cost more than the 4 KB a workgroup streams.
// PK: y is written as packed words of y / s (x3_common.hpp: pack_hl), s from the bound the finalisation left in slot 0
// of `amax` — the tensor is the next convolution's operand and nothing else.
template <bool PK = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ residual,
                                                       const float* __restrict__ scale_shift, float* __restrict__ y,
                                                       size_t n4, int C, int relu, uint32_t* __restrict__ amax,
                                                       uint32_t* __restrict__ relu_bits = nullptr) {
  const size_t base = (size_t)bn_blk() * 256;
  const size_t i = base + threadIdx.x;
  const bool valid = i < n4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  float pk_inv = 1.f;
  if constexpr (PK) pk_inv = op_scale(amax[0]).inv;
  if (valid) {
    const int c4 = C >> 2;
    const int cb = chunk_of(base, c4);
    const f32x4 sc = reinterpret_cast<const f32x4*>(scale_shift)[cb];
    const f32x4 sh = reinterpret_cast<const f32x4*>(scale_shift + C)[cb];
    v = bn_ld(x, i) * sc + sh;
    if (residual) v += bn_ld(residual, i);
  }
  // the backward's mask, one bit per element (common.hpp: relu_bits_*): every lane of the wave takes part (v = 0 where
  // the element does not exist), i >> 6 is this wave's chunk
  if (relu_bits) relu_bits_store(relu_bits, i >> 6, v.x > 0.f, v.y > 0.f, v.z > 0.f, v.w > 0.f);
  if (valid) {
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if constexpr (PK) {
      reinterpret_cast<u32x4*>(y)[i] = pack_hl4(v, pk_inv);
    } else {
      reinterpret_cast<f32x4*>(y)[i] = v;
    }
  }
  if constexpr (!PK)
    if (amax) block_absmax(v, valid, amax);   // all lanes of the workgroup arrive here together
}

// Backward stage 1: g = dy * (y > 0) ; partial[blk][0][C] = sum g, [1][C] = sum g * xhat.
// Optionally writes g to d_residual.
// PK (dx will be written packed, EVK_BN_PACK_DX): also pmax[blk][0][C] = max |g|, [1][C] = max |xhat| — what the
// finalisation needs to bound |dx| per channel BEFORE the apply pass writes it under that scale.
// (plain loads: the apply pass re-reads what this pass streams.  The non-temporal hint here, gated by map size, measured
// level from 128 MB up and -0.3 % below — removed)
__device__ __forceinline__ f32x4 bp_ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <bool PK>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float* __restrict__ d_residual,
                                                             float* __restrict__ partial, int64_t rows, int C,
                                                             int64_t rows_per_blk, int tpc, int rl, int relu,
                                                             float* __restrict__ pmax,
                                                             const uint32_t* __restrict__ bits = nullptr) {
  // relu: 0 none, 1 mask from the saved output y, 2 mask recomputed from x (no residual: the
  // forward's pre-activation is x*sc+sh with the same sc/sh arithmetic as bn_stats_final_kernel), 3 mask from `bits`
  // (common.hpp: relu_bits_*: the forward's own bits, or the mask of a gradient that arrives unmasked — EVK_BN_LAZY_RES)
  __shared__ f32x4 red[2][256];
  const int c4 = C >> 2;
  const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
  const int64_t r0 = (int64_t)bn_blk() * rows_per_blk;
  const int64_t r1 = min(rows, r0 + rows_per_blk);
  for (int cb = tc; cb < c4; cb += tpc) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + cb * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + cb * 4);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (relu == 2) {
      const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
      const f32x4 g4 = gamma ? *reinterpret_cast<const f32x4*>(gamma + cb * 4) : one;
      const f32x4 b4 = beta ? *reinterpret_cast<const f32x4*>(beta + cb * 4) : zero;
      sc = g4 * is;
      sh = b4 - mu * sc;
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    f32x4 gm = {0.f, 0.f, 0.f, 0.f}, xm = {0.f, 0.f, 0.f, 0.f};
    if (tr < rl) {
      auto one = [&](const f32x4 gin, const f32x4 xv, const f32x4 yin, size_t off) {
        f32x4 g = gin;
        if (relu == 3) {
          g = relu_bits_mask(g, bits, off >> 2);
        } else if (relu) {
          const f32x4 yy = (relu == 1) ? yin : xv * sc + sh;
          g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
          g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        }
        if (d_residual) *reinterpret_cast<f32x4*>(d_residual + off) = g;
        s += g;
        const f32x4 xh = (xv - mu) * is;
        q += g * xh;
        if constexpr (PK) {
          gm.x = fmaxf(gm.x, fabsf(g.x)); gm.y = fmaxf(gm.y, fabsf(g.y));
          gm.z = fmaxf(gm.z, fabsf(g.z)); gm.w = fmaxf(gm.w, fabsf(g.w));
          xm.x = fmaxf(xm.x, fabsf(xh.x)); xm.y = fmaxf(xm.y, fabsf(xh.y));
          xm.z = fmaxf(xm.z, fabsf(xh.z)); xm.w = fmaxf(xm.w, fabsf(xh.w));
        }
      };
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      int64_t r = r0 + tr;
      const int64_t st = rl;
      // four rows per trip: 8-12 independent 16-byte loads in flight per lane
      for (; r + 3 * st < r1; r += 4 * st) {
        const size_t o0 = (size_t)r * C + cb * 4, o1 = (size_t)(r + st) * C + cb * 4;
        const size_t o2 = (size_t)(r + 2 * st) * C + cb * 4, o3 = (size_t)(r + 3 * st) * C + cb * 4;
        const f32x4 g0 = bp_ld(dy + o0), g1 = bp_ld(dy + o1);
        const f32x4 g2 = bp_ld(dy + o2), g3 = bp_ld(dy + o3);
        const f32x4 x0 = bp_ld(x + o0), x1 = bp_ld(x + o1);
        const f32x4 x2 = bp_ld(x + o2), x3 = bp_ld(x + o3);
        f32x4 y0 = zero4, y1 = zero4, y2 = zero4, y3 = zero4;
        if (relu == 1) {
          y0 = bp_ld(y + o0);
          y1 = bp_ld(y + o1);
          y2 = bp_ld(y + o2);
          y3 = bp_ld(y + o3);
        }
        one(g0, x0, y0, o0);
        one(g1, x1, y1, o1);
        one(g2, x2, y2, o2);
        one(g3, x3, y3, o3);
      }
      for (; r < r1; r += st) {
        const size_t o0 = (size_t)r * C + cb * 4;
        const f32x4 g0 = bp_ld(dy + o0);
        const f32x4 x0 = bp_ld(x + o0);
        const f32x4 y0 = (relu == 1) ? bp_ld(y + o0) : zero4;
        one(g0, x0, y0, o0);
      }
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (tr == 0) {
      for (int k = 1; k < rl; ++k) {
        s += red[0][k * tpc + tc];
        q += red[1][k * tpc + tc];
      }
      float* o = partial + (size_t)bn_blk() * 2 * C;
      *reinterpret_cast<f32x4*>(o + cb * 4) = s;
      *reinterpret_cast<f32x4*>(o + C + cb * 4) = q;
    }
    __syncthreads();
    if constexpr (PK) {
      red[0][threadIdx.x] = gm;
      red[1][threadIdx.x] = xm;
      __syncthreads();
      if (tr == 0) {
        for (int k = 1; k < rl; ++k) {
          const f32x4 a = red[0][k * tpc + tc], b = red[1][k * tpc + tc];
          gm.x = fmaxf(gm.x, a.x); gm.y = fmaxf(gm.y, a.y); gm.z = fmaxf(gm.z, a.z); gm.w = fmaxf(gm.w, a.w);
          xm.x = fmaxf(xm.x, b.x); xm.y = fmaxf(xm.y, b.y); xm.z = fmaxf(xm.z, b.z); xm.w = fmaxf(xm.w, b.w);
        }
        float* o = pmax + (size_t)bn_blk() * 2 * C;
        *reinterpret_cast<f32x4*>(o + cb * 4) = gm;
        *reinterpret_cast<f32x4*>(o + C + cb * 4) = xm;
      }
      __syncthreads();
    }
  }
}

// coef[0][C] = gamma*invstd ; coef[1][C] = mean(g) ; coef[2][C] = mean(g*xhat)  (0 when !train)
// FC channels x 256 / FC lanes per workgroup: 8 x 32 by default; 2 x 128 (four times the workgroups) when the partials are many
template <int FC = kFinCh>
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ partial, int nblk, int C,
                                                           double inv_rows, const float* __restrict__ gamma,
                                                           const float* __restrict__ invstd,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ coef, int train,
                                                           uint32_t* __restrict__ amax,
                                                           const float* __restrict__ pmax) {
  // pmax == nullptr: stage 3 accumulates max|dx| into the slots (zeroed here).  pmax != nullptr (packed dx): the slots
  // arrive ZERO and every workgroup raises one of them to a bound of its channels' |dx| before stage 3 starts:
  //   |dx_c| = |k0 (g - k1 - xhat k2)| <= |k0| (max|g_c| + |k1| + max|xhat_c| |k2|)
  // (tight to a factor of ~2; an upper bound is all a scale needs, x3_common.hpp).
  if (!pmax) zero_amax(amax);
  float gmax = 0.f, xmax = 0.f;
  if (pmax) {
    __shared__ float mred[4][FC];
    const int tc = threadIdx.x % FC, tl = threadIdx.x / FC;
    const int cc = bn_fin_group() * FC + tc;
    if (cc < C) {
      const float* pm = pmax + cc;
      const size_t st = (size_t)2 * C;
      int b = tl;
      for (; b + 3 * (256 / FC) < nblk; b += 4 * (256 / FC)) {   // eight independent loads in flight per lane
        const float g0 = pm[b * st], g1 = pm[(b + (256 / FC)) * st], g2 = pm[(b + 2 * (256 / FC)) * st],
                    g3 = pm[(b + 3 * (256 / FC)) * st];
        const float x0 = pm[b * st + C], x1 = pm[(b + (256 / FC)) * st + C], x2 = pm[(b + 2 * (256 / FC)) * st + C],
                    x3 = pm[(b + 3 * (256 / FC)) * st + C];
        gmax = fmaxf(gmax, fmaxf(fmaxf(g0, g1), fmaxf(g2, g3)));
        xmax = fmaxf(xmax, fmaxf(fmaxf(x0, x1), fmaxf(x2, x3)));
      }
      for (; b < nblk; b += (256 / FC)) {
        gmax = fmaxf(gmax, pm[b * st]);
        xmax = fmaxf(xmax, pm[b * st + C]);
      }
    }
    auto mx = [](float a, float b) { return fmaxf(a, b); };
    gmax = fold_channel_lanes<FC>(gmax, mred, mx);
    xmax = fold_channel_lanes<FC>(xmax, mred, mx);
  }
  int c;
  double s, q;
  if (!reduce_partials<FC>(partial, nblk, C, c, s, q)) return;
  if (dbeta) dbeta[c] = (float)s;
  if (dgamma) dgamma[c] = (float)q;
  const float g = gamma ? gamma[c] : 1.f;
  const float k0 = g * invstd[c], k1 = train ? (float)(s * inv_rows) : 0.f, k2 = train ? (float)(q * inv_rows) : 0.f;
  coef[c] = k0;
  coef[C + c] = k1;
  coef[2 * C + c] = k2;
  if (pmax) {
    const float bound = fabsf(k0) * (gmax + fabsf(k1) + xmax * fabsf(k2));
    uint32_t bits = __builtin_bit_cast(uint32_t, bound);
    if (bound != bound) bits = 0x7fc00000u;
    // slot 0 only (C / 8 workgroups in this launch: nothing to spread), so that the apply pass — one 16-byte element
    // per thread — reads ONE word instead of folding 64 cache lines per wave (measured: the fold cost the pass 5 %)
    if (bits) atomicMax(&amax[0], bits);
  }
}

// dx = coef0 * (g - coef1 - xhat*coef2); one 16-byte element per thread (see bn_apply_kernel), the per-channel
// vectors read through L1.  PK: dx is written as packed words of dx / s (x3_common.hpp: pack_hl), s from the bound the
// finalisation left in `amax` — the tensor is the producing convolution's dy operand and nothing else.
template <bool PK>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ coef,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ dx,
                                                           size_t n4, int C, int relu, uint32_t* __restrict__ amax,
                                                           const uint32_t* __restrict__ bits = nullptr) {
  const size_t base = (size_t)bn_blk() * 256;
  const size_t i = base + threadIdx.x;
  const bool valid = i < n4;
  f32x4 out = {0.f, 0.f, 0.f, 0.f};
  float pk_inv = 1.f;
  if constexpr (PK) pk_inv = op_scale(amax[0]).inv;   // the finalisation's bound lives in slot 0 alone
  if (valid) {
    const int c4 = C >> 2;
    const int c = chunk_of(base, c4);
    const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 k0 = reinterpret_cast<const f32x4*>(coef)[c];
    const f32x4 k1 = reinterpret_cast<const f32x4*>(coef + C)[c];
    const f32x4 k2 = reinterpret_cast<const f32x4*>(coef + 2 * C)[c];
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c];
    const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[c];
    f32x4 g = bn_ld(dy, i);
    const f32x4 xv = bn_ld(x, i);
    if (relu == 3) {
      g = relu_bits_mask(g, bits, i);
    } else if (relu) {
      f32x4 yy;
      if (relu == 1) {
        yy = reinterpret_cast<const f32x4*>(y)[i];
      } else {  // the forward's pre-activation, same sc/sh arithmetic as bn_stats_final_kernel
        const f32x4 sc = (gamma ? reinterpret_cast<const f32x4*>(gamma)[c] : one4) * is;
        const f32x4 sh = (beta ? reinterpret_cast<const f32x4*>(beta)[c] : z4) - mu * sc;
        yy = xv * sc + sh;
      }
      g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
      g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
    }
    const f32x4 xh = (xv - mu) * is;
    out = k0 * (g - k1 - xh * k2);
    if constexpr (PK) {
      reinterpret_cast<u32x4*>(dx)[i] = pack_hl4(out, pk_inv);
    } else {
      reinterpret_cast<f32x4*>(dx)[i] = out;
    }
  }
  if constexpr (!PK)
    if (amax) block_absmax(out, valid, amax);   // all lanes of the workgroup arrive here together
}

// Statistics that arrive as per-row-part records (count, mean, M2) from the producing convolution's epilogue
// (igemm_common.hpp: igemm_store_rows_stats): Chan's pairwise merge in fp64, 32 lanes per channel over a fixed strided
// subset each, then the 32 lanes folded in index order — the same outputs as bn_stats_final_kernel, no pass over x.
// FC channels x FL record lanes per workgroup (FC * FL = 256): 8 x 32 for the small maps; 2 x 128 where a channel has
// hundreds of records (the 128^2 maps: 2048 per channel — at 32 lanes each lane walked 64 strided records: 9-47 us a launch)
template <int FC, int FL>
__global__ __launch_bounds__(256) void bn_parts_final_kernel(const float* __restrict__ parts, int nparts, int C,
                                                             double rows, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float momentum, float eps,
                                                             float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                             float* __restrict__ scale_shift,
                                                             uint32_t* __restrict__ amax, int pack) {
  // pack == 0: the apply pass accumulates max|y| into the slots (zeroed here).  pack != 0 (y will be written packed,
  // EVK_BN_PACK_Y): the slots arrive ZERO and slot 0 is raised HERE to a bound of |y|, from the records alone: a row of
  // record i deviates from the record's mean by at most sqrt(M2_i) (one term of the sum), so
  //   |x - mean_c| <= E_c + |pivot - mean_c|,  E_c = max_i (sqrt(M2_i) + |mean_i - pivot|),  |y_c| <= |sc| (..) + |beta|
  // — about 2-3x the true maximum at 64..256 rows per record; an upper bound is all a scale needs (x3_common.hpp).
  if (!pack) zero_amax(amax);
  float E = 0.f;
  // With a common pivot p (the first record's mean) the merge of all records is three plain sums,
  //   N = sum n_i,  A = sum n_i (mean_i - p),  B = sum [M2_i + n_i (mean_i - p)^2]:  mean = p + A/N,  M2 = B - A^2/N
  // (Chan's pairwise formula telescoped; no division inside the loop, no order dependence beyond the fixed one below).
  __shared__ double red[4][FC];
  const int tc = threadIdx.x % FC, tl = threadIdx.x / FC;
  const int c = bn_fin_group() * FC + tc;
  double N = 0.0, A = 0.0, B = 0.0, piv = 0.0;
  if (c < C) {
    const size_t st = (size_t)3 * C;
    piv = (double)parts[C + c];
    const float* r = parts + c;
    int b = tl;
    for (; b + 3 * FL < nparts; b += 4 * FL) {   // four independent records in flight per lane
      const float* r0 = r + (size_t)b * st;
      const float* r1 = r0 + (size_t)FL * st;
      const float* r2 = r1 + (size_t)FL * st;
      const float* r3 = r2 + (size_t)FL * st;
      const float n0 = r0[0], m0 = r0[C], q0 = r0[2 * C], n1 = r1[0], m1 = r1[C], q1 = r1[2 * C];
      const float n2 = r2[0], m2 = r2[C], q2 = r2[2 * C], n3 = r3[0], m3 = r3[C], q3 = r3[2 * C];
      const double d0 = (double)m0 - piv, d1 = (double)m1 - piv, d2 = (double)m2 - piv, d3 = (double)m3 - piv;
      N += ((double)n0 + (double)n1) + ((double)n2 + (double)n3);
      A += ((double)n0 * d0 + (double)n1 * d1) + ((double)n2 * d2 + (double)n3 * d3);
      B += (((double)q0 + (double)n0 * d0 * d0) + ((double)q1 + (double)n1 * d1 * d1)) +
           (((double)q2 + (double)n2 * d2 * d2) + ((double)q3 + (double)n3 * d3 * d3));
      if (pack) {
        const float e0 = n0 > 0.f ? sqrtf(q0) + fabsf((float)d0) : 0.f, e1 = n1 > 0.f ? sqrtf(q1) + fabsf((float)d1) : 0.f;
        const float e2 = n2 > 0.f ? sqrtf(q2) + fabsf((float)d2) : 0.f, e3 = n3 > 0.f ? sqrtf(q3) + fabsf((float)d3) : 0.f;
        E = fmaxf(E, fmaxf(fmaxf(e0, e1), fmaxf(e2, e3)));
      }
    }
    for (; b < nparts; b += FL) {
      const float* r0 = r + (size_t)b * st;
      const double n0 = (double)r0[0], d0 = (double)r0[C] - piv;
      N += n0;
      A += n0 * d0;
      B += (double)r0[2 * C] + n0 * d0 * d0;
      if (pack && n0 > 0.0) E = fmaxf(E, sqrtf(r0[2 * C]) + fabsf((float)d0));
    }
  }
  __shared__ float ered[4][FC];
  auto add = [](double a, double b) { return a + b; };
  N = fold_channel_lanes<FC>(N, red, add);
  A = fold_channel_lanes<FC>(A, red, add);
  B = fold_channel_lanes<FC>(B, red, add);
  if (pack) E = fold_channel_lanes<FC>(E, ered, [](float a, float b) { return fmaxf(a, b); });
  if (tl != 0 || c >= C) return;
  const double mean = N > 0.0 ? piv + A / N : 0.0;
  double var = N > 0.0 ? (B - A * A / N) / N : 0.0;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  save_mean[c] = meanf;
  save_invstd[c] = invstd;
  const double unbias = rows > 1.0 ? rows / (rows - 1.0) : 1.0;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
  const float g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
  const float sc = g * invstd;
  scale_shift[c] = sc;
  scale_shift[C + c] = bb - meanf * sc;
  if (pack) {
    const float bound = fabsf(sc) * (E * 1.001f + fabsf((float)(piv - mean))) + fabsf(bb);
    uint32_t bits = __builtin_bit_cast(uint32_t, bound);
    if (bound != bound) bits = 0x7fc00000u;
    if (bits) atomicMax(&amax[0], bits);   // slot 0 alone: the apply pass reads one word (see bn_bwd_final_kernel)
  }
}

// ------------------------------------------------------------------------------------------------
// The stem's BatchNorm + ReLU + MaxPool2d(3, 2, 1) as ONE pass each way (reference _resnets.py:150-153: bn1, relu,
// maxpool on the 7x7 convolution's output, a quarter-resolution consumer of a full-resolution tensor).  Separate
// kernels wrote and re-read the normalised 268 MB map forward, and scattered the pooled gradient into a 268 MB map
// for the BatchNorm backward to read twice.  Here the forward reads the convolution output once and writes the pooled
// map + the winning tap of every window (the codes of maxpool_fwd_kernel: first maximum in scan order, NaN wins); the
// backward's two passes rebuild g = dz * (z > 0) on the fly — an input pixel is tap (iy - 2oy + 1, ix - 2ox + 1) of at
// most 2 x 2 windows: g = sum of their dp where that tap won — from the 67 MB pooled gradient and the 17 MB codes.
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ scale_shift,
                                                               float* __restrict__ y, uint8_t* __restrict__ code, int N,
                                                               int H, int W, int C, int Ho, int Wo,
                                                               uint32_t* __restrict__ amax) {
  const int c4 = C >> 2;
  const uint32_t total = (uint32_t)N * Ho * Wo * c4;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const bool valid = i < total;
  f32x4 best = {0.f, 0.f, 0.f, 0.f};
  if (valid) {
    const int cb = (int)(i % (uint32_t)c4);
    uint32_t pix = i / (uint32_t)c4;
    const int ox = (int)(pix % (uint32_t)Wo);
    pix /= (uint32_t)Wo;
    const int oy = (int)(pix % (uint32_t)Ho);
    const int n = (int)(pix / (uint32_t)Ho);
    const f32x4 sc = reinterpret_cast<const f32x4*>(scale_shift)[cb];
    const f32x4 sh = reinterpret_cast<const f32x4*>(scale_shift + C)[cb];
    int bx = 0, by = 0, bz = 0, bw = 0;
    bool first = true;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        f32x4 v = pool_ld<2>(x + (((size_t)n * H + iy) * W + ix) * C + cb * 4) * sc + sh;
        v.x = v.x != v.x ? v.x : fmaxf(v.x, 0.f); v.y = v.y != v.y ? v.y : fmaxf(v.y, 0.f);   // relu keeps NaN visible
        v.z = v.z != v.z ? v.z : fmaxf(v.z, 0.f); v.w = v.w != v.w ? v.w : fmaxf(v.w, 0.f);
        const int t = ky * 3 + kx;
        if (first) {
          best = v; bx = by = bz = bw = t; first = false;
        } else {
          if (v.x > best.x || v.x != v.x) { best.x = v.x; bx = t; }
          if (v.y > best.y || v.y != v.y) { best.y = v.y; by = t; }
          if (v.z > best.z || v.z != v.z) { best.z = v.z; bz = t; }
          if (v.w > best.w || v.w != v.w) { best.w = v.w; bw = t; }
        }
      }
    }
    const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + cb * 4;
    *reinterpret_cast<f32x4*>(y + o) = best;
    *reinterpret_cast<uint32_t*>(code + o) = (uint32_t)bx | ((uint32_t)by << 8) | ((uint32_t)bz << 16) | ((uint32_t)bw << 24);
  }
  if (amax) block_absmax(best, valid, amax);
}

// dz of input pixel (n, iy, ix), channel chunk cb: the pooled gradient of the windows this pixel won
__device__ __forceinline__ f32x4 pool_gather(const float* __restrict__ dp, const uint8_t* __restrict__ code, int n, int iy,
                                             int ix, int Ho, int Wo, int C, int cb) {
  f32x4 g = {0.f, 0.f, 0.f, 0.f};
  const int oy_lo = iy >> 1, oy_hi = min(Ho - 1, (iy + 1) >> 1);
  const int ox_lo = ix >> 1, ox_hi = min(Wo - 1, (ix + 1) >> 1);
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const int ky = iy - 2 * oy + 1;
    if ((unsigned)ky > 2u) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const int kx = ix - 2 * ox + 1;
      if ((unsigned)kx > 2u) continue;
      const uint32_t t = (uint32_t)(ky * 3 + kx);
      const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + cb * 4;
      const uint32_t cw = *reinterpret_cast<const uint32_t*>(code + o);
      const f32x4 d = *reinterpret_cast<const f32x4*>(dp + o);
      g.x += (cw & 0xffu) == t ? d.x : 0.f;
      g.y += ((cw >> 8) & 0xffu) == t ? d.y : 0.f;
      g.z += ((cw >> 16) & 0xffu) == t ? d.z : 0.f;
      g.w += (cw >> 24) == t ? d.w : 0.f;
    }
  }
  return g;
}

// Even H and W (every stem this network sees): a thread takes a 2 x 2 input QUAD (2k + {0,1}, 2l + {0,1}).  Its four
// pixels are taps of the same four windows (k, l), (k, l+1), (k+1, l), (k+1, l+1) — tap 4 / 5,3 / 7,1 / 8,6,2,0 — so four
// window look-ups serve four pixels instead of nine (pool_gather per pixel: 1 + 2 + 2 + 4).
struct PoolQuad { f32x4 g[4]; };   // dz of (2k,2l), (2k,2l+1), (2k+1,2l), (2k+1,2l+1)
__device__ __forceinline__ PoolQuad pool_gather_quad(const float* __restrict__ dp, const uint8_t* __restrict__ code, int n,
                                                     int k, int l, int Ho, int Wo, int C, int cb) {
  PoolQuad q;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  q.g[0] = q.g[1] = q.g[2] = q.g[3] = z;
  // tap of window (k + a, l + b) that each quad pixel is; 255 = not in that window
  constexpr uint32_t kTap[2][2][4] = {{{4u, 5u, 7u, 8u}, {255u, 3u, 255u, 6u}}, {{255u, 255u, 1u, 2u}, {255u, 255u, 255u, 0u}}};
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    if (k + a >= Ho) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (l + b >= Wo) continue;
      const size_t o = (((size_t)n * Ho + k + a) * Wo + l + b) * C + cb * 4;
      const uint32_t cw = *reinterpret_cast<const uint32_t*>(code + o);
      const f32x4 d = *reinterpret_cast<const f32x4*>(dp + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t t = kTap[a][b][e];
        if (t == 255u) continue;
        q.g[e].x += (cw & 0xffu) == t ? d.x : 0.f;
        q.g[e].y += ((cw >> 8) & 0xffu) == t ? d.y : 0.f;
        q.g[e].z += ((cw >> 16) & 0xffu) == t ? d.z : 0.f;
        q.g[e].w += (cw >> 24) == t ? d.w : 0.f;
      }
    }
  }
  return q;
}

// quad forms of the two passes below (H, W even): rows -> quads; the sums run over the same elements, in quad order
__global__ __launch_bounds__(256) void bn_pool_bwd_partial_quad_kernel(const float* __restrict__ dp,
                                                                       const uint8_t* __restrict__ code,
                                                                       const float* __restrict__ x,
                                                                       const float* __restrict__ mean,
                                                                       const float* __restrict__ invstd,
                                                                       const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta,
                                                                       float* __restrict__ partial, int quads, int H, int W,
                                                                       int C, int Ho, int Wo, int quads_per_blk, int tpc,
                                                                       int rl) {
  __shared__ f32x4 red[2][256];
  const int c4 = C >> 2;
  const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
  const int q0 = blockIdx.x * quads_per_blk, q1 = min(quads, q0 + quads_per_blk);
  const int Hq = H >> 1, Wq = W >> 1;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  for (int cb = tc; cb < c4; cb += tpc) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + cb * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + cb * 4);
    const f32x4 sc = (gamma ? *reinterpret_cast<const f32x4*>(gamma + cb * 4) : one) * is;
    const f32x4 sh = (beta ? *reinterpret_cast<const f32x4*>(beta + cb * 4) : zero) - mu * sc;
    f32x4 s = zero, q = zero;
    if (tr < rl)
      for (int qi = q0 + tr; qi < q1; qi += rl) {
        const int n = qi / (Hq * Wq), rem = qi - n * Hq * Wq;
        const int k = rem / Wq, l = rem - k * Wq;
        const PoolQuad pq = pool_gather_quad(dp, code, n, k, l, Ho, Wo, C, cb);
        const float* xb = x + (((size_t)n * H + 2 * k) * W + 2 * l) * C + cb * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 xv = pool_ld<1>(xb + ((size_t)(e >> 1) * W + (e & 1)) * C);
          f32x4 g = pq.g[e];
          const f32x4 yy = xv * sc + sh;
          g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
          g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
          s += g;
          q += g * ((xv - mu) * is);
        }
      }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (tr == 0) {
      for (int kk = 1; kk < rl; ++kk) {
        s += red[0][kk * tpc + tc];
        q += red[1][kk * tpc + tc];
      }
      float* o = partial + (size_t)blockIdx.x * 2 * C;
      *reinterpret_cast<f32x4*>(o + cb * 4) = s;
      *reinterpret_cast<f32x4*>(o + C + cb * 4) = q;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void bn_pool_bwd_apply_quad_kernel(const float* __restrict__ dp,
                                                                     const uint8_t* __restrict__ code,
                                                                     const float* __restrict__ x,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     const float* __restrict__ coef,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, float* __restrict__ dx,
                                                                     uint32_t nq4, int H, int W, int C, int Ho, int Wo,
                                                                     uint32_t* __restrict__ amax) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const bool valid = i < nq4;
  f32x4 mx = {0.f, 0.f, 0.f, 0.f};
  bool nan = false;
  if (valid) {
    const int c4 = C >> 2, Hq = H >> 1, Wq = W >> 1;
    const int cb = (int)(i % (uint32_t)c4);
    const int qi = (int)(i / (uint32_t)c4);
    const int n = qi / (Hq * Wq), rem = qi - n * Hq * Wq;
    const int k = rem / Wq, l = rem - k * Wq;
    const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 k0 = reinterpret_cast<const f32x4*>(coef)[cb];
    const f32x4 k1 = reinterpret_cast<const f32x4*>(coef + C)[cb];
    const f32x4 k2 = reinterpret_cast<const f32x4*>(coef + 2 * C)[cb];
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[cb];
    const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[cb];
    const f32x4 sc = (gamma ? reinterpret_cast<const f32x4*>(gamma)[cb] : one4) * is;
    const f32x4 sh = (beta ? reinterpret_cast<const f32x4*>(beta)[cb] : z4) - mu * sc;
    const PoolQuad pq = pool_gather_quad(dp, code, n, k, l, Ho, Wo, C, cb);
    const size_t base = (((size_t)n * H + 2 * k) * W + 2 * l) * C + cb * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t off = base + ((size_t)(e >> 1) * W + (e & 1)) * C;
      const f32x4 xv = pool_ld<4>(x + off);
      f32x4 g = pq.g[e];
      const f32x4 yy = xv * sc + sh;
      g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
      g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
      const f32x4 out = k0 * (g - k1 - ((xv - mu) * is) * k2);
      *reinterpret_cast<f32x4*>(dx + off) = out;
      mx.x = fmaxf(mx.x, fabsf(out.x)); mx.y = fmaxf(mx.y, fabsf(out.y));
      mx.z = fmaxf(mx.z, fabsf(out.z)); mx.w = fmaxf(mx.w, fabsf(out.w));
      nan = nan || out.x != out.x || out.y != out.y || out.z != out.z || out.w != out.w;
    }
    if (nan) mx.x = __builtin_nanf("");   // fmaxf drops NaNs: keep them visible in the scale slots (block_absmax tests v)
  }
  if (amax) block_absmax(mx, valid, amax);
}

// stage 1 of the backward (as bn_bwd_partial_kernel, ReLU mask recomputed from x): partial[blk][0][C] = sum g,
// [1][C] = sum g * xhat
__global__ __launch_bounds__(256) void bn_pool_bwd_partial_kernel(const float* __restrict__ dp,
                                                                  const uint8_t* __restrict__ code,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  float* __restrict__ partial, int rows, int H, int W,
                                                                  int C, int Ho, int Wo, int rows_per_blk, int tpc, int rl) {
  __shared__ f32x4 red[2][256];
  const int c4 = C >> 2;
  const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
  const int r0 = blockIdx.x * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  for (int cb = tc; cb < c4; cb += tpc) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + cb * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + cb * 4);
    const f32x4 sc = (gamma ? *reinterpret_cast<const f32x4*>(gamma + cb * 4) : one) * is;
    const f32x4 sh = (beta ? *reinterpret_cast<const f32x4*>(beta + cb * 4) : zero) - mu * sc;
    f32x4 s = zero, q = zero;
    if (tr < rl)
      for (int r = r0 + tr; r < r1; r += rl) {
        const int n = r / (H * W), rem = r - n * H * W;
        const int iy = rem / W, ix = rem - iy * W;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)r * C + cb * 4);
        f32x4 g = pool_gather(dp, code, n, iy, ix, Ho, Wo, C, cb);
        const f32x4 yy = xv * sc + sh;
        g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
        g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        s += g;
        q += g * ((xv - mu) * is);
      }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = q;
    __syncthreads();
    if (tr == 0) {
      for (int k = 1; k < rl; ++k) {
        s += red[0][k * tpc + tc];
        q += red[1][k * tpc + tc];
      }
      float* o = partial + (size_t)blockIdx.x * 2 * C;
      *reinterpret_cast<f32x4*>(o + cb * 4) = s;
      *reinterpret_cast<f32x4*>(o + C + cb * 4) = q;
    }
    __syncthreads();
  }
}

// stage 3: dx = coef0 * (g - coef1 - xhat * coef2), one 16-byte element per thread
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const float* __restrict__ dp,
                                                                const uint8_t* __restrict__ code,
                                                                const float* __restrict__ x,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ coef,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ dx,
                                                                uint32_t n4, int H, int W, int C, int Ho, int Wo,
                                                                uint32_t* __restrict__ amax) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const bool valid = i < n4;
  f32x4 out = {0.f, 0.f, 0.f, 0.f};
  if (valid) {
    const int c4 = C >> 2;
    const int cb = (int)(i % (uint32_t)c4);
    const int r = (int)(i / (uint32_t)c4);
    const int n = r / (H * W), rem = r - n * H * W;
    const int iy = rem / W, ix = rem - iy * W;
    const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 k0 = reinterpret_cast<const f32x4*>(coef)[cb];
    const f32x4 k1 = reinterpret_cast<const f32x4*>(coef + C)[cb];
    const f32x4 k2 = reinterpret_cast<const f32x4*>(coef + 2 * C)[cb];
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[cb];
    const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[cb];
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 g = pool_gather(dp, code, n, iy, ix, Ho, Wo, C, cb);
    const f32x4 sc = (gamma ? reinterpret_cast<const f32x4*>(gamma)[cb] : one4) * is;
    const f32x4 sh = (beta ? reinterpret_cast<const f32x4*>(beta)[cb] : z4) - mu * sc;
    const f32x4 yy = xv * sc + sh;
    g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
    g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
    out = k0 * (g - k1 - ((xv - mu) * is) * k2);
    reinterpret_cast<f32x4*>(dx)[i] = out;
  }
  if (amax) block_absmax(out, valid, amax);
}

// (A one-launch backward for small maps — every thread's rows in registers across two grid-wide barriers, 3 tensor transfers
// instead of 5 — lived here in rounds 2 and 3.  Its grid had to be resident as a whole, so it was off beside the weight-
// gradient side stream, under RCCL and on any stream but one; at the end it ran on ~1 call in 30 of the default path.
// Removed in round 4: EVK_BN_FUSED=1 vs 0 measured 547.7 / 547.4 tiles/s on the default path, 524.4 / 519.4 single-stream,
// 521.6 / 521.2 captured, same box — and with it went the spin barrier, its device words and the stream ownership.)

static unsigned oneshot_grid(size_t n4) { return (unsigned)((n4 + 255) / 256); }

static void launch_parts_final(hipStream_t st, const float* parts, int nparts, int C, double rows, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* save_mean, float* save_invstd, float* scale_shift, uint32_t* amax, int pack) {
  // one channel per workgroup above 1024 records — eight records per lane, all in flight at once (level with the two-channel
  // form on the step, ahead in the family: 407 vs 460 us per step)
  if (nparts >= 1024)
    hipLaunchKernelGGL((bn_parts_final_kernel<1, 256>), dim3(C), dim3(256), 0, st, parts, nparts, C, rows, gamma, beta,
                       running_mean, running_var, momentum, eps, save_mean, save_invstd, scale_shift, amax, pack);
  else if (nparts >= 512)
    hipLaunchKernelGGL((bn_parts_final_kernel<2, 128>), dim3((C + 1) / 2), dim3(256), 0, st, parts, nparts, C, rows, gamma, beta,
                       running_mean, running_var, momentum, eps, save_mean, save_invstd, scale_shift, amax, pack);
  else
    hipLaunchKernelGGL((bn_parts_final_kernel<8, 32>), dim3((C + 7) / 8), dim3(256), 0, st, parts, nparts, C, rows, gamma, beta,
                       running_mean, running_var, momentum, eps, save_mean, save_invstd, scale_shift, amax, pack);
}

}  // namespace evk

using namespace evk;

extern "C" size_t evk_bn_workspace_bytes(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return 0;
  // partial sums [blocks][2][C], 8 C of coefficients, per-block maxima [blocks][2][C] (packed outputs)
  return ((size_t)kMaxStatBlocks * 4 * C + 8 * (size_t)C) * sizeof(float);
}


extern "C" int evk_bn_fwd_train(const float* x, const float* residual, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* y,
                                float* save_mean, float* save_invstd, int64_t rows, int32_t C, uint32_t flags,
                                void* workspace, size_t workspace_bytes, uint32_t* y_absmax, void* stream) {
  EVK_REQUIRE(x && y && save_mean && save_invstd, EVK_E_INVALID, "bn_fwd_train: null pointer");
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, EVK_E_UNSUPPORTED, "bn_fwd_train: rows=%lld C=%d",
              (long long)rows, C);
  EVK_REQUIRE(workspace && workspace_bytes >= evk_bn_workspace_bytes(rows, C), EVK_E_WORKSPACE,
              "bn_fwd_train: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const BnPlan pl = bn_plan(rows, C);
  float* partial = (float*)workspace;
  float* scale_shift = partial + (size_t)kMaxStatBlocks * 2 * C;
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(pl.nblk), dim3(256), 0, st, x, partial, rows, C, pl.rows_per_blk,
                     pl.tpc, pl.rl);
  int rc = check_launch("bn_stats_partial");
  if (rc) return rc;
  const double unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, st, x, partial, pl.nblk, C,
                     1.0 / (double)rows, unbias, gamma, beta, running_mean, running_var, momentum, eps, save_mean,
                     save_invstd, scale_shift, y_absmax);
  rc = check_launch("bn_stats_final");
  if (rc) return rc;
  const size_t n4 = (size_t)rows * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(oneshot_grid(n4)), dim3(256), 0, st, x, residual,
                     scale_shift, y, n4, C, (flags & EVK_BN_RELU) ? 1 : 0, y_absmax);
  return check_launch("bn_apply");
}

extern "C" int evk_bn_fwd_train_parts_bits(const float* x, const float* residual, const float* gamma, const float* beta,
                                           float* running_mean, float* running_var, float momentum, float eps, float* y,
                                           float* save_mean, float* save_invstd, int64_t rows, int32_t C, uint32_t flags,
                                           const float* parts, int32_t nparts, void* workspace, size_t workspace_bytes,
                                           uint32_t* y_absmax, uint32_t* relu_bits, void* stream) {
  EVK_REQUIRE(x && y && save_mean && save_invstd && parts && nparts > 0, EVK_E_INVALID, "bn_fwd_train_parts: bad argument");
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, EVK_E_UNSUPPORTED, "bn_fwd_train_parts: rows=%lld C=%d",
              (long long)rows, C);
  EVK_REQUIRE(workspace && workspace_bytes >= evk_bn_workspace_bytes(rows, C), EVK_E_WORKSPACE,
              "bn_fwd_train_parts: workspace too small");
  const bool pack = (flags & EVK_BN_PACK_Y) != 0;
  EVK_REQUIRE(!pack || (y_absmax && !residual), EVK_E_INVALID,
              "bn_fwd_train_parts: EVK_BN_PACK_Y needs y_absmax (slots zero on entry) and no residual");
  hipStream_t st = (hipStream_t)stream;
  float* scale_shift = (float*)workspace + (size_t)kMaxStatBlocks * 2 * C;
  launch_parts_final(st, parts, nparts, C, (double)rows, gamma, beta, running_mean, running_var, momentum, eps, save_mean,
                     save_invstd, scale_shift, y_absmax, pack ? 1 : 0);
  int rc = check_launch("bn_parts_final");
  if (rc) return rc;
  const size_t n4 = (size_t)rows * C / 4;
  if (pack)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(oneshot_grid(n4)), dim3(256), 0, st, x, residual,
                       scale_shift, y, n4, C, (flags & EVK_BN_RELU) ? 1 : 0, y_absmax, relu_bits);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(oneshot_grid(n4)), dim3(256), 0, st, x, residual,
                       scale_shift, y, n4, C, (flags & EVK_BN_RELU) ? 1 : 0, y_absmax, relu_bits);
  return check_launch("bn_apply");
}
extern "C" int evk_bn_fwd_train_parts(const float* x, const float* residual, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float momentum, float eps, float* y,
                                      float* save_mean, float* save_invstd, int64_t rows, int32_t C, uint32_t flags,
                                      const float* parts, int32_t nparts, void* workspace, size_t workspace_bytes,
                                      uint32_t* y_absmax, void* stream) {
  return evk_bn_fwd_train_parts_bits(x, residual, gamma, beta, running_mean, running_var, momentum, eps, y, save_mean,
                                     save_invstd, rows, C, flags, parts, nparts, workspace, workspace_bytes, y_absmax, nullptr,
                                     stream);
}

// BatchNorm (batch statistics from the convolution epilogue's records) + ReLU + MaxPool2d(3, 2, 1): x [N,H,W,C] ->
// y [N,Ho,Wo,C], code [N,Ho,Wo,C] uint8 (winning tap of each window), Ho = (H - 1) / 2 + 1
extern "C" int evk_bn_relu_pool_fwd_train_parts(const float* x, const float* gamma, const float* beta, float* running_mean,
                                                float* running_var, float momentum, float eps, float* y, uint8_t* code,
                                                float* save_mean, float* save_invstd, int32_t N, int32_t H, int32_t W,
                                                int32_t C, const float* parts, int32_t nparts, void* workspace,
                                                size_t workspace_bytes, uint32_t* y_absmax, void* stream) {
  EVK_REQUIRE(x && y && code && save_mean && save_invstd && parts && nparts > 0, EVK_E_INVALID, "bn_relu_pool_fwd: bad argument");
  const int64_t rows = (int64_t)N * H * W;
  EVK_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C <= 2048 && rows * (C / 4) < 0x7fffffffLL, EVK_E_UNSUPPORTED,
              "bn_relu_pool_fwd: N=%d H=%d W=%d C=%d", N, H, W, C);
  EVK_REQUIRE(workspace && workspace_bytes >= evk_bn_workspace_bytes(rows, C), EVK_E_WORKSPACE,
              "bn_relu_pool_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* scale_shift = (float*)workspace + (size_t)kMaxStatBlocks * 2 * C;
  launch_parts_final(st, parts, nparts, C, (double)rows, gamma, beta, running_mean, running_var, momentum, eps, save_mean,
                     save_invstd, scale_shift, y_absmax, 0);
  int rc = check_launch("bn_parts_final");
  if (rc) return rc;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const size_t total = (size_t)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3(oneshot_grid(total)), dim3(256), 0, st, x, (const float*)scale_shift, y,
                     code, N, H, W, C, Ho, Wo, y_absmax);
  return check_launch("bn_relu_pool_fwd");
}

// its backward: dp [N,Ho,Wo,C] -> dx [N,H,W,C] (gradient of the convolution output), dgamma, dbeta
extern "C" int evk_bn_relu_pool_bwd(const float* dp, const uint8_t* code, const float* x, const float* gamma,
                                    const float* beta, const float* save_mean, const float* save_invstd, float* dx,
                                    float* dgamma, float* dbeta, int32_t N, int32_t H, int32_t W, int32_t C, int32_t train,
                                    void* workspace, size_t workspace_bytes, uint32_t* dx_absmax, void* stream) {
  EVK_REQUIRE(dp && code && x && save_mean && save_invstd && dx, EVK_E_INVALID, "bn_relu_pool_bwd: null pointer");
  const int64_t rows = (int64_t)N * H * W;
  EVK_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C <= 2048 && rows * (C / 4) < 0x7fffffffLL, EVK_E_UNSUPPORTED,
              "bn_relu_pool_bwd: N=%d H=%d W=%d C=%d", N, H, W, C);
  EVK_REQUIRE(workspace && workspace_bytes >= evk_bn_workspace_bytes(rows, C), EVK_E_WORKSPACE,
              "bn_relu_pool_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  // (the quad reduce pass gathers pooled gradients and codes per element: it keeps the finer split it was measured with —
  // 1024 workgroups on the stem's map: 86 us, 123 us under the two-per-CU cap of the plain reduce passes)
  const BnPlan pl = bn_plan(rows, C, 65536, kMaxStatBlocks);
  float* partial = (float*)workspace;
  float* coef = partial + (size_t)kMaxStatBlocks * 2 * C;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const bool quad = (H % 2 == 0) && (W % 2 == 0);
  // quads in place of rows, as many workgroups as the row plan (a quad is four rows' worth of elements)
  const int quads = (int)(rows / 4);
  int qpb = (quads + pl.nblk - 1) / pl.nblk;
  qpb = ((qpb + pl.rl - 1) / pl.rl) * pl.rl;
  const int nblk = quad ? (quads + qpb - 1) / qpb : pl.nblk;
  if (quad)
    hipLaunchKernelGGL(bn_pool_bwd_partial_quad_kernel, dim3(nblk), dim3(256), 0, st, dp, code, x, save_mean, save_invstd,
                       gamma, beta, partial, quads, H, W, C, Ho, Wo, qpb, pl.tpc, pl.rl);
  else
    hipLaunchKernelGGL(bn_pool_bwd_partial_kernel, dim3(pl.nblk), dim3(256), 0, st, dp, code, x, save_mean, save_invstd, gamma,
                       beta, partial, (int)rows, H, W, C, Ho, Wo, (int)pl.rows_per_blk, pl.tpc, pl.rl);
  int rc = check_launch("bn_pool_bwd_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_final_kernel<kFinCh>, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, st, (const float*)partial, nblk, C,
                     1.0 / (double)rows, gamma, save_invstd, dgamma, dbeta, coef, train ? 1 : 0, dx_absmax,
                     (const float*)nullptr);
  rc = check_launch("bn_bwd_final");
  if (rc) return rc;
  const uint32_t n4 = (uint32_t)(rows * (C / 4));
  if (quad)
    hipLaunchKernelGGL(bn_pool_bwd_apply_quad_kernel, dim3(oneshot_grid(n4 / 4)), dim3(256), 0, st, dp, code, x, save_mean,
                       save_invstd, (const float*)coef, gamma, beta, dx, n4 / 4, H, W, C, Ho, Wo, dx_absmax);
  else
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(oneshot_grid(n4)), dim3(256), 0, st, dp, code, x, save_mean, save_invstd,
                       (const float*)coef, gamma, beta, dx, n4, H, W, C, Ho, Wo, dx_absmax);
  return check_launch("bn_pool_bwd_apply");
}

extern "C" int evk_bn_fwd_eval(const float* x, const float* residual, const float* gamma, const float* beta,
                               const float* running_mean, const float* running_var, float eps, float* y,
                               float* save_mean, float* save_invstd, int64_t rows, int32_t C, uint32_t flags,
                               void* workspace, size_t workspace_bytes, uint32_t* y_absmax, void* stream) {
  EVK_REQUIRE(x && y && running_mean && running_var, EVK_E_INVALID, "bn_fwd_eval: null pointer");
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, EVK_E_UNSUPPORTED, "bn_fwd_eval: rows=%lld C=%d",
              (long long)rows, C);
  EVK_REQUIRE(workspace && workspace_bytes >= 2 * (size_t)C * sizeof(float), EVK_E_WORKSPACE,
              "bn_fwd_eval: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* scale_shift = (float*)workspace;
  hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, running_mean,
                     running_var, eps, C, scale_shift, save_mean, save_invstd, y_absmax);
  int rc = check_launch("bn_eval_coef");
  if (rc) return rc;
  const size_t n4 = (size_t)rows * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(oneshot_grid(n4)), dim3(256), 0, st, x, residual,
                     scale_shift, y, n4, C, (flags & EVK_BN_RELU) ? 1 : 0, y_absmax);
  return check_launch("bn_apply");
}


extern "C" int evk_bn_bwd_bits(const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, float* dx, float* d_residual,
                               float* dgamma, float* dbeta, int64_t rows, int32_t C, uint32_t flags, int32_t train,
                               void* workspace, size_t workspace_bytes, uint32_t* dx_absmax, const uint32_t* relu_bits,
                               void* stream) {
  EVK_REQUIRE(dy && x && save_mean && save_invstd && dx, EVK_E_INVALID, "bn_bwd: null pointer");
  // mask of the incoming gradient: the given bits (the forward's ReLU bits, EVK_BN_RELU set — or, without EVK_BN_RELU, the
  // mask of a gradient that arrives unmasked from a residual block's lazy backward); else from the saved output y when
  // given (needed with a residual), else recomputed from x
  const int relu = relu_bits ? 3 : ((flags & EVK_BN_RELU) ? (y ? 1 : 2) : 0);
  EVK_REQUIRE(relu != 2 || !d_residual, EVK_E_INVALID,
              "bn_bwd: a residual branch needs the forward output y (or its ReLU bits) for the mask");
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, EVK_E_UNSUPPORTED, "bn_bwd: rows=%lld C=%d",
              (long long)rows, C);
  EVK_REQUIRE(workspace && workspace_bytes >= evk_bn_workspace_bytes(rows, C), EVK_E_WORKSPACE,
              "bn_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const BnPlan pl = bn_plan(rows, C);
  float* partial = (float*)workspace;
  float* coef = partial + (size_t)kMaxStatBlocks * 2 * C;
  const bool pack = (flags & EVK_BN_PACK_DX) != 0;
  EVK_REQUIRE(!pack || dx_absmax, EVK_E_INVALID, "bn_bwd: EVK_BN_PACK_DX needs dx_absmax (slots zero on entry)");
  float* pmax = pack ? coef + 8 * (size_t)C : nullptr;
  if (pack)
    hipLaunchKernelGGL((bn_bwd_partial_kernel<true>), dim3(pl.nblk), dim3(256), 0, st, dy, x, y, save_mean, save_invstd, gamma, beta,
                       d_residual, partial, rows, C, pl.rows_per_blk, pl.tpc, pl.rl, relu, pmax, relu_bits);
  else
    hipLaunchKernelGGL((bn_bwd_partial_kernel<false>), dim3(pl.nblk), dim3(256), 0, st, dy, x, y, save_mean, save_invstd, gamma, beta,
                       d_residual, partial, rows, C, pl.rows_per_blk, pl.tpc, pl.rl, relu, pmax, relu_bits);
  int rc = check_launch("bn_bwd_partial");
  if (rc) return rc;
  // (two channels x 128 lanes per workgroup, four times the workgroups, measured -0.2 % chosen from 128 partials up and -2.5 %
  // everywhere: the 8-byte pieces of a partial row it reads cost more than the shorter chains save — removed)
  hipLaunchKernelGGL(bn_bwd_final_kernel<kFinCh>, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, st, partial, pl.nblk, C,
                     1.0 / (double)rows, gamma, save_invstd, dgamma, dbeta, coef, train ? 1 : 0, dx_absmax,
                     (const float*)pmax);
  rc = check_launch("bn_bwd_final");
  if (rc) return rc;
  const size_t n4 = (size_t)rows * C / 4;
  // when d_residual holds g already, stage 3 can read it instead of re-masking dy
  const float* gsrc = d_residual ? d_residual : dy;
  const int relu3 = d_residual ? 0 : relu;
  if (pack)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(oneshot_grid(n4)), dim3(256), 0, st, gsrc, x, y,
                       save_mean, save_invstd, coef, gamma, beta, dx, n4, C, relu3, dx_absmax, relu_bits);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(oneshot_grid(n4)), dim3(256), 0, st, gsrc, x, y,
                       save_mean, save_invstd, coef, gamma, beta, dx, n4, C, relu3, dx_absmax, relu_bits);
  return check_launch("bn_bwd_apply");
}

extern "C" int evk_bn_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
                          const float* save_mean, const float* save_invstd, float* dx, float* d_residual,
                          float* dgamma, float* dbeta, int64_t rows, int32_t C, uint32_t flags, int32_t train,
                          void* workspace, size_t workspace_bytes, uint32_t* dx_absmax, void* stream) {
  return evk_bn_bwd_bits(dy, x, y, gamma, beta, save_mean, save_invstd, dx, d_residual, dgamma, dbeta, rows, C, flags, train,
                         workspace, workspace_bytes, dx_absmax, nullptr, stream);
}

// ---- BatchNorm around a fused consumer (FS-Relation, pointwise.hip: relation_bn_*): the statistics records of the
// producing convolution are merged into (mean, invstd, scale, shift) WITHOUT an apply pass — the consumer applies scale /
// shift / ReLU while it reads z — and the backward takes per-workgroup partial sums that the consumer's backward formed
// while it had g and z in registers, instead of a reduce pass of its own.
extern "C" int evk_bn_finalize_parts(const float* parts, int32_t nparts, int32_t C, int64_t rows, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                     float* save_mean, float* save_invstd, float* scale_shift, void* stream) {
  EVK_REQUIRE(parts && nparts > 0 && save_mean && save_invstd && scale_shift, EVK_E_INVALID, "bn_finalize_parts: null pointer");
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, EVK_E_UNSUPPORTED, "bn_finalize_parts: rows=%lld C=%d",
              (long long)rows, C);
  launch_parts_final((hipStream_t)stream, parts, nparts, C, (double)rows, gamma, beta, running_mean, running_var, momentum, eps,
                     save_mean, save_invstd, scale_shift, nullptr, 0);
  return check_launch("bn_parts_final");
}
// dx from g (already masked), x and `nparts` partial records: partial [nparts][2][C] = (sum g, sum g * xhat) and, with
// EVK_BN_PACK_DX, maxima [nparts][2][C] = (max|g|, max|xhat|).  workspace: 16 C floats.
extern "C" int evk_bn_bwd_from_partials(const float* g, const float* x, const float* gamma, const float* save_mean,
                                        const float* save_invstd, const float* partial, const float* maxima, int32_t nparts,
                                        float* dx, float* dgamma, float* dbeta, int64_t rows, int32_t C, uint32_t flags,
                                        int32_t train, void* workspace, size_t workspace_bytes, uint32_t* dx_absmax,
                                        void* stream) {
  EVK_REQUIRE(g && x && save_mean && save_invstd && partial && dx && nparts > 0, EVK_E_INVALID, "bn_bwd_from_partials: null pointer");
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, EVK_E_UNSUPPORTED, "bn_bwd_from_partials: rows=%lld C=%d",
              (long long)rows, C);
  EVK_REQUIRE(workspace && workspace_bytes >= (size_t)16 * C * sizeof(float), EVK_E_WORKSPACE,
              "bn_bwd_from_partials: workspace too small");
  const bool pack = (flags & EVK_BN_PACK_DX) != 0;
  EVK_REQUIRE(!pack || (dx_absmax && maxima), EVK_E_INVALID,
              "bn_bwd_from_partials: EVK_BN_PACK_DX needs dx_absmax (slots zero on entry) and the maxima");
  hipStream_t st = (hipStream_t)stream;
  float* coef = (float*)workspace;
  hipLaunchKernelGGL(bn_bwd_final_kernel<kFinCh>, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, st, partial, nparts, C,
                     1.0 / (double)rows, gamma, save_invstd, dgamma, dbeta, coef, train ? 1 : 0, dx_absmax,
                     pack ? maxima : (const float*)nullptr);
  int rc = check_launch("bn_bwd_final");
  if (rc) return rc;
  const size_t n4 = (size_t)rows * C / 4;
  if (pack)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(oneshot_grid(n4)), dim3(256), 0, st, g, x, (const float*)nullptr, save_mean,
                       save_invstd, coef, gamma, (const float*)nullptr, dx, n4, C, 0, dx_absmax, (const uint32_t*)nullptr);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(oneshot_grid(n4)), dim3(256), 0, st, g, x, (const float*)nullptr, save_mean,
                       save_invstd, coef, gamma, (const float*)nullptr, dx, n4, C, 0, dx_absmax, (const uint32_t*)nullptr);
  return check_launch("bn_bwd_apply");
}

// ---- BatchNorm + ReLU + a narrow 1x1 convolution as one consumer (the decoder's classifier applied per branch,
// module/fpn.py:_forward_commuted; reference fpn.py:163-170,179-193): out[pix][k] = sum_c relu(bn(z))[pix][c] * w[k][c] + b[k]
// for K <= 16 classes.  The normalised map is never written: the forward reads z once; the backward reads z twice (sums,
// then dz) and the K-channel gradient dl, rebuilding g[pix][c] = (y > 0) * sum_k dl[pix][k] w[k][c] in registers — against
// apply (r z, w y), convolution (r y), its data gradient (w g), its weight gradient (r y), BatchNorm reduce (r g, r z) and
// apply (r g, r z, w dz) of the layer-by-layer form: 4 tensor passes instead of 10.
namespace evk {
constexpr int kDotMaxK = 16;
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ f32x4 bn_relu_4(const f32x4 z, const f32x4 sc, const f32x4 sh) {
  f32x4 y = z * sc + sh;
  y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
  return y;
}
// one wave per pixel; a lane owns NCH 16-byte channel chunks whose scale / shift / classifier weights stay in registers
// (re-loading them per pixel — L1 hits, but a vector-memory round trip in front of every row — held the first form to
// 2.3 TB/s); a workgroup walks a contiguous pixel range, two rows of z in flight per wave
template <int NCH, int KT>
__global__ __launch_bounds__(256) void bn_relu_dot_fwd_kernel(const float* __restrict__ z, const float* __restrict__ ss,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ out, size_t npix, int C, int K,
                                                              size_t pix_per_blk) {
  const int c4 = C >> 2, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 sc[NCH], sh[NCH], wk[NCH][KT];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int cb = lane + 64 * j;
    const bool ok = cb < c4;
    sc[j] = ok ? *reinterpret_cast<const f32x4*>(ss + cb * 4) : zero4;
    sh[j] = ok ? *reinterpret_cast<const f32x4*>(ss + C + cb * 4) : zero4;
#pragma unroll
    for (int k = 0; k < KT; ++k) wk[j][k] = (ok && k < K) ? *reinterpret_cast<const f32x4*>(w + (size_t)k * C + cb * 4) : zero4;
  }
  float bk[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) bk[k] = (bias && k < K) ? bias[k] : 0.f;
  const size_t p0 = (size_t)blockIdx.x * pix_per_blk, p1 = p0 + pix_per_blk < npix ? p0 + pix_per_blk : npix;
  for (size_t pix = p0 + wave; pix < p1; pix += 8) {
    f32x4 za[NCH], zb[NCH];
    const bool inb = pix + 4 < p1;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int cb = lane + 64 * j;
      za[j] = cb < c4 ? *reinterpret_cast<const f32x4*>(z + pix * C + cb * 4) : zero4;
      zb[j] = (inb && cb < c4) ? *reinterpret_cast<const f32x4*>(z + (pix + 4) * C + cb * 4) : zero4;
    }
    float da[KT], db[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) da[k] = db[k] = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const f32x4 ya = bn_relu_4(za[j], sc[j], sh[j]), yb = bn_relu_4(zb[j], sc[j], sh[j]);
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        da[k] += ya.x * wk[j][k].x + ya.y * wk[j][k].y + ya.z * wk[j][k].z + ya.w * wk[j][k].w;
        db[k] += yb.x * wk[j][k].x + yb.y * wk[j][k].y + yb.z * wk[j][k].z + yb.w * wk[j][k].w;
      }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < K) {
        const float a = wave_sum_f(da[k]), b = wave_sum_f(db[k]);
        if (lane == 0) {
          out[pix * K + k] = a + bk[k];
          if (inb) out[(pix + 4) * K + k] = b + bk[k];
        }
      }
  }
}
// Per-lane REGISTER accumulators over a workgroup's pixels (a lane owns NCH 16-byte channel chunks: C <= 256 NCH): sum g,
// sum g*xhat, max|g|, max|xhat|, dW[k] = sum_pix y * dl[k]; the four waves are folded through LDS once, at the end
// ([4 waves][4 + KT][C] floats).  (A first form accumulated in LDS per pixel: five dependent read-modify-writes per pixel
// held it to 0.9 TB/s.)  Two pixels per iteration keep two rows of z in flight per wave.
template <int NCH, int KT>
__global__ __launch_bounds__(256) void bn_relu_dot_bwd_partial_kernel(
    const float* __restrict__ dl, const float* __restrict__ z, const float* __restrict__ ss, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ w, float* __restrict__ bnp, float* __restrict__ bnm,
    float* __restrict__ dwp, float* __restrict__ dbp, size_t npix, int C, int K, size_t pix_per_blk) {
  extern __shared__ __attribute__((aligned(16))) float sdot[];
  const int c4 = C >> 2, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int R = 4 + K;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 sg[NCH], sq[NCH], mg[NCH], mx[NCH], dwa[NCH][KT];
  f32x4 sc[NCH], sh[NCH], mu[NCH], is[NCH], wk[NCH][KT];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int cb = lane + 64 * j;
    const bool ok = cb < c4;
    sg[j] = sq[j] = mg[j] = mx[j] = zero4;
    sc[j] = ok ? *reinterpret_cast<const f32x4*>(ss + cb * 4) : zero4;
    sh[j] = ok ? *reinterpret_cast<const f32x4*>(ss + C + cb * 4) : zero4;
    mu[j] = ok ? *reinterpret_cast<const f32x4*>(mean + cb * 4) : zero4;
    is[j] = ok ? *reinterpret_cast<const f32x4*>(invstd + cb * 4) : zero4;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      dwa[j][k] = zero4;
      wk[j][k] = (ok && k < K) ? *reinterpret_cast<const f32x4*>(w + (size_t)k * C + cb * 4) : zero4;
    }
  }
  const size_t p0 = (size_t)blockIdx.x * pix_per_blk, p1 = p0 + pix_per_blk < npix ? p0 + pix_per_blk : npix;
  float db[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) db[k] = 0.f;
  auto one = [&](const f32x4 (&zz)[NCH], const float (&d)[KT]) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const f32x4 y = bn_relu_4(zz[j], sc[j], sh[j]);
      f32x4 g = zero4;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        g += wk[j][k] * d[k];
        dwa[j][k] += y * d[k];
      }
      g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f; g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
      const f32x4 xh = (zz[j] - mu[j]) * is[j];
      sg[j] += g;
      sq[j] += g * xh;
      mg[j].x = fmaxf(mg[j].x, fabsf(g.x)); mg[j].y = fmaxf(mg[j].y, fabsf(g.y));
      mg[j].z = fmaxf(mg[j].z, fabsf(g.z)); mg[j].w = fmaxf(mg[j].w, fabsf(g.w));
      mx[j].x = fmaxf(mx[j].x, fabsf(xh.x)); mx[j].y = fmaxf(mx[j].y, fabsf(xh.y));
      mx[j].z = fmaxf(mx[j].z, fabsf(xh.z)); mx[j].w = fmaxf(mx[j].w, fabsf(xh.w));
    }
  };
  auto fetch = [&](size_t pix, f32x4 (&zz)[NCH], float (&d)[KT]) {
    const bool in = pix < p1;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      d[k] = (in && k < K) ? dl[pix * K + k] : 0.f;
      db[k] += d[k];
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int cb = lane + 64 * j;
      // out of range: z = mean gives xhat = 0, d = 0 gives g = 0 and no dW contribution
      zz[j] = (in && cb < c4) ? *reinterpret_cast<const f32x4*>(z + pix * C + cb * 4) : mu[j];
    }
  };
  for (size_t pix = p0 + wave; pix < p1; pix += 8) {
    f32x4 za[NCH], zb[NCH];
    float da[KT], dbb[KT];
    fetch(pix, za, da);
    fetch(pix + 4, zb, dbb);
    one(za, da);
    one(zb, dbb);
  }
  float* my = sdot + (size_t)wave * R * C;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int cb = lane + 64 * j;
    if (cb < c4) {
      *reinterpret_cast<f32x4*>(my + cb * 4) = sg[j];
      *reinterpret_cast<f32x4*>(my + C + cb * 4) = sq[j];
      *reinterpret_cast<f32x4*>(my + 2 * C + cb * 4) = mg[j];
      *reinterpret_cast<f32x4*>(my + 3 * C + cb * 4) = mx[j];
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K) *reinterpret_cast<f32x4*>(my + (size_t)(4 + k) * C + cb * 4) = dwa[j][k];
    }
  }
  __shared__ float sdb[4][kDotMaxK];
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < KT; ++k) sdb[wave][k] = db[k];   // (every lane of a wave accumulated the same dl values)
  __syncthreads();
  const size_t W = (size_t)R * C, blk = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    auto fold = [&](int k) { return (sdot[(size_t)k * C + c] + sdot[W + (size_t)k * C + c]) + (sdot[2 * W + (size_t)k * C + c] + sdot[3 * W + (size_t)k * C + c]); };
    auto fmx = [&](int k) {
      return fmaxf(fmaxf(sdot[(size_t)k * C + c], sdot[W + (size_t)k * C + c]), fmaxf(sdot[2 * W + (size_t)k * C + c], sdot[3 * W + (size_t)k * C + c]));
    };
    bnp[blk * 2 * C + c] = fold(0);
    bnp[blk * 2 * C + C + c] = fold(1);
    bnm[blk * 2 * C + c] = fmx(2);
    bnm[blk * 2 * C + C + c] = fmx(3);
    for (int k = 0; k < K; ++k) dwp[(blk * K + k) * C + c] = fold(4 + k);
  }
  if (threadIdx.x < K) dbp[blk * K + threadIdx.x] = (sdb[0][threadIdx.x] + sdb[1][threadIdx.x]) + (sdb[2][threadIdx.x] + sdb[3][threadIdx.x]);
}
// dW[k][c] = sum_blk dwp[blk][k][c], dbias[k] = sum_blk dbp[blk][k] (fp64, fixed order): 8 outputs x 32 partial-lanes per
// workgroup, as the other finalisations (a serial walk over up to 2048 partials per thread is latency)
__global__ __launch_bounds__(256) void bn_relu_dot_bwd_wfinal_kernel(const float* __restrict__ dwp, const float* __restrict__ dbp,
                                                                     float* __restrict__ dw, float* __restrict__ dbias, int nblk,
                                                                     int C, int K) {
  __shared__ double red[32][8];
  const int to = threadIdx.x & 7, tl = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + to;           // output index: [0, K*C) weights, [K*C, K*C + K) biases
  const int nw = K * C;
  double s = 0.0;
  if (i < nw) {
    for (int b = tl; b < nblk; b += 32) s += (double)dwp[(size_t)b * nw + i];
  } else if (i < nw + K) {
    for (int b = tl; b < nblk; b += 32) s += (double)dbp[(size_t)b * K + (i - nw)];
  }
  red[tl][to] = s;
  __syncthreads();
  if (tl == 0) {
    for (int k = 1; k < 32; ++k) s += red[k][to];
    if (i < nw) dw[i] = (float)s;
    else if (i < nw + K && dbias) dbias[i - nw] = (float)s;
  }
}
// dz = k0 (g - k1 - xhat k2) with g rebuilt from dl and w; one 16-byte element per thread
template <bool PK>
__global__ __launch_bounds__(256) void bn_relu_dot_bwd_apply_kernel(const float* __restrict__ dl, const float* __restrict__ z,
                                                                    const float* __restrict__ ss, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd,
                                                                    const float* __restrict__ coef, const float* __restrict__ w,
                                                                    float* __restrict__ dz, size_t n4, int C, int K,
                                                                    uint32_t* __restrict__ amax) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = i < n4;
  f32x4 out = {0.f, 0.f, 0.f, 0.f};
  float pk_inv = 1.f;
  if constexpr (PK) pk_inv = op_scale(amax[0]).inv;
  if (valid) {
    const int c4 = C >> 2;
    const size_t pix = i / c4;
    const int c = (int)(i - pix * c4);
    const f32x4 zz = reinterpret_cast<const f32x4*>(z)[i];
    const f32x4 y = zz * reinterpret_cast<const f32x4*>(ss)[c] + reinterpret_cast<const f32x4*>(ss + C)[c];
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) g += reinterpret_cast<const f32x4*>(w + (size_t)k * C)[c] * dl[pix * K + k];
    g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f; g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
    const f32x4 xh = (zz - reinterpret_cast<const f32x4*>(mean)[c]) * reinterpret_cast<const f32x4*>(invstd)[c];
    out = reinterpret_cast<const f32x4*>(coef)[c] * (g - reinterpret_cast<const f32x4*>(coef + C)[c] - xh * reinterpret_cast<const f32x4*>(coef + 2 * C)[c]);
    if constexpr (PK) {
      reinterpret_cast<u32x4*>(dz)[i] = pack_hl4(out, pk_inv);
    } else {
      reinterpret_cast<f32x4*>(dz)[i] = out;
    }
  }
  if constexpr (!PK) {
    if (amax) block_absmax(out, valid, amax);
  }
}
static int dot_blocks(size_t npix) {
  size_t b = (npix + 63) / 64;      // (16 pixels per wave at least: the 64^2 maps get 1024 workgroups, four per CU)
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}
}  // namespace evk

extern "C" int evk_bn_relu_dot_fwd(const float* z, const float* scale_shift, const float* w, const float* bias, float* out,
                                   int64_t rows, int32_t C, int32_t K, void* stream) {
  EVK_REQUIRE(z && scale_shift && w && out, EVK_E_INVALID, "bn_relu_dot_fwd: null pointer");
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && K > 0 && K <= kDotMaxK, EVK_E_UNSUPPORTED, "bn_relu_dot_fwd: C=%d K=%d", C, K);
  int nch = (C / 4 + 63) / 64;
  if (nch == 3) nch = 4;
  const int kt = K <= 1 ? 1 : (K <= 4 ? 4 : 16);
  EVK_REQUIRE((nch == 1) || (nch == 2 && kt <= 4) || (nch == 4 && kt <= 4), EVK_E_UNSUPPORTED,
              "bn_relu_dot_fwd: C=%d K=%d has no instantiation", C, K);
  const int nb = dot_blocks((size_t)rows);
  const size_t ppb = ((size_t)rows + nb - 1) / nb;
  hipStream_t st = (hipStream_t)stream;
#define EVK_DOT_FWD(NCH, KT)                                                                                              \
  hipLaunchKernelGGL((bn_relu_dot_fwd_kernel<NCH, KT>), dim3(nb), dim3(256), 0, st, z, scale_shift, w, bias, out, (size_t)rows, \
                     C, K, ppb)
  if (nch == 1 && kt == 1) EVK_DOT_FWD(1, 1);
  else if (nch == 1 && kt == 4) EVK_DOT_FWD(1, 4);
  else if (nch == 1) EVK_DOT_FWD(1, 16);
  else if (nch == 2 && kt == 1) EVK_DOT_FWD(2, 1);
  else if (nch == 2) EVK_DOT_FWD(2, 4);
  else if (kt == 1) EVK_DOT_FWD(4, 1);
  else EVK_DOT_FWD(4, 4);
#undef EVK_DOT_FWD
  return check_launch("bn_relu_dot_fwd");
}
extern "C" size_t evk_bn_relu_dot_workspace_bytes(int64_t rows, int32_t C, int32_t K) {
  if (rows <= 0 || C <= 0 || K <= 0) return 0;
  const size_t nb = (size_t)dot_blocks((size_t)rows);
  return (nb * (size_t)(4 + K) * C + nb * K + (size_t)16 * C) * sizeof(float);
}
// dz (the BatchNorm input's gradient; EVK_BN_PACK_DX: packed, dx_absmax zero on entry), dgamma, dbeta, dw [K][C], dbias [K]
extern "C" int evk_bn_relu_dot_bwd(const float* dl, const float* z, const float* scale_shift, const float* gamma,
                                   const float* save_mean, const float* save_invstd, const float* w, float* dz, float* dgamma,
                                   float* dbeta, float* dw, float* dbias, int64_t rows, int32_t C, int32_t K, uint32_t flags,
                                   void* workspace, size_t workspace_bytes, uint32_t* dx_absmax, void* stream) {
  EVK_REQUIRE(dl && z && scale_shift && save_mean && save_invstd && w && dz && dw, EVK_E_INVALID, "bn_relu_dot_bwd: null pointer");
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 1024 && K > 0 && K <= kDotMaxK, EVK_E_UNSUPPORTED,
              "bn_relu_dot_bwd: C=%d K=%d", C, K);
  EVK_REQUIRE(workspace && workspace_bytes >= evk_bn_relu_dot_workspace_bytes(rows, C, K), EVK_E_WORKSPACE,
              "bn_relu_dot_bwd: workspace too small");
  const bool pack = (flags & EVK_BN_PACK_DX) != 0;
  EVK_REQUIRE(!pack || dx_absmax, EVK_E_INVALID, "bn_relu_dot_bwd: EVK_BN_PACK_DX needs dx_absmax (slots zero on entry)");
  hipStream_t st = (hipStream_t)stream;
  const int nb = dot_blocks((size_t)rows);
  const size_t ppb = ((size_t)rows + nb - 1) / nb;
  float* bnp = (float*)workspace;
  float* bnm = bnp + (size_t)nb * 2 * C;
  float* dwp = bnm + (size_t)nb * 2 * C;
  float* dbp = dwp + (size_t)nb * K * C;
  float* coef = dbp + (size_t)nb * K;
  const size_t lds = (size_t)4 * (4 + K) * C * sizeof(float);
  EVK_REQUIRE(lds <= 64 * 1024, EVK_E_UNSUPPORTED, "bn_relu_dot_bwd: C=%d K=%d do not fit the LDS", C, K);
  int nch = (C / 4 + 63) / 64;
  if (nch == 3) nch = 4;
  const int kt = K <= 1 ? 1 : (K <= 4 ? 4 : 16);
  EVK_REQUIRE((nch == 1) || (nch == 2 && kt <= 4) || (nch == 4 && kt <= 4), EVK_E_UNSUPPORTED,
              "bn_relu_dot_bwd: C=%d K=%d has no instantiation", C, K);
#define EVK_DOT_PARTIAL(NCH, KT)                                                                                          \
  hipLaunchKernelGGL((bn_relu_dot_bwd_partial_kernel<NCH, KT>), dim3(nb), dim3(256), lds, st, dl, z, scale_shift, save_mean, \
                     save_invstd, w, bnp, bnm, dwp, dbp, (size_t)rows, C, K, ppb)
  if (nch == 1 && kt == 1) EVK_DOT_PARTIAL(1, 1);
  else if (nch == 1 && kt == 4) EVK_DOT_PARTIAL(1, 4);
  else if (nch == 1) EVK_DOT_PARTIAL(1, 16);
  else if (nch == 2 && kt == 1) EVK_DOT_PARTIAL(2, 1);
  else if (nch == 2) EVK_DOT_PARTIAL(2, 4);
  else if (kt == 1) EVK_DOT_PARTIAL(4, 1);
  else EVK_DOT_PARTIAL(4, 4);
#undef EVK_DOT_PARTIAL
  int rc = check_launch("bn_relu_dot_bwd_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_final_kernel<kFinCh>, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, st, (const float*)bnp, nb, C,
                     1.0 / (double)rows, gamma, save_invstd, dgamma, dbeta, coef, 1, dx_absmax,
                     pack ? (const float*)bnm : (const float*)nullptr);
  rc = check_launch("bn_bwd_final");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_relu_dot_bwd_wfinal_kernel, dim3((K * C + K + 7) / 8), dim3(256), 0, st, (const float*)dwp,
                     (const float*)dbp, dw, dbias, nb, C, K);
  rc = check_launch("bn_relu_dot_bwd_wfinal");
  if (rc) return rc;
  const size_t n4 = (size_t)rows * C / 4;
  if (pack)
    hipLaunchKernelGGL(bn_relu_dot_bwd_apply_kernel<true>, dim3(oneshot_grid(n4)), dim3(256), 0, st, dl, z, scale_shift, save_mean,
                       save_invstd, (const float*)coef, w, dz, n4, C, K, dx_absmax);
  else
    hipLaunchKernelGGL(bn_relu_dot_bwd_apply_kernel<false>, dim3(oneshot_grid(n4)), dim3(256), 0, st, dl, z, scale_shift, save_mean,
                       save_invstd, (const float*)coef, w, dz, n4, C, K, dx_absmax);
  return check_launch("bn_relu_dot_bwd_apply");
}

// ---- ReLU bits (common.hpp: relu_bits_*)
namespace evk {
__global__ __launch_bounds__(256) void relu_bits_apply_kernel(const float* __restrict__ g, const uint32_t* __restrict__ bits,
                                                              float* __restrict__ out, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) reinterpret_cast<f32x4*>(out)[i] = relu_bits_mask(reinterpret_cast<const f32x4*>(g)[i], bits, i);
}
}  // namespace evk
extern "C" size_t evk_relu_bits_bytes(int64_t n) { return n > 0 ? relu_bits_words(((size_t)n + 3) / 4) * sizeof(uint32_t) : 0; }
// out = g where the bit is set, 0 elsewhere (materialises a lazily masked gradient for a reader that cannot take the bits)
extern "C" int evk_relu_bits_apply(const float* g, const uint32_t* bits, float* out, int64_t n, void* stream) {
  EVK_REQUIRE(g && bits && out && n > 0 && n % 4 == 0, EVK_E_INVALID, "relu_bits_apply: null pointer or n %% 4 != 0");
  const size_t n4 = (size_t)n / 4;
  hipLaunchKernelGGL(relu_bits_apply_kernel, dim3(oneshot_grid(n4)), dim3(256), 0, (hipStream_t)stream, g, bits, out, n4);
  return check_launch("relu_bits_apply");
}

// ------------------------------------------------------------------------------------------------
// Staged entry points for synchronized BatchNorm (torch.nn.SyncBatchNorm under
// `sync_bn=True`, reference ever/trainer/th_ddp_trainer.py): the same kernels as above with the
// cross-rank exchange left to the caller between the stages (one small all-gather forward, one small
// all-reduce backward, on torch.distributed / RCCL).
namespace evk {

// stats[c] = local mean, stats[C + c] = local sum of squared deviations from it (fp64)
__global__ __launch_bounds__(256) void bn_local_final_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                             int nblk, int C, double rows, double* __restrict__ stats) {
  int c;
  double s, q;
  if (!reduce_partials(partial, nblk, C, c, s, q)) return;
  const double dm = s / rows;  // mean of (x - pivot)
  stats[c] = (double)x[c] + dm;
  double m2 = q - s * dm;
  stats[C + c] = m2 > 0.0 ? m2 : 0.0;
}
__global__ void bn_coef_from_stats_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                          const float* __restrict__ mean, const float* __restrict__ invstd, int C,
                                          float* __restrict__ scale_shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = (gamma ? gamma[c] : 1.f) * invstd[c];
  scale_shift[c] = sc;
  scale_shift[C + c] = (beta ? beta[c] : 0.f) - mean[c] * sc;
}
__global__ __launch_bounds__(256) void bn_bwd_sums_final_kernel(const float* __restrict__ partial, int nblk, int C,
                                                                double* __restrict__ sums) {
  int c;
  double s, q;
  if (!reduce_partials(partial, nblk, C, c, s, q)) return;
  sums[c] = s;
  sums[C + c] = q;
}
__global__ void bn_bwd_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ invstd,
                                   const float* __restrict__ mean_g, const float* __restrict__ mean_gx, int C,
                                   float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  coef[c] = (gamma ? gamma[c] : 1.f) * invstd[c];
  coef[C + c] = mean_g[c];
  coef[2 * C + c] = mean_gx[c];
}

}  // namespace evk

static int bn_stage_check(const char* what, int64_t rows, int32_t C, const void* ws, size_t ws_bytes) {
  EVK_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 2048, EVK_E_UNSUPPORTED, "%s: rows=%lld C=%d", what, (long long)rows,
              C);
  EVK_REQUIRE(ws && ws_bytes >= evk_bn_workspace_bytes(rows, C), EVK_E_WORKSPACE, "%s: workspace too small", what);
  return EVK_OK;
}

extern "C" int evk_bn_local_stats(const float* x, double* stats, int64_t rows, int32_t C, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(x && stats, EVK_E_INVALID, "bn_local_stats: null pointer");
  int rc = bn_stage_check("bn_local_stats", rows, C, workspace, workspace_bytes);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const BnPlan pl = bn_plan(rows, C);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(pl.nblk), dim3(256), 0, st, x, partial, rows, C, pl.rows_per_blk,
                     pl.tpc, pl.rl);
  hipLaunchKernelGGL(bn_local_final_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, st, x, (const float*)partial,
                     pl.nblk, C, (double)rows, stats);
  return check_launch("bn_local_stats");
}

extern "C" int evk_bn_apply_stats(const float* x, const float* residual, const float* gamma, const float* beta,
                                  const float* mean, const float* invstd, float* y, int64_t rows, int32_t C,
                                  uint32_t flags, void* workspace, size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(x && y && mean && invstd, EVK_E_INVALID, "bn_apply_stats: null pointer");
  int rc = bn_stage_check("bn_apply_stats", rows, C, workspace, workspace_bytes);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  float* scale_shift = (float*)workspace;
  hipLaunchKernelGGL(bn_coef_from_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, mean, invstd, C,
                     scale_shift);
  const size_t n4 = (size_t)rows * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(oneshot_grid(n4)), dim3(256), 0, st, x, residual,
                     (const float*)scale_shift, y, n4, C, (flags & EVK_BN_RELU) ? 1 : 0, (uint32_t*)nullptr);
  return check_launch("bn_apply_stats");
}

extern "C" int evk_bn_bwd_local_sums(const float* dy, const float* x, const float* y, const float* gamma,
                                     const float* beta, const float* mean, const float* invstd, float* d_residual,
                                     double* sums, int64_t rows, int32_t C, uint32_t flags, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(dy && x && mean && invstd && sums, EVK_E_INVALID, "bn_bwd_local_sums: null pointer");
  const int relu = (flags & EVK_BN_RELU) ? (y ? 1 : 2) : 0;
  EVK_REQUIRE(relu != 2 || !d_residual, EVK_E_INVALID, "bn_bwd_local_sums: a residual branch needs y for the ReLU mask");
  int rc = bn_stage_check("bn_bwd_local_sums", rows, C, workspace, workspace_bytes);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const BnPlan pl = bn_plan(rows, C);
  float* partial = (float*)workspace;
  hipLaunchKernelGGL(bn_bwd_partial_kernel<false>, dim3(pl.nblk), dim3(256), 0, st, dy, x, y, mean, invstd, gamma, beta,
                     d_residual, partial, rows, C, pl.rows_per_blk, pl.tpc, pl.rl, relu, (float*)nullptr);
  hipLaunchKernelGGL(bn_bwd_sums_final_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(256), 0, st, (const float*)partial,
                     pl.nblk, C, sums);
  return check_launch("bn_bwd_local_sums");
}

extern "C" int evk_bn_bwd_apply_sums(const float* dy, const float* x, const float* y, const float* gamma,
                                     const float* beta, const float* mean, const float* invstd, const float* mean_g,
                                     const float* mean_gx, float* dx, int64_t rows, int32_t C, uint32_t flags,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  EVK_REQUIRE(dy && x && mean && invstd && mean_g && mean_gx && dx, EVK_E_INVALID, "bn_bwd_apply_sums: null pointer");
  const int relu = (flags & EVK_BN_RELU) ? (y ? 1 : 2) : 0;
  int rc = bn_stage_check("bn_bwd_apply_sums", rows, C, workspace, workspace_bytes);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  float* coef = (float*)workspace;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, invstd, mean_g, mean_gx, C, coef);
  const size_t n4 = (size_t)rows * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(oneshot_grid(n4)), dim3(256), 0, st, dy, x, y, mean,
                     invstd, (const float*)coef, gamma, beta, dx, n4, C, relu, (uint32_t*)nullptr);
  return check_launch("bn_bwd_apply_sums");
}
