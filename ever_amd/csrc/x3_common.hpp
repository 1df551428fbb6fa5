// Shared pieces of the bf16-split ("x3") MFMA kernels: exact 3-term bf16 splitting of fp32 operands and
// the LDS plane geometry.  See conv_igemm_x3.hip for the arithmetic argument.
#pragma once
#include "common.hpp"

namespace evk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK3 = 32;             // fp32 K elements per step = two 32x32x16 MFMA k-blocks
constexpr int kRowBytes = BK3 * 2;  // one LDS row of one bf16 plane

// the six retained partial products, smallest magnitude first: (A part, B part); 0 = h, 1 = m, 2 = l
__device__ constexpr int kPA[6] = {2, 0, 1, 1, 0, 0};
__device__ constexpr int kPB[6] = {0, 2, 1, 0, 1, 0};

// exact 3-term bf16 split of two floats (x = h + m + l, round-to-nearest-even at each level);
// each result packs the pair, first element in the low half
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& H, uint32_t& M, uint32_t& L) {
  const f32x2 v = {x0, x1};
  H = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {x0 - __builtin_bit_cast(float, H << 16), x1 - __builtin_bit_cast(float, H & 0xffff0000u)};
  M = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
  const f32x2 s = {r.x - __builtin_bit_cast(float, M << 16), r.y - __builtin_bit_cast(float, M & 0xffff0000u)};
  L = __builtin_bit_cast(uint32_t, __builtin_convertvector(s, bf16x2));
}

// NP = number of bf16 planes per operand: 3 = the exact split above (six partial products, fp32-grade);
// 1 = plain bf16 (operands rounded to bf16 once, ONE product, fp32 accumulate): the counterpart of the reference's
// `--mixed_precision bf16` (launcher.py:40-80), selected by the *_bf16 entry points.  LDS / HBM plane layouts are
// the same; only plane 0 (h) is produced and read.
template <int NP> struct X3Prod { static constexpr int N = NP == 3 ? 6 : 1; };
__device__ __forceinline__ constexpr int x3_pa(int np, int t) { return np == 3 ? kPA[t] : 0; }
__device__ __forceinline__ constexpr int x3_pb(int np, int t) { return np == 3 ? kPB[t] : 0; }
// two floats -> packed bf16 pair (round-to-nearest-even), first element in the low half
__device__ __forceinline__ uint32_t cvt2(float x0, float x1) {
  const f32x2 v = {x0, x1};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// byte offset of 16-byte chunk c16 (0..3) of row `row` inside one plane; the XOR spreads the
// ds_read_b128 / ds_write_b128 of 16 consecutive rows over 16 distinct 16-byte slots of a 256-byte bank row
__device__ __forceinline__ int plane_off(int row, int c16) { return row * kRowBytes + ((c16 ^ ((row >> 2) & 3)) << 4); }

}  // namespace evk
