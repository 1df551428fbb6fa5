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

// NP selects the arithmetic of a kernel instantiation (the LDS / HBM plane geometry is the same for all three):
//   3  exact 3-term bf16 split above: six partial products, error per product < 2^-24 ("bf16x3", fp32 grade)
//   2  2-term fp16 split of the operand scaled by a power of two ("f16x2"): x / s = h + l with h, l fp16 (11-bit
//      significands, round-to-nearest-even at both levels, |x/s - h - l| <= 2^-22 |x/s|), three partial products
//      l*wh + h*wl + h*wh (dropped: l*wl <= 2^-22 |x*w|) on v_mfma_f32_32x32x16_f16, the accumulator multiplied by the
//      two operands' scales at the end.  s = 2^(floor(log2 max|x|) - 13) per TENSOR (evk_absmax), so the largest
//      element lands in [2^13, 2^14): every element down to 2^-17 of the largest keeps its 22 bits, smaller ones an
//      absolute 2^-38 of the largest; products cannot overflow fp32 (2^28 * K).  Only planes 0 (h) and 1 (l) exist.
//   1  plain bf16 (operands rounded to bf16 once, ONE product, fp32 accumulate): the counterpart of the reference's
//      `--mixed_precision bf16` (launcher.py:40-80), selected by the *_bf16 entry points.  Only plane 0 exists.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int NP> struct X3Prod { static constexpr int N = NP == 3 ? 6 : (NP == 2 ? 3 : 1); };
__device__ constexpr int kHA[3] = {1, 0, 0};   // f16x2: (A part, B part), smallest magnitude first; 0 = h, 1 = l
__device__ constexpr int kHB[3] = {0, 1, 0};
__device__ __forceinline__ constexpr int x3_pa(int np, int t) { return np == 3 ? kPA[t] : (np == 2 ? kHA[t] : 0); }
__device__ __forceinline__ constexpr int x3_pb(int np, int t) { return np == 3 ? kPB[t] : (np == 2 ? kHB[t] : 0); }
// two floats -> packed bf16 pair (round-to-nearest-even), first element in the low half
__device__ __forceinline__ uint32_t cvt2(float x0, float x1) {
  const f32x2 v = {x0, x1};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// 2-term fp16 split of two (already scaled) floats; each result packs the pair, first element in the low half
__device__ __forceinline__ void split2h(float x0, float x1, uint32_t& H, uint32_t& L) {
  const f32x2 v = {x0, x1};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);   // exact: at most 13 significant bits remain
  H = __builtin_bit_cast(uint32_t, h);
  L = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}
// the planes of one pair of operand values (M is the l plane of the f16x2 arithmetic); `inv` = 1 / operand scale (NP == 2)
template <int NP>
__device__ __forceinline__ void split_np(float x0, float x1, float inv, uint32_t& H, uint32_t& M, uint32_t& L) {
  if constexpr (NP == 3) split2(x0, x1, H, M, L);
  else if constexpr (NP == 2) split2h(x0 * inv, x1 * inv, H, M);
  else H = cvt2(x0, x1);
}
// "Packed" activation operands of the f16x2 arithmetic: the producer (evk_pack_f16x2, the BatchNorm apply passes) already
// split x / s and stored ONE 32-bit word per element, h in the low half and l in the high half — same addressing as the
// fp32 tensor, the consumer's staging is two v_perm_b32 per element pair instead of scale, 2 x convert, subtract, convert
// (the staging VALU was the bound of the weight gradient: 1292 -> 944 us on 3x3x256 @128^2 x16 with packed operands).
// Kernels take the mode as NPX: 1, 2, 3 = NP with fp32 operands, 4 = NP 2 with a packed activation operand.
template <int NPX> struct X3Mode {
  static constexpr int NP = NPX == 4 ? 2 : NPX;
  static constexpr bool PK = NPX == 4;
};
__device__ __forceinline__ uint32_t pack_hl(float xs) {   // xs = x / s; the same two roundings as split2h
  const _Float16 h = (_Float16)xs;
  const _Float16 l = (_Float16)(xs - (float)h);
  return (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
}
// two elements at once: the pair split of the kernels (packed converts), then one byte permute per word
__device__ __forceinline__ void pack_hl2(float xs0, float xs1, uint32_t& w0, uint32_t& w1) {
  uint32_t H, L;
  split2h(xs0, xs1, H, L);
  w0 = __builtin_amdgcn_perm(L, H, 0x05040100u);
  w1 = __builtin_amdgcn_perm(L, H, 0x07060302u);
}
__device__ __forceinline__ u32x4 pack_hl4(const f32x4 v, float inv) {
  u32x4 w;
  uint32_t a, b;
  pack_hl2(v.x * inv, v.y * inv, a, b); w.x = a; w.y = b;
  pack_hl2(v.z * inv, v.w * inv, a, b); w.z = a; w.w = b;
  return w;
}
__device__ __forceinline__ float unpack_hl(uint32_t w) {  // h + l (22 bits), still in units of s
  return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)) + (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
}
template <int NP, bool PK>
__device__ __forceinline__ void split_op(float x0, float x1, float inv, uint32_t& H, uint32_t& M, uint32_t& L) {
  if constexpr (PK) {
    static_assert(NP == 2, "packed operands exist for the f16x2 arithmetic only");
    const uint32_t w0 = __builtin_bit_cast(uint32_t, x0), w1 = __builtin_bit_cast(uint32_t, x1);
    H = __builtin_amdgcn_perm(w1, w0, 0x05040100u);
    M = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
  } else {
    split_np<NP>(x0, x1, inv, H, M, L);
  }
}
// one MFMA of the instantiation's operand type on two 16-byte LDS fragments
template <int NP>
__device__ __forceinline__ f32x16 mfma_np(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (NP == 2)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// An ACTIVATION operand's max|x| lives in kAmaxSlots words spaced one cache line apart (evk_absmax fills slot 0 and zeroes
// the rest; the BatchNorm apply passes raise slot (workgroup & 63) with one atomic per workgroup — 64 addresses on 64
// lines take a 268 MB map's 65 K atomics without a measurable cost, where ONE hot word serialised the eight XCDs).  A
// wave reads one slot per lane and folds them; every wave of a kernel gets the same value.  Weights keep a single word.
constexpr int kAmaxSlots = 64, kAmaxStride = 32, kAmaxWords = kAmaxSlots * kAmaxStride;
__device__ __forceinline__ uint32_t act_absmax(const uint32_t* __restrict__ slots) {
  uint32_t m = slots[(threadIdx.x & 63) * kAmaxStride];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  return m;
}

// Operand scale of the f16x2 arithmetic from the bit image of max|x| (what evk_absmax leaves in device memory):
// s = 2^(E - 13) with E the biased exponent, clamped to the normal range; zero / non-finite tensors use s = 1.
struct OpScale { float inv, s; };
__device__ __forceinline__ OpScale op_scale(uint32_t absmax_bits) {
  const int E = (int)((absmax_bits >> 23) & 0xff);
  int f = E - 13;
  if (E == 0 || E == 255) f = 127;
  f = f < 1 ? 1 : f;
  OpScale o;
  o.s = __builtin_bit_cast(float, (uint32_t)f << 23);
  o.inv = __builtin_bit_cast(float, (uint32_t)(254 - f) << 23);
  return o;
}

// byte offset of 16-byte chunk c16 (0..3) of row `row` inside one plane; the XOR spreads the
// ds_read_b128 / ds_write_b128 of 16 consecutive rows over 16 distinct 16-byte slots of a 256-byte bank row
__device__ __forceinline__ int plane_off(int row, int c16) { return row * kRowBytes + ((c16 ^ ((row >> 2) & 3)) << 4); }
// The weight-gradient kernels' form: their staging waves write 16 bytes per lane with CONSECUTIVE LANES ON CONSECUTIVE
// ROWS, and ds_write_b128 is served in groups of 8 contiguous lanes against 32 banks (a 128-byte window, guide table):
// under plane_off rows r and r + 2 of a group land on the same banks (PMC: SQ_LDS_BANK_CONFLICT = half the LDS cycles of
// the staging waves).  XOR-ing with ((row >> 1) ^ (row >> 2)) & 3 gives the 4 even and the 4 odd rows of every 8-row
// group four different slots each, and still spreads the ds_read_b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}
// over 16 distinct 16-byte slots of a 256-byte bank row (exhaustive check: tools/probes/lds_swizzle.py).  Measured:
// neutral in time (3x3x256 @128^2 1288 -> 1285 us) — a 2-way store conflict costs 16 LDS cycles against the 13 the
// VGPR -> LDS transfer of a ds_write_b128 takes anyway; kept because it is the layout the counters call clean.
__device__ __forceinline__ int wg_off(int row, int c16) {
  return row * kRowBytes + ((c16 ^ (((row >> 1) ^ (row >> 2)) & 3)) << 4);
}

}  // namespace evk
