// Shared pieces of the weight-gradient kernels (fp32 MFMA and 3-way-bf16-split MFMA).
#pragma once
#include "common.hpp"

namespace evk {

struct WGradArgs {
  const float* x;
  const float* dy;
  float* out;  // dw (splitk==1) or workspace [splitk][Cout][Ktot]
  int N, H, W, Cin, Ho, Wo, Cout;
  int kh, kw, cpt;
  int sh, sw, ph, pw, dh, dw;
  int M, Ktot;
  int chunk;  // pixels per split (multiple of 32)
  int tiles_co, tiles_k, splitk;
  FastDiv fd_hw, fd_w;
  int planes;  // 3 (0 means 3): exact split, six products; 1: plain bf16 operands, one product (evk_conv2d_wgrad_bf16);
               // 2: 2-term fp16 split of scaled operands, three products (evk_conv2d_wgrad_f16x2)
  const uint32_t* x_scale;   // planes == 2: bit images of max|x| and max|dy| (evk_absmax)
  const uint32_t* dy_scale;
  int x_packed, dy_packed;   // planes == 2: the operand holds packed (h | l << 16) words of value / s (x3_common.hpp)
  int planar;                // planes == 2: BOTH operands are planar fp16 pairs (conv_wgrad_tr.hip)
  int reserved;  // (was: run-time ablation switches; they are compile-time now, EVK_WG_ABL)
};

constexpr int BKP = 32;  // pixels per step

struct WGradPlan {
  int bm, bn, tiles_co, tiles_k, splitk, chunk;
  int ws;  // 1: wave-specialised 128x256 kernel (conv_wgrad_x3ws.hip)
};
// x3 = 1: plan for the bf16-split kernel (different LDS footprint => different residency)
// tr = 1 / 2: the planar-operand kernel (conv_wgrad_tr.hip), one workgroup per CU: 128 x 256 tile / nine-tap form
// (128 output channels x 9 taps x 64 input channels per tile)
WGradPlan plan_wgrad(const evk_conv_desc* d, int x3, int planes = 3, int tr = 0, int shared = 0);
int launch_wgrad_x3(const WGradArgs& a, const WGradPlan& pl, hipStream_t stream);
int launch_wgrad_x3ws(const WGradArgs& a, hipStream_t stream);
bool wgrad_tr_applicable(const WGradArgs& a);
bool wgrad_tr_nine_tap(const evk_conv_desc* d);
int launch_wgrad_tr(const WGradArgs& a, int nine_tap, hipStream_t stream);

}  // namespace evk
