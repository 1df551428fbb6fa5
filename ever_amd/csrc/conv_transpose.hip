// nn.ConvTranspose2d on the implicit-GEMM kernels.
//
// A transposed convolution with weight w[Cin_t][kh][kw][Cout_t] is the ADJOINT (data gradient) of the convolution
// C: u[N,Ho_t,Wo_t,Cout_t] -> z[N,H_t,W_t,Cin_t] with that same weight read as OHWI (O = Cin_t, I = Cout_t):
//     forward   y  = C^T x          = evk_conv2d_dgrad   (the residue-class kernel: no MFMA on structurally-zero taps)
//     d/dx      dx = C  g_y         = evk_conv2d_fwd
//     d/dw      dw = wgrad_C(u := g_y, dz := x);   d/dbias = column sums of g_y
// Every entry point takes the descriptor `d` of C: (N, H, W, Cin) = the transposed convolution's OUTPUT size and channel
// count, (Ho, Wo, Cout) = its INPUT.  Nothing here is a new kernel: the reference's path has no transposed convolution
// (SURVEY 2.3), BASELINE.json's north_star names it, and the strided data-gradient already is it.
#include "common.hpp"

namespace evk {

__global__ void bias_rows_kernel(float* __restrict__ y, const float* __restrict__ bias, size_t n4, int c4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = reinterpret_cast<f32x4*>(y)[i];
  v += reinterpret_cast<const f32x4*>(bias)[i % c4];
  reinterpret_cast<f32x4*>(y)[i] = v;
}

int launch_colsum(const float* src, float* out, int64_t rows, int c, float* workspace, hipStream_t st);  // conv_wgrad.hip
size_t colsum_workspace_bytes(int64_t rows, int c);

static int add_bias(const evk_conv_desc* d, float* y, const float* bias, hipStream_t st) {
  if (!bias) return EVK_OK;
  EVK_REQUIRE((d->Cin & 3) == 0, EVK_E_UNSUPPORTED, "conv_transpose2d: output channels (%d) must be a multiple of 4", d->Cin);
  const size_t n4 = (size_t)d->N * d->H * d->W * d->Cin / 4;
  hipLaunchKernelGGL(bias_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, y, bias, n4, d->Cin / 4);
  return check_launch("conv_transpose2d bias");
}

}  // namespace evk

using namespace evk;

extern "C" int evk_conv_transpose2d_fwd(const evk_conv_desc* d, const float* x, const float* wt, const float* bias, float* y,
                                        void* stream) {
  const int rc = evk_conv2d_dgrad(d, x, wt, nullptr, y, stream);
  return rc ? rc : add_bias(d, y, bias, (hipStream_t)stream);
}

extern "C" int evk_conv_transpose2d_fwd_x3(const evk_conv_desc* d, const float* x, const void* wsplit_t, const float* bias,
                                           float* y, void* stream) {
  const int rc = evk_conv2d_dgrad_x3(d, x, wsplit_t, nullptr, y, stream);
  return rc ? rc : add_bias(d, y, bias, (hipStream_t)stream);
}

extern "C" int evk_conv_transpose2d_dgrad(const evk_conv_desc* d, const float* dy, const float* w, float* dx, void* stream) {
  return evk_conv2d_fwd(d, dy, w, nullptr, dx, 0, stream);
}

extern "C" int evk_conv_transpose2d_dgrad_x3(const evk_conv_desc* d, const float* dy, const void* wsplit, float* dx,
                                             void* stream) {
  return evk_conv2d_fwd_x3(d, dy, wsplit, nullptr, dx, 0, stream);
}

extern "C" size_t evk_conv_transpose2d_wgrad_workspace_bytes(const evk_conv_desc* d, int32_t x3) {
  if (!d) return 0;
  const size_t a = x3 ? evk_conv2d_wgrad_x3_workspace_bytes(d) : evk_conv2d_wgrad_workspace_bytes(d);
  const size_t b = colsum_workspace_bytes((int64_t)d->N * d->H * d->W, d->Cin);
  return a > b ? a : b;
}

static int ct_wgrad(const evk_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias, void* workspace,
                    size_t workspace_bytes, void* stream, bool x3) {
  EVK_REQUIRE(d && x && dy && workspace, EVK_E_INVALID, "conv_transpose2d_wgrad: null pointer");
  EVK_REQUIRE(workspace_bytes >= evk_conv_transpose2d_wgrad_workspace_bytes(d, x3 ? 1 : 0), EVK_E_WORKSPACE,
              "conv_transpose2d_wgrad: workspace too small");
  int rc = EVK_OK;
  if (dw)
    rc = x3 ? evk_conv2d_wgrad_x3(d, dy, x, dw, nullptr, workspace, workspace_bytes, stream)
            : evk_conv2d_wgrad(d, dy, x, dw, nullptr, workspace, workspace_bytes, stream);
  if (rc || !dbias) return rc;
  return launch_colsum(dy, dbias, (int64_t)d->N * d->H * d->W, d->Cin, (float*)workspace, (hipStream_t)stream);
}

extern "C" int evk_conv_transpose2d_wgrad(const evk_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  return ct_wgrad(d, x, dy, dw, dbias, workspace, workspace_bytes, stream, false);
}

extern "C" int evk_conv_transpose2d_wgrad_x3(const evk_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                             void* workspace, size_t workspace_bytes, void* stream) {
  return ct_wgrad(d, x, dy, dw, dbias, workspace, workspace_bytes, stream, true);
}
