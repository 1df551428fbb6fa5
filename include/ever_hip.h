/*
 * ever_hip.h — C-ABI of libever_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * EVer hot path  ResNet encoder -> FPN / FS-Relation / decoder -> per-pixel head -> pixel loss,
 * forward + backward.
 *
 * The reference (Z-Zheng/ever 0.5.6) has no FFI: its device boundary is the set of ATen ops its
 * torch.nn modules issue.  Every entry point below cites the reference
 * call site (file:line under the reference tree) whose ATen op it replaces.
 *
 * Conventions
 *   - All activations are dense fp32 **NHWC** (pixel-major, channel-minor) in HBM.  A logical
 *     [N,C,H,W] torch tensor with channels_last strides is exactly this layout.
 *   - Convolution weights are dense fp32 **OHWI** ([Cout][kh][kw][Cin]; an OIHW torch tensor with
 *     channels_last strides).  Gradients are produced in the same layout.
 *   - Pointers are raw device pointers (hipMalloc / torch caching allocator).  The library
 *     allocates nothing: scratch is caller-provided, sizes come from the *_workspace_bytes query.
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on that stream,
 *     is re-entrant across streams and never synchronises.
 *   - Return value: 0 = ok, <0 = error (EVK_E_*); evk_last_error() returns a thread-local text.
 *     No C++ exception crosses this boundary.
 */
#ifndef EVER_HIP_H_
#define EVER_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVK_OK 0
#define EVK_E_INVALID (-1)     /* bad argument (null pointer, non-positive dim, misaligned channels) */
#define EVK_E_UNSUPPORTED (-2) /* shape outside what the kernels implement */
#define EVK_E_LAUNCH (-3)      /* hipLaunch / runtime failure; see evk_last_error() */
#define EVK_E_WORKSPACE (-4)   /* caller workspace too small */

const char* evk_last_error(void);
/* library ABI version (bumped on any signature change) and the gfx arch it was built for */
int evk_abi_version(void);
const char* evk_build_arch(void);

/* ------------------------------------------------------------------ convolution ------------ */
/* Geometry of one 2-D convolution, square kernel/stride/dilation not assumed.
 * Replaces nn.Conv2d at: _resnets.py:21-29 (3x3 / 1x1), _resnets.py:149 (7x7 stem),
 * fpn.py:23-37,72-73 (FPN lateral / output), fs_relation.py:23-53 (scene MLP, content /
 * re-encode 1x1 + bias), fpn.py:165 (decoder 3x3), fpn.py:179 (classifier + bias). */
typedef struct evk_conv_desc {
  int32_t N, H, W, Cin;   /* input  x: [N,H,W,Cin]   (Cin % 4 == 0; pad channels with evk_pad_channels) */
  int32_t Ho, Wo, Cout;   /* output y: [N,Ho,Wo,Cout] */
  int32_t kh, kw;
  int32_t stride_h, stride_w;
  int32_t pad_h, pad_w;
  int32_t dil_h, dil_w;
} evk_conv_desc;

/* flags for evk_conv2d_fwd */
#define EVK_CONV_RELU 1u /* y = max(y, 0) in the epilogue */
/* f16x2 entry points only: that activation operand is stored PACKED (evk_pack_f16x2: one 32-bit word per element) */
#define EVK_CONV_X_PACKED 2u
#define EVK_CONV_DY_PACKED 4u
/* PLANAR operands of the f16x2 arithmetic (round 3): the same (h, l) pair of value / s, stored as two fp16 planes —
 * H[M][C] followed by L[M][C] in one allocation of the fp32 tensor's size (evk_pack_planar_f16x2).  The weight gradient
 * then stages nothing: both operands go global -> LDS by DMA and the fragments come from transposing LDS reads
 * (csrc/conv_wgrad_tr.hip).  Needs both flags, Cin % 64 == 0, Cout % 64 == 0, dbias == NULL. */
#define EVK_CONV_X_PLANAR 8u
#define EVK_CONV_DY_PLANAR 16u
/* evk_conv2d_wgrad_f16x2_ex only (round 5): the launch runs BESIDE other work on another stream (the weight-gradient side stream
 * of hip/functional.py).  The wide-tile kernels — one 8-wave workgroup per CU — are then split for half of the CUs, so that the
 * other stream's kernels find free CUs in every XCD instead of queueing behind a 700 us launch (+1.1 .. +2.1 % on the training
 * step).  Same result bits as without the flag only if the split count happens to coincide; the workspace size does not change. */
#define EVK_CONV_WGRAD_SHARED 32u

/* y = conv(x, w) (+ bias).  w: [Cout][kh][kw][Cin].  bias may be NULL.
 * Implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32), im2col rows gathered to LDS. */
int evk_conv2d_fwd(const evk_conv_desc* d, const float* x, const float* w, const float* bias,
                   float* y, uint32_t flags, void* stream);

/* dx = conv_transpose(dy, w)  (autograd of nn.Conv2d wrt input; aten::convolution_backward).
 * wt is the weight re-packed by evk_conv2d_pack_dgrad_weight: [Cin][kh][kw][Cout] (K = taps x Cout
 * contiguous per input channel).
 * dx is fully overwritten.  `accum` folds the sum with another gradient of the same tensor (the
 * residual branch of a ResNet block) into the epilogue instead of a separate add pass. */
int evk_conv2d_dgrad(const evk_conv_desc* d, const float* dy, const float* wt,
                     const float* accum /* NULL, or a tensor of dx's shape: dx = dgrad + accum */,
                     float* dx, void* stream);
int evk_conv2d_pack_dgrad_weight(const evk_conv_desc* d, const float* w, float* wt, void* stream);

/* The same two operations on the bf16 matrix pipe at fp32-grade accuracy ("x3": each fp32 operand is
 * split exactly into three bf16 terms; six of the nine partial products — everything above 2^-26 of
 * the product — are accumulated in fp32 by v_mfma_f32_32x32x16_bf16; 2.67x the MFMA rate of the exact
 * fp32 form).  Same reference call sites as evk_conv2d_fwd / _dgrad.  Requires Cin % 8 == 0 (fwd) /
 * Cout % 8 == 0 (dgrad).  The weight is pre-split once per optimiser step:
 *   for_dgrad = 0: planes [3][Cout][Kpad] bf16, K = (ky,kx,ci), Kpad = K rounded up to 32, zero padded;
 *   for_dgrad = 1: per residue class of the strided data gradient, planes [3][Cin][Kpad_c], K = (jy,jx,co),
 *                  produced straight from the OHWI parameter (no evk_conv2d_pack_dgrad_weight needed). */
size_t evk_conv2d_split_weight_bytes(const evk_conv_desc* d, int32_t for_dgrad);
int evk_conv2d_split_weight(const evk_conv_desc* d, const float* w, int32_t for_dgrad, void* wsplit,
                            void* stream);
/* All weights of a model in ONE launch (the planes change once per optimiser step; one launch per convolution
 * and direction was 164 launches per FarSeg-R50 step).  evk_conv2d_split_jobs fills (on the HOST) the 1..stride^2
 * jobs that evk_conv2d_split_weight(d, w, for_dgrad, wsplit) would launch — same layouts, chosen from the
 * descriptor — and returns their number (negative = error).  The caller sets job.arg[12] = workgroups it gives
 * the job (about evk_split_job_pairs(job) / 2048, at least 1), builds block_map[nblocks][2] = (job index, block
 * index inside the job), copies both tables to the device once, and calls evk_conv2d_split_multi after every
 * weight update.  Replaces nothing in the reference (operand preparation of the split arithmetic). */
typedef struct evk_split_job {
  const float* w;   /* OHWI parameter */
  void* out;        /* start of this job's planes inside the convolution's wsplit buffer */
  int32_t kind;     /* 0 forward, 1 data gradient (one residue class), 2 LDS-halo 3x3 (arg[3] != 0: under the f16x2
                     * arithmetic these planes are the Winograd F(2,3) kernel's, 12 transformed taps instead of 9) */
  int32_t arg[13];  /* layout parameters (opaque); arg[12] = workgroups assigned by the caller */
} evk_split_job;
int32_t evk_conv2d_split_job_count(const evk_conv_desc* d, int32_t for_dgrad);
int evk_conv2d_split_jobs(const evk_conv_desc* d, const float* w, int32_t for_dgrad, void* wsplit,
                          evk_split_job* jobs /* host */, int32_t max_jobs);
int64_t evk_split_job_pairs(const evk_split_job* job /* host */);
int evk_conv2d_split_multi(const evk_split_job* jobs_dev, const int32_t* block_map_dev, int32_t nblocks,
                           void* stream);
int evk_conv2d_fwd_x3(const evk_conv_desc* d, const float* x, const void* wsplit, const float* bias,
                      float* y, uint32_t flags, void* stream);
/* Forward convolution that also produces the BatchNorm statistics of its OUTPUT ("stats fusable into
 * conv epilogue"; reference call sites: every conv -> BatchNorm pair of _resnets.py:95-112, fs_relation.py:39-53,
 * fpn.py:163-167).  The epilogue parks each 32-row accumulator block in LDS, stores it as whole output rows and keeps
 * per-lane running (count, mean, M2) of its columns; one record per row-part goes to bn_parts[part][3][Cout].
 * *nparts (host) receives the number of parts written, 0 when this shape's kernel cannot do it (then the call was a
 * plain evk_conv2d_fwd_x3 and the caller runs evk_bn_fwd_train).  bn_capacity: parts the buffer holds
 * (evk_conv2d_stats_max_parts).  evk_bn_fwd_train_parts (below) consumes the records. */
int32_t evk_conv2d_stats_max_parts(const evk_conv_desc* d);
int evk_conv2d_fwd_x3_stats(const evk_conv_desc* d, const float* x, const void* wsplit, const float* bias,
                            float* y, uint32_t flags, float* bn_parts, int32_t bn_capacity, int32_t* nparts /* host */,
                            void* stream);
/* The reference's optional `--mixed_precision bf16` (core/launcher.py:40-80: the model runs under torch.autocast):
 * the same three convolution directions with PLAIN bf16 operands — activations and weights rounded to bf16 once, one
 * v_mfma_f32_32x32x16_bf16 product per operand pair, fp32 accumulate, fp32 tensors in HBM.  Same arguments and the same
 * weight-plane buffers as the x3 forms (only plane 0, the bf16 rounding of the weight, is read).  evk_conv2d_fwd_bf16
 * also takes the statistics arguments of evk_conv2d_fwd_x3_stats (bn_parts may be NULL).  Accuracy is that of bf16
 * inputs (relative 2^-9 per operand): never the default, never the headline benchmark. */
int evk_conv2d_fwd_bf16(const evk_conv_desc* d, const float* x, const void* wsplit, const float* bias, float* y,
                        uint32_t flags, float* bn_parts, int32_t bn_capacity, int32_t* nparts /* host */, void* stream);
int evk_conv2d_dgrad_bf16(const evk_conv_desc* d, const float* dy, const void* wsplit_t, const float* accum, float* dx,
                          void* stream);
int evk_conv2d_wgrad_bf16(const evk_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                          void* workspace, size_t workspace_bytes, void* stream);
/* "f16x2" arithmetic — the default of the training step since round 2: every fp32 operand is divided by a per-tensor
 * power of two s and split into TWO fp16 terms x / s = h + l (11-bit significands, round-to-nearest-even; the largest
 * element lands in [2^13, 2^14), |x/s - h - l| <= 2^-22 |x/s|), the product is rebuilt from the three partial products
 * l*wh + h*wl + h*wh on v_mfma_f32_32x32x16_f16 (dropped: l*wl <= 2^-22 |x*w|), accumulated in fp32 and multiplied by
 * the two scales at the end.  Same call sites as the x3 forms (nn.Conv2d at _resnets.py:21-29,149, fpn.py:23-37,
 * fs_relation.py:23-53), same plane buffers (planes 0 and 1 are used), half the matrix work.
 *   An ACTIVATION's scale source is a device buffer of evk_absmax_words() uint32: 64 slots one cache line apart, each
 *   the bit image of a partial max|x| (a non-negative float read as uint32); consumers take the maximum of the slots.
 *   evk_absmax:        fills such a buffer from a tensor (slot 0 = max|x|, the others zero); `workspace` =
 *                      evk_absmax_workspace_bytes() bytes, ZERO before the first call, private to one stream
 *   the BatchNorm calls fill one for their output as a by-product (y_absmax / dx_absmax arguments)
 *   evk_absmax_multi:  ONE word per tensor for n tensors in one launch (the weights, once per optimiser step)
 *   x_absmax / dy_absmax arguments below are activation buffers, w_absmax a single word; the kernels derive
 *   s = 2^(exponent - 13) from them. */
size_t evk_absmax_words(void);
size_t evk_absmax_workspace_bytes(void);
int evk_absmax(const float* x, int64_t n, uint32_t* out_bits, void* workspace, void* stream);
int evk_absmax_multi(const float* const* ptrs_dev, const int64_t* sizes_dev, int32_t n_tensors, uint32_t* out_bits,
                     void* stream);
int evk_conv2d_split_weight_f16x2(const evk_conv_desc* d, const float* w, int32_t for_dgrad, void* wsplit,
                                  const uint32_t* w_absmax, void* stream);
/* job.arg[11] = index of the job's weight in absmax_dev */
int evk_conv2d_split_multi_f16x2(const evk_split_job* jobs_dev, const int32_t* block_map_dev, int32_t nblocks,
                                 const uint32_t* absmax_dev, void* stream);
/* residual, bn_parts (+ nparts) may be NULL; with bn_parts the epilogue also emits the BatchNorm records (fwd_x3_stats).
 * y_absmax / dx_absmax (may be NULL): an activation scale buffer whose slots are ZERO on entry; the epilogue raises them
 * to max|output| (the output is a later convolution's operand: saves its evk_absmax pass).  accum may alias dx. */
int evk_conv2d_fwd_f16x2(const evk_conv_desc* d, const float* x, const uint32_t* x_absmax, const void* wsplit,
                         const uint32_t* w_absmax, const float* bias, const float* residual, float* y, uint32_t flags,
                         float* bn_parts, int32_t bn_capacity, int32_t* nparts /* host */, uint32_t* y_absmax,
                         void* stream);
int evk_conv2d_dgrad_f16x2(const evk_conv_desc* d, const float* dy, const uint32_t* dy_absmax, const void* wsplit_t,
                           const uint32_t* w_absmax, const float* accum, float* dx, uint32_t* dx_absmax, void* stream);
int evk_conv2d_wgrad_f16x2(const evk_conv_desc* d, const float* x, const uint32_t* x_absmax, const float* dy,
                           const uint32_t* dy_absmax, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                           void* stream);
/* Packed activation operands of the f16x2 arithmetic.  The kernels split every fp32 operand element into two fp16
 * (h, l) of x / s while staging it — the bound of the weight gradient.  A producer that knows the
 * scale can store the element already split: one 32-bit word, h in the low half, l in the high half, same shape and
 * strides as the fp32 tensor; the consumer's staging becomes a byte permute and its results are bit-identical to the
 * fp32 operand's under the same scale buffer.  evk_pack_f16x2 is the stand-alone producer (the BatchNorm passes are the
 * fused ones, EVK_BN_PACK_*), evk_unpack_f16x2 returns (h + l) * s (tests, debugging).  n % 4 == 0.
 * evk_conv2d_fwd_f16x2 takes EVK_CONV_X_PACKED in flags; the _ex forms of the two gradients take both flags
 * (dbias must be NULL with EVK_CONV_DY_PACKED). */
int evk_pack_f16x2(const float* x, int64_t n, const uint32_t* x_absmax, uint32_t* out, void* stream);
int evk_unpack_f16x2(const uint32_t* packed, int64_t n, const uint32_t* x_absmax, float* out, void* stream);
/* planar form (EVK_CONV_*_PLANAR): n % 8 == 0; `out` / `planar` hold n fp16 of h followed by n fp16 of l (4 n bytes).
 * Replaces nothing in the reference: operand preparation of this build's arithmetic, as evk_pack_f16x2. */
int evk_pack_planar_f16x2(const float* x, int64_t n, const uint32_t* x_absmax, void* out, void* stream);
int evk_unpack_planar_f16x2(const void* planar, int64_t n, const uint32_t* x_absmax, float* out, void* stream);
int evk_conv2d_dgrad_f16x2_ex(const evk_conv_desc* d, const void* dy, const uint32_t* dy_absmax, const void* wsplit_t,
                              const uint32_t* w_absmax, const float* accum, float* dx, uint32_t* dx_absmax,
                              uint32_t flags, void* stream);
/* dx = dgrad(dy) + (accum where its ReLU bit is set): the identity branch of a residual block hands its gradient over
 * UNMASKED together with the bits of the block's output (evk_bn_fwd_train_parts_bits / evk_bn_bwd_bits), and the mask is
 * applied here, in the epilogue that adds it.  Replaces the `out += identity; relu` backward of reference
 * ever/module/_resnets.py:95-112 as autograd would run it (a masked tensor written by the ReLU backward, read by the add). */
int evk_conv2d_dgrad_f16x2_masked(const evk_conv_desc* d, const void* dy, const uint32_t* dy_absmax, const void* wsplit_t,
                                  const uint32_t* w_absmax, const float* accum, const uint32_t* accum_bits, float* dx,
                                  uint32_t* dx_absmax, uint32_t flags, void* stream);
int evk_conv2d_wgrad_f16x2_ex(const evk_conv_desc* d, const void* x, const uint32_t* x_absmax, const void* dy,
                              const uint32_t* dy_absmax, float* dw, float* dbias, void* workspace,
                              size_t workspace_bytes, uint32_t flags, void* stream);
/* y = act(conv(x, w) + bias + residual): inference form of a residual block's last convolution with its
 * BatchNorm folded into (w, bias) — `out += identity; relu` of reference _resnets.py:95-112 in the epilogue. */
int evk_conv2d_fwd_res(const evk_conv_desc* d, const float* x, const float* w, const float* bias,
                       const float* residual, float* y, uint32_t flags, void* stream);
int evk_conv2d_fwd_x3_res(const evk_conv_desc* d, const float* x, const void* wsplit, const float* bias,
                          const float* residual, float* y, uint32_t flags, void* stream);
int evk_conv2d_dgrad_x3(const evk_conv_desc* d, const float* dy, const void* wsplit_t,
                        const float* accum, float* dx, void* stream);
/* weight / bias gradient in the same arithmetic: both operands (dy and im2col(x)) are split in registers
 * while they are transposed into the pixel-contiguous LDS image the bf16 MFMA needs. */
size_t evk_conv2d_wgrad_x3_workspace_bytes(const evk_conv_desc* d);
int evk_conv2d_wgrad_x3(const evk_conv_desc* d, const float* x, const float* dy, float* dw,
                        float* dbias, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ transposed convolution -- */
/* nn.ConvTranspose2d (any kernel / stride / padding / output_padding / dilation, groups = 1).
 * NO reference call site: the reference's hot path contains no transposed convolution (SURVEY.md §2.3 note;
 * `grep -rn ConvTranspose /root/reference/ever` is empty).  BASELINE.json's north_star names the operator
 * ("conv3x3/1x1 and transposed-conv lowered to MFMA"), so it is exposed; parity is pinned against
 * torch.nn.ConvTranspose2d (tests/test_conv_transpose_gpu.py), the only available oracle.
 * A transposed convolution with weight wT[Cin_t][kh][kw][Cout_t] (a ConvTranspose2d parameter with channels_last
 * strides) is the ADJOINT of the convolution C: u[N,H,W,Cin] -> z[N,Ho,Wo,Cout] that reads the same memory as
 * OHWI with Cout = Cin_t, Cin = Cout_t.  `d` is the descriptor of C: (N,H,W,Cin) = this operator's OUTPUT,
 * (Ho,Wo,Cout) = its INPUT.  forward = evk_conv2d_dgrad's residue-class kernel (no MFMA on structurally-zero taps)
 * + bias; input gradient = evk_conv2d_fwd; weight gradient = evk_conv2d_wgrad with the operand roles swapped;
 * dbias = column sums of dy.  Weight operands are prepared exactly as for the convolution C:
 * evk_conv2d_pack_dgrad_weight / evk_conv2d_split_weight(for_dgrad = 1) for the forward, the plain parameter /
 * evk_conv2d_split_weight(for_dgrad = 0) for the input gradient. */
int evk_conv_transpose2d_fwd(const evk_conv_desc* d, const float* x, const float* wt, const float* bias, float* y,
                             void* stream);
int evk_conv_transpose2d_fwd_x3(const evk_conv_desc* d, const float* x, const void* wsplit_t, const float* bias,
                                float* y, void* stream);
int evk_conv_transpose2d_dgrad(const evk_conv_desc* d, const float* dy, const float* w, float* dx, void* stream);
int evk_conv_transpose2d_dgrad_x3(const evk_conv_desc* d, const float* dy, const void* wsplit, float* dx, void* stream);
size_t evk_conv_transpose2d_wgrad_workspace_bytes(const evk_conv_desc* d, int32_t x3);
/* x: the operator's input [N,Ho,Wo,Cout], dy: gradient of its output [N,H,W,Cin]; dw in the parameter's memory order
 * [Cout][kh][kw][Cin] of C; dbias[Cin] (either may be NULL) */
int evk_conv_transpose2d_wgrad(const evk_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                               void* workspace, size_t workspace_bytes, void* stream);
int evk_conv_transpose2d_wgrad_x3(const evk_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* dw[Cout][kh][kw][Cin] = sum_pixels dy (x) im2col(x); dbias[Cout] = sum_pixels dy (dbias may be
 * NULL).  Split over pixel ranges; partials go to `workspace` and are reduced deterministically. */
size_t evk_conv2d_wgrad_workspace_bytes(const evk_conv_desc* d);
int evk_conv2d_wgrad(const evk_conv_desc* d, const float* x, const float* dy, float* dw,
                     float* dbias, void* workspace, size_t workspace_bytes, void* stream);

/* [N,H,W,C] -> [N,H,W,Cp] zero padded (forward) and the adjoint / slice (backward). Also used for
 * OHWI weights with rows = Cout*kh*kw.  C, Cp arbitrary. */
/* The ResNet stem (conv 7x7 stride 2 padding 3 on a 3- or 4-band image; _resnets.py:149, resnet.py:100-117) as a
 * space-to-depth 4x4 stride-1 convolution with 16 input channels, so that it runs on the split-MFMA kernels
 * (evk_conv2d_fwd_x3 / _wgrad_x3 with desc N, H/2+3, W/2+3, 16 -> H/2, W/2, Cout, k 4, stride 1, pad 0):
 *   evk_stem_s2d:            image [N,H,W,C] (or [N,C,H,W] when src_is_nchw) -> [N][H/2+3][W/2+3][16], zero borders
 *   evk_stem_s2d_weight:     w7 [Cout][7][7][C] -> w4 [Cout][4][4][16]
 *   evk_stem_s2d_weight_bwd: dw4 -> dw7 (the adjoint gather)
 * Same products, same sum; C <= 4, H and W even. */
int evk_stem_s2d(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t src_is_nchw,
                 void* stream);
int evk_stem_s2d_weight(const float* w7, float* w4, int32_t Cout, int32_t C, void* stream);
int evk_stem_s2d_weight_bwd(const float* dw4, float* dw7, int32_t Cout, int32_t C, void* stream);
int evk_pad_channels(const float* src, float* dst, int64_t rows, int32_t C, int32_t Cp, void* stream);
int evk_unpad_channels(const float* src, float* dst, int64_t rows, int32_t Cp, int32_t C, void* stream);
/* NCHW <-> NHWC(+pad) transposes at the model boundary (image in, logits out). */
int evk_nchw_to_nhwc(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W,
                     int32_t Cp, void* stream);
int evk_nhwc_to_nchw(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W,
                     int32_t Cp, void* stream);

/* ------------------------------------------------------------------ batch norm ------------- */
/* nn.BatchNorm2d (+ReLU, + residual add) — _resnets.py:95-112 (bn1/bn2/bn3, `out += identity`,
 * relu), fs_relation.py:39-53, fpn.py:163-167.  rows = N*H*W, C % 4 == 0, C <= 2048.
 * Training forward: batch statistics (biased var for normalisation, unbiased for running_var,
 * momentum update, as torch), y = act(gamma * (x-mean)*invstd + beta [+ residual]).
 * save_mean / save_invstd: [C] outputs kept for backward. */
#define EVK_BN_RELU 1u
/* evk_bn_bwd only: dx is written PACKED (evk_pack_f16x2's format) instead of fp32 — it is the producing convolution's dy
 * operand (EVK_CONV_DY_PACKED) and nothing else.  dx_absmax is required and its slots must be ZERO on entry: the
 * finalisation raises them to a per-channel bound of |dx| (from max|g|, max|xhat| gathered by the reduce pass) before the
 * apply pass writes dx under that scale; the bound is within ~2x of max|dx| (an upper bound is all a scale needs). */
#define EVK_BN_PACK_DX 2u
/* evk_bn_fwd_train_parts only, residual == NULL: y is written PACKED — it is the next convolution's operand
 * (EVK_CONV_X_PACKED in its forward and weight gradient) and nothing else.  y_absmax as for EVK_BN_PACK_DX: zero on entry,
 * slot 0 raised by the finalisation to a bound of |y| derived from the statistics records (2-3x the true maximum). */
#define EVK_BN_PACK_Y 4u
/* (ABI <= 19: EVK_BN_NO_FUSE = 8 asked evk_bn_bwd for its three-launch form instead of a one-launch form whose grid met at
 * device-wide barriers.  That form was removed in ABI 20 — measured level with the three launches on the default path — and
 * the bit is ignored.) */
#define EVK_BN_NO_FUSE 8u
size_t evk_bn_workspace_bytes(int64_t rows, int32_t C);
int evk_bn_fwd_train(const float* x, const float* residual, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps,
                     float* y, float* save_mean, float* save_invstd, int64_t rows, int32_t C,
                     uint32_t flags, void* workspace, size_t workspace_bytes, uint32_t* y_absmax, void* stream);
/* y_absmax / dx_absmax (may be NULL) in the four calls of this section: an activation scale buffer of
 * evk_absmax_words() uint32 (see evk_absmax) that the apply pass fills with the bit images of max|output| as a
 * by-product — the tensor is the next convolution's operand and the f16x2 arithmetic needs its scale; producing it here
 * saves that tensor one read pass. */
/* Eval forward (running statistics): y = act((x-rm)/sqrt(rv+eps)*gamma+beta [+ residual]). */
/* evk_bn_fwd_train with the statistics pass replaced by the merge (Chan, fp64, fixed order) of the (count, mean, M2)
 * records a convolution epilogue wrote (evk_conv2d_fwd_x3_stats): one pass over x less per layer. */
int evk_bn_fwd_train_parts(const float* x, const float* residual, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* y,
                           float* save_mean, float* save_invstd, int64_t rows, int32_t C, uint32_t flags,
                           const float* parts, int32_t nparts, void* workspace, size_t workspace_bytes,
                           uint32_t* y_absmax, void* stream);
/* ... and with the ReLU bits of the output written beside it (relu_bits: evk_relu_bits_bytes(rows * C) bytes, one bit per
 * element, "y > 0"; layout in csrc/common.hpp).  For the BatchNorm + add + ReLU that ends a residual block (reference
 * _resnets.py:95-112): its backward (evk_bn_bwd_bits) then reads the bits instead of the whole output tensor. */
int evk_bn_fwd_train_parts_bits(const float* x, const float* residual, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* y,
                                float* save_mean, float* save_invstd, int64_t rows, int32_t C, uint32_t flags,
                                const float* parts, int32_t nparts, void* workspace, size_t workspace_bytes,
                                uint32_t* y_absmax, uint32_t* relu_bits, void* stream);
size_t evk_relu_bits_bytes(int64_t n);
/* out = g where the bit is set, else 0 (n % 4 == 0): materialises a gradient that travels unmasked with its bits */
int evk_relu_bits_apply(const float* g, const uint32_t* bits, float* out, int64_t n, void* stream);
/* The stem: BatchNorm (batch statistics from the convolution epilogue's records, as evk_bn_fwd_train_parts) + ReLU +
 * MaxPool2d(3, 2, 1) in one pass each way — reference _resnets.py:150-153 (bn1, relu, maxpool).  x [N,H,W,C] ->
 * y [N,Ho,Wo,C] and code [N,Ho,Wo,C] uint8 (winning tap ky*3+kx of each window, first maximum in scan order, as
 * evk_maxpool3x3s2_fwd), Ho = (H-1)/2+1.  The backward rebuilds dz * (z > 0) from the pooled gradient dp and the
 * codes inside its two passes: the normalised full-resolution map and its gradient are never written. */
int evk_bn_relu_pool_fwd_train_parts(const float* x, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, float momentum, float eps, float* y, uint8_t* code,
                                     float* save_mean, float* save_invstd, int32_t N, int32_t H, int32_t W, int32_t C,
                                     const float* parts, int32_t nparts, void* workspace, size_t workspace_bytes,
                                     uint32_t* y_absmax, void* stream);
int evk_bn_relu_pool_bwd(const float* dp, const uint8_t* code, const float* x, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta,
                         int32_t N, int32_t H, int32_t W, int32_t C, int32_t train, void* workspace,
                         size_t workspace_bytes, uint32_t* dx_absmax, void* stream);
int evk_bn_fwd_eval(const float* x, const float* residual, const float* gamma, const float* beta,
                    const float* running_mean, const float* running_var, float eps, float* y,
                    float* save_mean /* may be NULL */, float* save_invstd /* may be NULL */,
                    int64_t rows, int32_t C, uint32_t flags, void* workspace, size_t workspace_bytes,
                    uint32_t* y_absmax, void* stream);
/* Backward of the training forward.  ReLU mask: taken from the forward output y when y != NULL
 * (required when the forward had a residual), else recomputed from x, gamma, beta and the saved
 * statistics (one HBM read less per pass).  d_residual (may be NULL) receives the masked upstream
 * gradient.  `train`=0 differentiates the eval forward (statistics are constants). */
int evk_bn_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
               const float* save_mean, const float* save_invstd, float* dx, float* d_residual,
               float* dgamma, float* dbeta, int64_t rows, int32_t C, uint32_t flags, int32_t train,
               void* workspace, size_t workspace_bytes, uint32_t* dx_absmax, void* stream);
/* evk_bn_bwd with the mask of the incoming gradient given as bits (relu_bits != NULL; y is then not read):
 *  - with EVK_BN_RELU: the forward's own ReLU bits (evk_bn_fwd_train_parts_bits) — a residual block's last BatchNorm reads
 *    dy, x and 1/32 of a tensor instead of dy, x and y in both passes; with d_residual == NULL the masked gradient of the
 *    identity branch is not written either: the caller hands (dy, bits) on, and the consumer masks while it adds
 *    (evk_conv2d_dgrad_f16x2_masked), or another BatchNorm takes them as below — 5 tensor transfers instead of 7;
 *  - without EVK_BN_RELU: dy is such an unmasked gradient and relu_bits its mask (the down-sampling branch's BatchNorm
 *    behind a residual block's add). */
int evk_bn_bwd_bits(const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_invstd, float* dx, float* d_residual,
                    float* dgamma, float* dbeta, int64_t rows, int32_t C, uint32_t flags, int32_t train,
                    void* workspace, size_t workspace_bytes, uint32_t* dx_absmax, const uint32_t* relu_bits, void* stream);

/* `to` waits for everything enqueued on `from` so far (an event of a per-device ring recorded on `from`, waited for on
 * `to`): the fork of the weight-gradient side stream from the backward's stream, once per convolution layer
 * (ever_amd/hip/functional.py).  Not thread-safe per device (the backward of one device runs on one thread).
 * No reference counterpart (the reference runs its backward on one stream). */
int evk_stream_fork(void* from, void* to);

/* 1 when kernels of streams `a` and `b` run at the same time, 0 when the runtime serialises them (HIP multiplexes streams
 * onto a few hardware queues; two streams on one queue never overlap), -1 on error.  Measured: a single-workgroup kernel
 * spinning `us` microseconds on each stream, forked and joined by events; *elapsed_us (optional) receives what the pair
 * took.  Synchronises stream `a`; not to be called under a stream capture.  Used once per device to choose the
 * weight-gradient side stream (ever_amd/hip/functional.py).  No reference counterpart. */
int evk_streams_overlap(void* a, void* b, int32_t us, float* elapsed_us);

/* ------------------------------------------------------------------ pointwise / resampling - */
/* nn.ReLU (fs_relation.py:25) and its backward; elementwise add (fpn.py:105). */
int evk_relu_fwd(const float* x, float* y, int64_t n, void* stream);
int evk_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
int evk_add(const float* a, const float* b, float* out, int64_t n, void* stream);
int evk_scale(const float* a, float alpha, float* out, int64_t n, void* stream);
/* Grouped convolution (reference _resnets.py:21-24 `groups=groups`; the ResNeXt bodies :291-324) runs as a DENSE one: the
 * weight w [Cout][taps][Cin / groups] (OHWI memory) is expanded to the block-diagonal dense [Cout][taps][Cin] with exact zeros
 * outside an output channel's group, the dense kernels above take it, and the adjoint gathers the diagonal blocks of the
 * dense weight gradient.  Exact in every arithmetic; at 32 groups of 4..8 channels the dense form is what fills an MFMA
 * tile anyway. */
int evk_group_weight_expand(const float* w, float* dense, int32_t Cout, int32_t taps, int32_t Cin, int32_t groups, void* stream);
int evk_group_weight_gather(const float* ddense, float* dw, int32_t Cout, int32_t taps, int32_t Cin, int32_t groups, void* stream);
/* out = a * b * alpha — nn.Dropout(p) with a 0/1 keep mask (fpn.py:183,190 `self.dropout(out_feat)`), alpha = 1/(1-p) */
int evk_mul_scale(const float* a, const float* b, float alpha, float* out, int64_t n, void* stream);
/* nn.GELU() (exact, erf form) and its derivative: the decoder's activation when norm_fn is not BatchNorm2d — fpn.py:167 */
int evk_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int evk_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);

/* nn.MaxPool2d(3, 2, 1) — _resnets.py:153.  code: one byte per output element = winning tap
 * ky*3+kx (first maximum in scan order).  x: [N,H,W,C] -> y: [N,Ho,Wo,C], Ho = (H-1)/2+1. */
int evk_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* code, int32_t N, int32_t H, int32_t W,
                         int32_t C, void* stream);
int evk_maxpool3x3s2_bwd(const float* dy, const uint8_t* code, float* dx, int32_t N, int32_t H,
                         int32_t W, int32_t C, void* stream);

/* F.interpolate(scale_factor=2, mode="nearest") + lateral add — fpn.py:100-105.
 * out[n,y,x,:] = lateral[n,y,x,:] + top[n,y/2,x/2,:];  top is [N,H/2,W/2,C]. */
/* out_absmax (may be NULL) here and in evk_mean4_fwd: an activation scale buffer, zero on entry, raised to max|out| */
int evk_upsample_nearest2x_add_fwd(const float* top, const float* lateral, float* out, int32_t N,
                                   int32_t H, int32_t W, int32_t C, uint32_t* out_absmax, void* stream);
/* adjoint wrt `top`: dtop[n,y,x,:] = sum of the 2x2 block of dout. (d lateral = dout.) */
int evk_upsample_nearest2x_bwd(const float* dout, float* dtop, int32_t N, int32_t H, int32_t W,
                               int32_t C, void* stream);

/* F.max_pool2d(x, 1, 2, 0) of LastLevelMaxPool — fpn.py:118-120 (the FPN's optional `top_blocks`): a one-pixel window at
 * stride 2, y[n,yo,xo,:] = x[n,2yo,2xo,:] with Ho = (H-1)/2+1; _bwd is its adjoint (dx = dy at the even pixels, 0 elsewhere;
 * H, W are x's dims in both). */
int evk_subsample2_fwd(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int evk_subsample2_bwd(const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);

/* nn.UpsamplingBilinear2d(scale_factor=s) == bilinear, align_corners=True — fpn.py:168,180.
 * x: [N,Hi,Wi,C] -> y: [N,Ho,Wo,C]; src = dst*(in-1)/(out-1). Backward is the gather-form adjoint. */
int evk_upsample_bilinear_fwd(const float* x, float* y, int32_t N, int32_t Hi, int32_t Wi,
                              int32_t Ho, int32_t Wo, int32_t C, void* stream);
int evk_upsample_bilinear_bwd(const float* dy, float* dx, int32_t N, int32_t Hi, int32_t Wi,
                              int32_t Ho, int32_t Wo, int32_t C, void* stream);

/* F.adaptive_avg_pool2d(x, 1) — fs_relation.py:177. x: [N,HW,C] -> y: [N,C]. */
int evk_gap_fwd(const float* x, float* y, int32_t N, int32_t HW, int32_t C, void* stream);
int evk_gap_bwd(const float* dy, float* dx, int32_t N, int32_t HW, int32_t C, void* stream);

/* FS-Relation — fs_relation.py:61-71: r = sigmoid(sum_c scene[n,c]*content[n,p,c]); out = r*feat.
 * scene: [N,C], content/feat/out: [N,HW,C], r: [N,HW] (saved for backward). */
int evk_relation_fwd(const float* scene, const float* content, const float* feat, float* out,
                     float* r, int32_t N, int32_t HW, int32_t C, void* stream);
/* dscene must be zero-initialised by the caller? No: fully written (two-stage reduction inside,
 * workspace from evk_relation_workspace_bytes). */
size_t evk_relation_workspace_bytes(int32_t N, int32_t HW, int32_t C);
int evk_relation_bwd(const float* dout, const float* scene, const float* content, const float* feat,
                     const float* r, float* dscene, float* dcontent, float* dfeat, int32_t N,
                     int32_t HW, int32_t C, void* workspace, size_t workspace_bytes, void* stream);

/* `sum(list)/len(list)` over 4 decoder branches — fpn.py:189. */
int evk_mean4_fwd(const float* a, const float* b, const float* c, const float* d, float* out,
                  int64_t n, uint32_t* out_absmax, void* stream);

/* ------------------------------------------------------------------ pixel losses ----------- */
/* Labels are int64 [N*H*W]; ignore_index pixels are dropped from every sum (the reference
 * compacts with masked_select, loss.py:10-17,26-37).  logits: [N*H*W, C] (NHWC).
 * Each forward writes `stats` (doubles) that the matching backward consumes; loss is a device
 * float scalar.  grad_scale is the upstream dL/dloss (device float pointer, may be NULL = 1).
 * `stats` must hold evk_loss_stats_doubles(K) doubles (K finals followed by per-workgroup
 * partials); K = 2 (BCE), 2*C (dice), 3 (CE).  Only the first K are meaningful to the caller. */
int64_t evk_loss_stats_doubles(int32_t K);

/* binary_cross_entropy_with_logits(ignore) — loss.py:229-235; with label_smoothing > 0 it is
 * label_smoothing_binary_cross_entropy — loss.py:222-226 (target 0 -> eps, 1 -> 1-eps).
 * C == 1. stats: double[2]={sum,count} */
int evk_bce_fwd(const float* logits, const int64_t* labels, int64_t npix, int64_t ignore_index,
                float label_smoothing, float* loss, double* stats, void* stream);
int evk_bce_bwd(const float* logits, const int64_t* labels, int64_t npix, int64_t ignore_index,
                float label_smoothing, const double* stats, const float* grad_scale, float* dlogits,
                int32_t accumulate, void* stream);

/* the same with F.binary_cross_entropy_with_logits' `pos_weight` (one float: the path's heads have one logit channel)
 * and `reduction` (0 = 'mean', 1 = 'sum') — loss.py:229-235 passes both through */
int evk_bce_fwd_ex(const float* logits, const int64_t* labels, int64_t npix, int64_t ignore_index,
                   float label_smoothing, float pos_weight, int32_t reduction, float* loss, double* stats, void* stream);
int evk_bce_bwd_ex(const float* logits, const int64_t* labels, int64_t npix, int64_t ignore_index,
                   float label_smoothing, float pos_weight, int32_t reduction, const double* stats,
                   const float* grad_scale, float* dlogits, int32_t accumulate, void* stream);

/* dice_loss_with_logits — loss.py:40-75.  C==1: p=sigmoid; C>1: p=softmax, one-hot target.
 * stats: double[2*C] = {inter[c], z[c]} (z = sum p + sum y, before smoothing).  In distributed
 * training the caller all-reduces `stats` between evk_dice_stats and evk_dice_finish
 * (loss.py:20-23,46-48). */
int evk_dice_stats(const float* logits, const int64_t* labels, int64_t npix, int32_t C,
                   int64_t ignore_index, double* stats, void* stream);
int evk_dice_finish(const double* stats, int32_t C, float smooth, int32_t ignore_channel,
                    float* loss, void* stream);
int evk_dice_bwd(const float* logits, const int64_t* labels, int64_t npix, int32_t C,
                 int64_t ignore_index, const double* stats, float smooth, int32_t ignore_channel,
                 const float* grad_scale, float* dlogits, int32_t accumulate, void* stream);

/* F.cross_entropy(ignore_index) (user model code; label-smoothing variant loss.py:207-219).
 * stats: double[3] = {sum nll, count, sum(-sum_c logp)} ; loss = (1-eps)*nll/count + eps/C*smooth/count */
int evk_ce_fwd(const float* logits, const int64_t* labels, int64_t npix, int32_t C,
               int64_t ignore_index, float label_smoothing, float* loss, double* stats, void* stream);
int evk_ce_bwd(const float* logits, const int64_t* labels, int64_t npix, int32_t C,
               int64_t ignore_index, float label_smoothing, const double* stats,
               const float* grad_scale, float* dlogits, int32_t accumulate, void* stream);

/* soft_cross_entropy(input, target) — loss.py:238-242: -(target*log_softmax(input,1)).mean((0,2,3)).sum().
 * target: float [N*H*W, C] (NHWC, same layout as logits).  stats: K = 1. */
int evk_soft_ce_fwd(const float* logits, const float* target, int64_t npix, int32_t C, float* loss,
                    double* stats, void* stream);
int evk_soft_ce_bwd(const float* logits, const float* target, int64_t npix, int32_t C,
                    const float* grad_scale, float* dlogits, void* stream);

/* ------------------------------------------------------------------ optimizer -------------- */
/* torch.optim.SGD step over a flat list of tensors (opt/optimizer.py:7) and
 * clip_grad_norm_ (interface/module.py:96-108).  ptr tables live in device memory. */
/* partial: double[evk_opt_blocks_per_tensor() * ntensors] scratch.  Writes the global L2 norm and
 * clip_coef = min(1, max_norm / (norm + 1e-6)) to device scalars (no host sync). */
int32_t evk_opt_blocks_per_tensor(void);
int evk_sqnorm_multi(const float* const* grads, const int64_t* sizes, int32_t ntensors,
                     double* partial, float max_norm, float* total_norm, float* clip_coef,
                     void* stream);
int evk_sgd_multi(float* const* params, const float* const* grads, float* const* momentum_bufs,
                  const int64_t* sizes, int32_t ntensors, float lr, float momentum, float dampening,
                  float weight_decay, int32_t nesterov, int32_t first_step,
                  const float* clip_coef /* device scalar or NULL */, void* stream);
/* ... with the learning rate read from a device word when lr_dev != NULL (lr is then ignored): a launch captured into a
 * hipGraph (ever_amd/core/graph.py) must follow the schedule on replay. */
int evk_sgd_multi_lr(float* const* params, const float* const* grads, float* const* momentum_bufs,
                     const int64_t* sizes, int32_t ntensors, float lr, const float* lr_dev, float momentum,
                     float dampening, float weight_decay, int32_t nesterov, int32_t first_step,
                     const float* clip_coef, void* stream);
/* torch.optim.Adam (decoupled = 0) / AdamW (decoupled = 1) step, no amsgrad — opt/optimizer.py:8-9 registers both.
 * bias_correction1 = 1 - beta1^step, sqrt_bias_correction2 = sqrt(1 - beta2^step) for the step being taken. */
int evk_adam_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   const int64_t* sizes, int32_t ntensors, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int32_t decoupled, float bias_correction1, float sqrt_bias_correction2,
                   const float* clip_coef /* device scalar or NULL */, void* stream);

/* ------------------------------------------------------------------ "next" rows ---- */
/* Class-probability statistics over the valid pixels: stats[0..C) = tp_c = sum p_c*y_c, [C..2C) = sum p_c,
 * [2C..3C) = sum y_c  (p = sigmoid for C == 1, softmax otherwise; y one-hot / the 0-1 label).  They are the
 * sufficient statistics of tversky_loss_with_logits (reference ever/module/loss.py:78-143; the caller
 * all-reduces them under sync_statistics and forms the ratio), and evk_prob_stats_bwd is their adjoint:
 * dlogits for any loss L(tp, sp) given g_tp[c] = dL/dtp_c, g_sp[c] = dL/dsp_c (device arrays).
 * `stats` needs evk_prob_stats_doubles(C) doubles. */
int64_t evk_prob_stats_doubles(int32_t C);
int evk_prob_stats(const float* logits, const int64_t* labels, int64_t npix, int32_t C,
                   int64_t ignore_index, double* stats, void* stream);
int evk_prob_stats_bwd(const float* logits, const int64_t* labels, int64_t npix, int32_t C,
                       int64_t ignore_index, const float* g_tp, const float* g_sp, float* dlogits,
                       int32_t accumulate, void* stream);
/* Focal losses on float targets of the logits' shape (any layout; n elements).
 * mode 0: focal_loss(normalize=False), loss.py:158-176 (modulating factor detached);
 * mode 1: sigmoid_focal_loss, loss.py:179-201 (alpha < 0 disables the alpha weighting);
 * mode 2: focal_loss(normalize=True) — the normalisation cancels identically: sum of BCE terms.
 * mean != 0 divides by n.  stats: 1 + 256 doubles. */
int evk_focal_fwd(const float* logits, const float* target, int64_t n, float gamma, float alpha,
                  int32_t mode, int32_t mean, float* loss, double* stats, void* stream);
int evk_focal_bwd(const float* logits, const float* target, int64_t n, float gamma, float alpha,
                  int32_t mode, int32_t mean, const float* grad_scale, float* dlogits, void* stream);
/* cm[t*C + p] += #{i : y_true[i] = t, y_pred[i] = p}; pairs with either index outside [0, C) are skipped
 * (ignore labels).  Replaces the host/scipy.sparse accumulation of ever/metric/confusion_matrix.py:11-24.
 * The _from_logits form fuses the prediction: threshold 0 for one logit channel (classes {0,1}), first
 * argmax otherwise; logits are [npix][C_logits] (NHWC). */
int evk_confusion_matrix(const int64_t* y_true, const int64_t* y_pred, int64_t n, int32_t num_classes,
                         int64_t* cm, void* stream);
int evk_confusion_from_logits(const float* logits, const int64_t* y_true, int64_t npix,
                              int32_t C_logits, int32_t num_classes, int64_t* cm, void* stream);

/* nn.GroupNorm(G, C) (+ReLU) on [N, HW, C] — reference fs_relation.py:88-116 (FSRelationV2 scene encoders).
 * flags bit 0: ReLU fused (backward then needs y).  save_mean / save_rstd: [N*G].  C % 4 == 0, C % G == 0. */
size_t evk_gn_workspace_bytes(int32_t N, int64_t HW, int32_t C, int32_t G);
int evk_gn_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y,
               float* save_mean, float* save_rstd, int32_t N, int64_t HW, int32_t C, int32_t G,
               uint32_t flags, void* workspace, size_t workspace_bytes, void* stream);
int evk_gn_bwd(const float* dy, const float* x, const float* y, const float* gamma,
               const float* save_mean, const float* save_rstd, float* dx, float* dgamma, float* dbeta,
               int32_t N, int64_t HW, int32_t C, int32_t G, uint32_t flags, void* workspace,
               size_t workspace_bytes, void* stream);
/* torch.cat([a, b], dim=1) on NHWC rows and its adjoint (either output of the split may be NULL) —
 * fs_relation.py:155.  Ca, Cb multiples of 4. */
int evk_concat_channels(const float* a, const float* b, float* out, int64_t rows, int32_t Ca,
                        int32_t Cb, void* stream);
int evk_split_channels(const float* src, float* a, float* b, int64_t rows, int32_t Ca, int32_t Cb,
                       void* stream);
/* y[n][hw][c] = x[n][hw][c] * scale[n][c]: nn.Dropout2d with the caller's keep-mask/(1-p) — fs_relation.py:104,
 * 121; the same call is its backward. */
int evk_channel_scale(const float* x, const float* scale, float* y, int32_t N, int64_t HW, int32_t C,
                      void* stream);

/* FS-Relation with its two BatchNorm + ReLU passes inside (reference fs_relation.py:39-53: content / re-encoding =
 * ReLU(BN(conv1x1(p))); :61-71: out = sigmoid(<scene, content>) * re-encoded).  The stand-alone passes wrote both normalised
 * maps only for the relation kernel to read them back, and their backward read (g, z) of both once more just for the
 * per-channel sums.  Here:
 *  - evk_bn_finalize_parts merges the producing convolution's statistics records (evk_conv2d_fwd_f16x2: bn_parts) into
 *    save_mean / save_invstd / scale_shift [2][C] and updates the running statistics — no apply pass;
 *  - evk_relation_bn_fwd reads the convolution outputs zc, zf and applies scale / shift / ReLU on the fly; it raises the
 *    64 slots of out_absmax (zero on entry; may be NULL) with max|out|;
 *  - evk_relation_bn_bwd rebuilds both activations from z, writes the MASKED gradients gc, gf (w.r.t. the BatchNorm
 *    outputs, zero where the ReLU was off) and leaves, per workgroup, (sum g, sum g * xhat) and (max|g|, max|xhat|) of both
 *    BatchNorms in the workspace: floats [nb C] scene partials | content sums [nb][2][C] | content maxima [nb][2][C] |
 *    re-encoding sums | re-encoding maxima, nb = evk_relation_bn_parts(N, HW); mean_invstd = save_mean, save_invstd
 *    contiguous [2][C];
 *  - evk_bn_bwd_from_partials turns such records into dgamma, dbeta and dx = BatchNorm backward of the masked g (with
 *    EVK_BN_PACK_DX: packed, dx_absmax zero on entry); workspace 16 C floats. */
int evk_bn_finalize_parts(const float* parts, int32_t nparts, int32_t C, int64_t rows, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                          float* save_invstd, float* scale_shift, void* stream);
int evk_relation_bn_fwd(const float* scene, const float* zc, const float* scale_shift_c, const float* zf,
                        const float* scale_shift_f, float* out, float* r, int32_t N, int32_t HW, int32_t C,
                        uint32_t* out_absmax, void* stream);
int32_t evk_relation_bn_parts(int32_t N, int32_t HW);
size_t evk_relation_bn_workspace_bytes(int32_t N, int32_t HW, int32_t C);
int evk_relation_bn_bwd(const float* dout, const float* scene, const float* zc, const float* scale_shift_c,
                        const float* mean_invstd_c, const float* zf, const float* scale_shift_f, const float* mean_invstd_f,
                        const float* r, float* dscene, float* gc, float* gf, int32_t N, int32_t HW, int32_t C, void* workspace,
                        size_t workspace_bytes, void* stream);
int evk_bn_bwd_from_partials(const float* g, const float* x, const float* gamma, const float* save_mean,
                             const float* save_invstd, const float* partial, const float* maxima, int32_t nparts, float* dx,
                             float* dgamma, float* dbeta, int64_t rows, int32_t C, uint32_t flags, int32_t train,
                             void* workspace, size_t workspace_bytes, uint32_t* dx_absmax, void* stream);

/* BatchNorm + ReLU + a narrow 1x1 convolution as ONE consumer of a convolution output z (the decoder's classifier applied
 * per branch: reference fpn.py:163-170 `conv3x3 -> BN -> ReLU`, :179-193 the 1x1 classifier; ever_amd/module/fpn.py runs the
 * classifier before the last upsampling): out[pix][k] = sum_c relu(bn(z))[pix][c] * w[k][c] + bias[k], K <= 16.  The
 * normalised map is never written.  scale_shift / save_mean / save_invstd from evk_bn_finalize_parts.  Backward: dl
 * [rows][K] -> dz (gradient of z; EVK_BN_PACK_DX: packed, dx_absmax zero on entry), dgamma, dbeta, dw [K][C], dbias [K];
 * it reads z twice and never forms the C-channel gradient of the normalised map. */
int evk_bn_relu_dot_fwd(const float* z, const float* scale_shift, const float* w, const float* bias, float* out, int64_t rows,
                        int32_t C, int32_t K, void* stream);
size_t evk_bn_relu_dot_workspace_bytes(int64_t rows, int32_t C, int32_t K);
int evk_bn_relu_dot_bwd(const float* dl, const float* z, const float* scale_shift, const float* gamma, const float* save_mean,
                        const float* save_invstd, const float* w, float* dz, float* dgamma, float* dbeta, float* dw,
                        float* dbias, int64_t rows, int32_t C, int32_t K, uint32_t flags, void* workspace,
                        size_t workspace_bytes, uint32_t* dx_absmax, void* stream);

/* Synchronized BatchNorm in stages (torch.nn.SyncBatchNorm under the trainer's `sync_bn`, reference
 * ever/trainer/th_ddp_trainer.py + SURVEY §8 C5): the exchange between the stages is the caller's
 * (torch.distributed over RCCL): forward all-gather of `stats` (local mean, local sum of squared deviations,
 * fp64 [2C]) + the row count; backward all-reduce of `sums` (sum g, sum g*xhat, fp64 [2C]).
 * Workspace: evk_bn_workspace_bytes(rows, C).  flags / y / d_residual as evk_bn_fwd_train / evk_bn_bwd. */
int evk_bn_local_stats(const float* x, double* stats, int64_t rows, int32_t C, void* workspace,
                       size_t workspace_bytes, void* stream);
int evk_bn_apply_stats(const float* x, const float* residual, const float* gamma, const float* beta,
                       const float* mean, const float* invstd, float* y, int64_t rows, int32_t C,
                       uint32_t flags, void* workspace, size_t workspace_bytes, void* stream);
int evk_bn_bwd_local_sums(const float* dy, const float* x, const float* y, const float* gamma,
                          const float* beta, const float* mean, const float* invstd, float* d_residual,
                          double* sums, int64_t rows, int32_t C, uint32_t flags, void* workspace,
                          size_t workspace_bytes, void* stream);
int evk_bn_bwd_apply_sums(const float* dy, const float* x, const float* y, const float* gamma,
                          const float* beta, const float* mean, const float* invstd, const float* mean_g,
                          const float* mean_gx, float* dx, int64_t rows, int32_t C, uint32_t flags,
                          void* workspace, size_t workspace_bytes, void* stream);

/* dst[offsets[t] + i] = srcs[t][i] * scale for t < ntensors (a NULL source packs zeros): one launch gathers the
 * gradients of a bucket into the flat RCCL all-reduce buffer, pre-divided by the world size — the gradient
 * exchange of the DDP trainer (reference ever/trainer/th_ddp_trainer.py: DistributedDataParallel's reducer).
 * srcs / sizes / offsets are device arrays.  A source may be EXACTLY its own destination slot (srcs[t] == dst +
 * offsets[t]: gradient accumulation over several backward passes without no_sync, reference core/launcher.py:196,317-321 —
 * the second pass accumulates into the bucket view and the pack scales it in place); partially overlapping ranges are not
 * allowed.  tests/world2_gpu_worker.py:case_forward_times_2 holds the in-place case at world size 2. */
int evk_pack_multi(const float* const* srcs, const int64_t* sizes, const int64_t* offsets,
                   int32_t ntensors, float scale, float* dst, void* stream);

/* F.cross_entropy(reduction='none', ignore_index) per pixel (0 on ignored pixels) and its backward with a per-pixel
 * upstream gradient — the input of online_hard_example_mining. */
int evk_ce_pixel_fwd(const float* logits, const int64_t* labels, int64_t npix, int32_t C,
                     int64_t ignore_index, float* loss_pix, void* stream);
int evk_ce_pixel_bwd(const float* logits, const int64_t* labels, int64_t npix, int32_t C,
                     int64_t ignore_index, const float* grad_pix, float* dlogits, void* stream);
/* online_hard_example_mining (reference loss.py:146-155): loss = mean of the non-zero values among the `keep`
 * largest of losses[0..n); exact k-th value by a 3-pass radix select.  `state`: evk_ohem_state_bytes() bytes, kept
 * for the backward (dlosses = grad / #kept on the kept non-zero elements, 0 elsewhere). */
int64_t evk_ohem_state_bytes(void);
int evk_ohem_fwd(const float* losses, int64_t n, int64_t keep, float* loss, void* state, void* stream);
int evk_ohem_bwd(const float* losses, int64_t n, void* state, const float* grad_scale, float* dlosses,
                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVER_HIP_H_ */
