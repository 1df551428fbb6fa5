"""ctypes binding of libever_hip.so (the C-ABI declared in include/ever_hip.h).

The product path has no CPU fallback: if the shared library is missing or a kernel call fails,
the error is raised here.  Tensors stay torch-owned; only raw device pointers, sizes and the
current HIP stream cross the boundary.
"""
import ctypes as C
import os

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libever_hip.so')
# dev A/B runs on one box: EVK_LIB=<path of another build of the same ABI>
_LIB_PATH = os.environ.get('EVK_LIB', _LIB_PATH)

c_void_p, c_int, c_i32, c_i64, c_u32, c_f32, c_size_t = (
    C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_uint32, C.c_float, C.c_size_t)


class ConvDesc(C.Structure):
    """Mirror of `evk_conv_desc`."""
    _fields_ = [(n, c_i32) for n in (
        'N', 'H', 'W', 'Cin', 'Ho', 'Wo', 'Cout', 'kh', 'kw',
        'stride_h', 'stride_w', 'pad_h', 'pad_w', 'dil_h', 'dil_w')]


class SplitJob(C.Structure):
    """Mirror of `evk_split_job` (one weight-plane job of evk_conv2d_split_multi)."""
    _fields_ = [('w', c_void_p), ('out', c_void_p), ('kind', c_i32), ('arg', c_i32 * 13)]


P = c_void_p  # every device pointer and the stream are passed as void*
_DP = C.POINTER(ConvDesc)
_JP = C.POINTER(SplitJob)

# name -> (restype, argtypes).  Order and types follow include/ever_hip.h exactly;
# tests/test_host_api_cpu.py::test_library_exports_every_declared_symbol checks that every symbol the header declares
# is exported and listed here.
SIGNATURES = {
    'evk_last_error': (C.c_char_p, []),
    'evk_abi_version': (c_int, []),
    'evk_build_arch': (C.c_char_p, []),
    'evk_conv2d_fwd': (c_int, [_DP, P, P, P, P, c_u32, P]),
    'evk_conv2d_dgrad': (c_int, [_DP, P, P, P, P, P]),
    'evk_conv2d_pack_dgrad_weight': (c_int, [_DP, P, P, P]),
    'evk_conv2d_split_weight_bytes': (c_size_t, [_DP, c_i32]),
    'evk_conv2d_split_weight': (c_int, [_DP, P, c_i32, P, P]),
    'evk_conv2d_split_job_count': (c_i32, [_DP, c_i32]),
    'evk_conv2d_split_jobs': (c_int, [_DP, P, c_i32, P, _JP, c_i32]),
    'evk_split_job_pairs': (c_i64, [_JP]),
    'evk_conv2d_split_multi': (c_int, [P, P, c_i32, P]),
    'evk_conv2d_fwd_x3': (c_int, [_DP, P, P, P, P, c_u32, P]),
    'evk_conv2d_stats_max_parts': (c_i32, [_DP]),
    'evk_conv2d_fwd_x3_stats': (c_int, [_DP, P, P, P, P, c_u32, P, c_i32, C.POINTER(c_i32), P]),
    'evk_conv2d_fwd_bf16': (c_int, [_DP, P, P, P, P, c_u32, P, c_i32, C.POINTER(c_i32), P]),
    'evk_conv2d_dgrad_bf16': (c_int, [_DP, P, P, P, P, P]),
    'evk_conv2d_wgrad_bf16': (c_int, [_DP, P, P, P, P, P, c_size_t, P]),
    'evk_conv2d_dgrad_x3': (c_int, [_DP, P, P, P, P, P]),
    'evk_absmax_words': (c_size_t, []),
    'evk_absmax_workspace_bytes': (c_size_t, []),
    'evk_absmax': (c_int, [P, c_i64, P, P, P]),
    'evk_absmax_multi': (c_int, [P, P, c_i32, P, P]),
    'evk_conv2d_split_weight_f16x2': (c_int, [_DP, P, c_i32, P, P, P]),
    'evk_conv2d_split_multi_f16x2': (c_int, [P, P, c_i32, P, P]),
    'evk_conv2d_fwd_f16x2': (c_int, [_DP, P, P, P, P, P, P, P, c_u32, P, c_i32, C.POINTER(c_i32), P, P]),
    'evk_conv2d_dgrad_f16x2': (c_int, [_DP, P, P, P, P, P, P, P, P]),
    'evk_conv2d_wgrad_f16x2': (c_int, [_DP, P, P, P, P, P, P, P, c_size_t, P]),
    'evk_pack_f16x2': (c_int, [P, c_i64, P, P, P]),
    'evk_unpack_f16x2': (c_int, [P, c_i64, P, P, P]),
    'evk_pack_planar_f16x2': (c_int, [P, c_i64, P, P, P]),
    'evk_unpack_planar_f16x2': (c_int, [P, c_i64, P, P, P]),
    'evk_conv2d_dgrad_f16x2_ex': (c_int, [_DP, P, P, P, P, P, P, P, c_u32, P]),
    'evk_conv2d_dgrad_f16x2_masked': (c_int, [_DP, P, P, P, P, P, P, P, P, c_u32, P]),
    'evk_conv2d_wgrad_f16x2_ex': (c_int, [_DP, P, P, P, P, P, P, P, c_size_t, c_u32, P]),
    'evk_conv_transpose2d_fwd': (c_int, [_DP, P, P, P, P, P]),
    'evk_conv_transpose2d_fwd_x3': (c_int, [_DP, P, P, P, P, P]),
    'evk_conv_transpose2d_dgrad': (c_int, [_DP, P, P, P, P]),
    'evk_conv_transpose2d_dgrad_x3': (c_int, [_DP, P, P, P, P]),
    'evk_conv_transpose2d_wgrad_workspace_bytes': (c_size_t, [_DP, c_i32]),
    'evk_conv_transpose2d_wgrad': (c_int, [_DP, P, P, P, P, P, c_size_t, P]),
    'evk_conv_transpose2d_wgrad_x3': (c_int, [_DP, P, P, P, P, P, c_size_t, P]),
    'evk_conv2d_fwd_res': (c_int, [_DP, P, P, P, P, P, c_u32, P]),
    'evk_conv2d_fwd_x3_res': (c_int, [_DP, P, P, P, P, P, c_u32, P]),
    'evk_conv2d_wgrad_x3_workspace_bytes': (c_size_t, [_DP]),
    'evk_conv2d_wgrad_x3': (c_int, [_DP, P, P, P, P, P, c_size_t, P]),
    'evk_conv2d_wgrad_workspace_bytes': (c_size_t, [_DP]),
    'evk_conv2d_wgrad': (c_int, [_DP, P, P, P, P, P, c_size_t, P]),
    'evk_stem_s2d': (c_int, [P, P, c_i32, c_i32, c_i32, c_i32, c_i32, P]),
    'evk_stem_s2d_weight': (c_int, [P, P, c_i32, c_i32, P]),
    'evk_stem_s2d_weight_bwd': (c_int, [P, P, c_i32, c_i32, P]),
    'evk_pad_channels': (c_int, [P, P, c_i64, c_i32, c_i32, P]),
    'evk_unpad_channels': (c_int, [P, P, c_i64, c_i32, c_i32, P]),
    'evk_nchw_to_nhwc': (c_int, [P, P, c_i32, c_i32, c_i32, c_i32, c_i32, P]),
    'evk_nhwc_to_nchw': (c_int, [P, P, c_i32, c_i32, c_i32, c_i32, c_i32, P]),
    'evk_bn_workspace_bytes': (c_size_t, [c_i64, c_i32]),
    'evk_bn_fwd_train': (c_int, [P, P, P, P, P, P, c_f32, c_f32, P, P, P, c_i64, c_i32, c_u32, P, c_size_t, P, P]),
    'evk_bn_fwd_train_parts': (c_int, [P, P, P, P, P, P, c_f32, c_f32, P, P, P, c_i64, c_i32, c_u32, P, c_i32, P, c_size_t, P, P]),
    'evk_bn_fwd_train_parts_bits': (c_int, [P, P, P, P, P, P, c_f32, c_f32, P, P, P, c_i64, c_i32, c_u32, P, c_i32, P, c_size_t, P, P, P]),
    'evk_relu_bits_bytes': (c_size_t, [c_i64]),
    'evk_relu_bits_apply': (c_int, [P, P, P, c_i64, P]),
    'evk_bn_fwd_eval': (c_int, [P, P, P, P, P, P, c_f32, P, P, P, c_i64, c_i32, c_u32, P, c_size_t, P, P]),
    'evk_bn_relu_pool_fwd_train_parts': (c_int, [P, P, P, P, P, c_f32, c_f32, P, P, P, P, c_i32, c_i32, c_i32, c_i32, P, c_i32,
                                                 P, c_size_t, P, P]),
    'evk_bn_relu_pool_bwd': (c_int, [P, P, P, P, P, P, P, P, P, P, c_i32, c_i32, c_i32, c_i32, c_i32, P, c_size_t, P, P]),
    'evk_bn_bwd': (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_i64, c_i32, c_u32, c_i32, P, c_size_t, P, P]),
    'evk_stream_fork': (c_int, [P, P]),
    'evk_streams_overlap': (c_int, [P, P, c_i32, P]),
    'evk_bn_relu_dot_fwd': (c_int, [P, P, P, P, P, c_i64, c_i32, c_i32, P]),
    'evk_bn_relu_dot_workspace_bytes': (c_size_t, [c_i64, c_i32, c_i32]),
    'evk_bn_relu_dot_bwd': (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, c_i64, c_i32, c_i32, c_u32, P, c_size_t, P, P]),
    'evk_bn_finalize_parts': (c_int, [P, c_i32, c_i32, c_i64, P, P, P, P, c_f32, c_f32, P, P, P, P]),
    'evk_relation_bn_fwd': (c_int, [P, P, P, P, P, P, P, c_i32, c_i32, c_i32, P, P]),
    'evk_relation_bn_parts': (c_i32, [c_i32, c_i32]),
    'evk_relation_bn_workspace_bytes': (c_size_t, [c_i32, c_i32, c_i32]),
    'evk_relation_bn_bwd': (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, c_i32, c_i32, c_i32, P, c_size_t, P]),
    'evk_bn_bwd_from_partials': (c_int, [P, P, P, P, P, P, P, c_i32, P, P, P, c_i64, c_i32, c_u32, c_i32, P, c_size_t, P, P]),
    'evk_bn_bwd_bits': (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_i64, c_i32, c_u32, c_i32, P, c_size_t, P, P, P]),
    'evk_relu_fwd': (c_int, [P, P, c_i64, P]),
    'evk_relu_bwd': (c_int, [P, P, P, c_i64, P]),
    'evk_add': (c_int, [P, P, P, c_i64, P]),
    'evk_scale': (c_int, [P, c_f32, P, c_i64, P]),
    'evk_group_weight_expand': (c_int, [P, P, c_i32, c_i32, c_i32, c_i32, P]),
    'evk_group_weight_gather': (c_int, [P, P, c_i32, c_i32, c_i32, c_i32, P]),
    'evk_mul_scale': (c_int, [P, P, c_f32, P, c_i64, P]),
    'evk_gelu_fwd': (c_int, [P, P, c_i64, P]),
    'evk_gelu_bwd': (c_int, [P, P, P, c_i64, P]),
    'evk_maxpool3x3s2_fwd': (c_int, [P, P, P, c_i32, c_i32, c_i32, c_i32, P]),
    'evk_maxpool3x3s2_bwd': (c_int, [P, P, P, c_i32, c_i32, c_i32, c_i32, P]),
    'evk_upsample_nearest2x_add_fwd': (c_int, [P, P, P, c_i32, c_i32, c_i32, c_i32, P, P]),
    'evk_upsample_nearest2x_bwd': (c_int, [P, P, c_i32, c_i32, c_i32, c_i32, P]),
    'evk_subsample2_fwd': (c_int, [P, P, c_i32, c_i32, c_i32, c_i32, P]),
    'evk_subsample2_bwd': (c_int, [P, P, c_i32, c_i32, c_i32, c_i32, P]),
    'evk_upsample_bilinear_fwd': (c_int, [P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, P]),
    'evk_upsample_bilinear_bwd': (c_int, [P, P, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, P]),
    'evk_gap_fwd': (c_int, [P, P, c_i32, c_i32, c_i32, P]),
    'evk_gap_bwd': (c_int, [P, P, c_i32, c_i32, c_i32, P]),
    'evk_relation_fwd': (c_int, [P, P, P, P, P, c_i32, c_i32, c_i32, P]),
    'evk_relation_workspace_bytes': (c_size_t, [c_i32, c_i32, c_i32]),
    'evk_relation_bwd': (c_int, [P, P, P, P, P, P, P, P, c_i32, c_i32, c_i32, P, c_size_t, P]),
    'evk_mean4_fwd': (c_int, [P, P, P, P, P, c_i64, P, P]),
    'evk_loss_stats_doubles': (c_i64, [c_i32]),
    'evk_bce_fwd': (c_int, [P, P, c_i64, c_i64, c_f32, P, P, P]),
    'evk_bce_bwd': (c_int, [P, P, c_i64, c_i64, c_f32, P, P, P, c_i32, P]),
    'evk_bce_fwd_ex': (c_int, [P, P, c_i64, c_i64, c_f32, c_f32, c_i32, P, P, P]),
    'evk_bce_bwd_ex': (c_int, [P, P, c_i64, c_i64, c_f32, c_f32, c_i32, P, P, P, c_i32, P]),
    'evk_soft_ce_fwd': (c_int, [P, P, c_i64, c_i32, P, P, P]),
    'evk_soft_ce_bwd': (c_int, [P, P, c_i64, c_i32, P, P, P]),
    'evk_dice_stats': (c_int, [P, P, c_i64, c_i32, c_i64, P, P]),
    'evk_dice_finish': (c_int, [P, c_i32, c_f32, c_i32, P, P]),
    'evk_dice_bwd': (c_int, [P, P, c_i64, c_i32, c_i64, P, c_f32, c_i32, P, P, c_i32, P]),
    'evk_ce_fwd': (c_int, [P, P, c_i64, c_i32, c_i64, c_f32, P, P, P]),
    'evk_ce_bwd': (c_int, [P, P, c_i64, c_i32, c_i64, c_f32, P, P, P, c_i32, P]),
    'evk_opt_blocks_per_tensor': (c_i32, []),
    'evk_sqnorm_multi': (c_int, [P, P, c_i32, P, c_f32, P, P, P]),
    'evk_sgd_multi': (c_int, [P, P, P, P, c_i32, c_f32, c_f32, c_f32, c_f32, c_i32, c_i32, P, P]),
    'evk_sgd_multi_lr': (c_int, [P, P, P, P, c_i32, c_f32, P, c_f32, c_f32, c_f32, c_i32, c_i32, P, P]),
    'evk_adam_multi': (c_int, [P, P, P, P, P, c_i32, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_f32, c_f32, P, P]),
    'evk_prob_stats_doubles': (c_i64, [c_i32]),
    'evk_prob_stats': (c_int, [P, P, c_i64, c_i32, c_i64, P, P]),
    'evk_prob_stats_bwd': (c_int, [P, P, c_i64, c_i32, c_i64, P, P, P, c_i32, P]),
    'evk_focal_fwd': (c_int, [P, P, c_i64, c_f32, c_f32, c_i32, c_i32, P, P, P]),
    'evk_focal_bwd': (c_int, [P, P, c_i64, c_f32, c_f32, c_i32, c_i32, P, P, P]),
    'evk_confusion_matrix': (c_int, [P, P, c_i64, c_i32, P, P]),
    'evk_confusion_from_logits': (c_int, [P, P, c_i64, c_i32, c_i32, P, P]),
    'evk_gn_workspace_bytes': (c_size_t, [c_i32, c_i64, c_i32, c_i32]),
    'evk_gn_fwd': (c_int, [P, P, P, c_f32, P, P, P, c_i32, c_i64, c_i32, c_i32, c_u32, P, c_size_t, P]),
    'evk_gn_bwd': (c_int, [P, P, P, P, P, P, P, P, P, c_i32, c_i64, c_i32, c_i32, c_u32, P, c_size_t, P]),
    'evk_concat_channels': (c_int, [P, P, P, c_i64, c_i32, c_i32, P]),
    'evk_split_channels': (c_int, [P, P, P, c_i64, c_i32, c_i32, P]),
    'evk_channel_scale': (c_int, [P, P, P, c_i32, c_i64, c_i32, P]),
    'evk_bn_local_stats': (c_int, [P, P, c_i64, c_i32, P, c_size_t, P]),
    'evk_bn_apply_stats': (c_int, [P, P, P, P, P, P, P, c_i64, c_i32, c_u32, P, c_size_t, P]),
    'evk_bn_bwd_local_sums': (c_int, [P, P, P, P, P, P, P, P, P, c_i64, c_i32, c_u32, P, c_size_t, P]),
    'evk_bn_bwd_apply_sums': (c_int, [P, P, P, P, P, P, P, P, P, P, c_i64, c_i32, c_u32, P, c_size_t, P]),
    'evk_pack_multi': (c_int, [P, P, P, c_i32, c_f32, P, P]),
    'evk_ce_pixel_fwd': (c_int, [P, P, c_i64, c_i32, c_i64, P, P]),
    'evk_ce_pixel_bwd': (c_int, [P, P, c_i64, c_i32, c_i64, P, P, P]),
    'evk_ohem_state_bytes': (c_i64, []),
    'evk_ohem_fwd': (c_int, [P, c_i64, c_i64, P, P, P]),
    'evk_ohem_bwd': (c_int, [P, c_i64, P, P, P, P]),
}

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


class HipKernelError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load():
    """Load libever_hip.so (once). Raises HipExtensionMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipExtensionMissing(
            f'{_LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C ever_amd/csrc`. ever_amd has no CPU fallback for its kernels.')
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().evk_last_error().decode('utf-8', 'replace')
        raise HipKernelError(f'{what} failed with status {rc}: {msg}')


def call(name, *args):
    """Call an int-status entry point and raise on failure."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        check(rc, name)
