"""BatchNorm family of the host wrappers (reference _resnets.py:95-112, fs_relation.py:39-53, fpn.py:163-167;
include/ever_hip.h: evk_bn_*): BatchNorm + residual + ReLU forward / backward, the stem's BatchNorm + ReLU + max-pool.
Part of the hip/functional.py facade."""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from . import timing, weight_planes
from .workspace import workspace
from ._base import (  # noqa: F401
    HipPathError, _AMAX_HANDOFF, _LAZY_RES, _PACKED, _RELU_BITS, _amax_out, _amax_zeroed, _f16x2, _inherit_amax,
    _lazy_bits, _mark_packed, _note_amax, _ptr, _require_cuda, _stream, _timed_call, as_nhwc, empty_nhwc, materialize_lazy,
    observers_active, relu_bits_stats,
)


# ------------------------------------------------------------------------------------ batch norm
class _BatchNormActFn(Function):
    """BatchNorm2d (+ residual add) (+ ReLU) in one pass.

    Replaces aten::batch_norm, `out += identity`, relu_ at reference ever/module/_resnets.py:95-112,
    fs_relation.py:39-53, fpn.py:163-167.
    """

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, training, momentum, eps, relu, parts=None,
                pack_out=False, lazy_res=False):
        n, c, h, w = x.shape
        rows = n * h * w
        dev = x.device
        st = _stream()
        lib = _C.load()
        ws_bytes = lib.evk_bn_workspace_bytes(rows, c)
        ws = workspace(dev, ws_bytes)
        y = empty_nhwc(n, c, h, w, dev)
        save_mean = torch.empty((c,), device=dev, dtype=torch.float32)
        save_invstd = torch.empty((c,), device=dev, dtype=torch.float32)
        flags = 1 if relu else 0
        # pack_out: y is one convolution's operand and nothing else — written packed, its scale bounded from the
        # statistics records before the apply pass (EVK_BN_PACK_Y); rows: that convolution must be on the plane kernels
        pack = bool(pack_out and _PACKED and _f16x2() and training and parts is not None and residual is None
                    and c % 8 == 0 and rows >= 256 and not observers_active())
        abits = _amax_zeroed(dev) if pack else _amax_out(dev)
        pack = pack and abits is not None
        if pack:
            flags |= 4
        # algorithmic bytes (fp32): statistics read + apply read/write (+ residual read)
        nb = 4.0 * x.numel() * ((3 if training else 2) + (1 if residual is not None else 0))
        rbits = None
        if training and parts is not None and residual is not None and relu and _RELU_BITS:
            # the end of a residual block: the ReLU bits go out beside y and the backward reads them instead of y
            rbits = torch.empty((lib.evk_relu_bits_bytes(x.numel()) // 4,), device=dev, dtype=torch.int32)
            relu_bits_stats['forward'] += 1
            _timed_call('bn', nb - 4.0 * x.numel(), 'evk_bn_fwd_train_parts_bits', x.data_ptr(), _ptr(residual), _ptr(weight),
                        _ptr(bias), _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), y.data_ptr(),
                        save_mean.data_ptr(), save_invstd.data_ptr(), rows, c, flags, parts[0].data_ptr(), parts[1],
                        ws.data_ptr(), ws_bytes, _ptr(abits), rbits.data_ptr(), st)
        elif training and parts is not None:
            # statistics came with x from the convolution's epilogue: merge the records, apply (2|x| of traffic)
            _timed_call('bn', nb - 4.0 * x.numel(), 'evk_bn_fwd_train_parts', x.data_ptr(), _ptr(residual), _ptr(weight),
                        _ptr(bias), _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), y.data_ptr(),
                        save_mean.data_ptr(), save_invstd.data_ptr(), rows, c, flags, parts[0].data_ptr(), parts[1],
                        ws.data_ptr(), ws_bytes, _ptr(abits), st)
        elif training:
            _timed_call('bn', nb, 'evk_bn_fwd_train', x.data_ptr(), _ptr(residual), _ptr(weight), _ptr(bias), _ptr(running_mean),
                    _ptr(running_var), float(momentum), float(eps), y.data_ptr(), save_mean.data_ptr(),
                    save_invstd.data_ptr(), rows, c, flags, ws.data_ptr(), ws_bytes, _ptr(abits), st)
        else:
            _timed_call('bn', nb, 'evk_bn_fwd_eval', x.data_ptr(), _ptr(residual), _ptr(weight), _ptr(bias), running_mean.data_ptr(),
                    running_var.data_ptr(), float(eps), y.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr(),
                    rows, c, flags, ws.data_ptr(), ws_bytes, _ptr(abits), st)
        _AMAX_HANDOFF[0] = (abits, pack)
        ctx.training = training
        ctx.pack_dx = bool(parts is not None and len(parts) > 2 and parts[2])
        ctx.relu = relu
        ctx.has_res = residual is not None
        # the ReLU mask is recomputed from x in backward unless a residual was added (then y — or its bits — is needed)
        ctx.lazy_res = bool(lazy_res and rbits is not None and _LAZY_RES and not observers_active())
        ctx.save_for_backward(x, y if (relu and residual is not None and rbits is None) else None, weight, bias, save_mean,
                              save_invstd, rbits)
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var) if t is not None])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y, weight, bias, save_mean, save_invstd, rbits = ctx.saved_tensors
        n, c, h, w = x.shape
        rows = n * h * w
        dev = x.device
        st = _stream()
        in_bits = _lazy_bits(dy)
        if in_bits is not None and (ctx.relu or rbits is not None):
            dy, in_bits = materialize_lazy(dy), None     # an own mask AND an incoming one: apply the incoming one first
        dy = as_nhwc(dy, 'batch_norm.backward')
        lib = _C.load()
        ws_bytes = lib.evk_bn_workspace_bytes(rows, c)
        ws = workspace(dev, ws_bytes)
        dx = torch.empty_like(x)
        need_res = ctx.has_res and ctx.needs_input_grad[1]
        lazy = need_res and ctx.lazy_res and rbits is not None and in_bits is None
        dres = torch.empty_like(x) if (need_res and not lazy) else None
        has_affine = weight is not None
        dgamma = torch.empty((c,), device=dev, dtype=torch.float32) if has_affine else None
        dbeta = torch.empty((c,), device=dev, dtype=torch.float32) if has_affine else None
        # reduce pass reads dy, x (+y mask); apply pass reads g, x and writes dx (+ the residual gradient write)
        nb = 4.0 * x.numel() * (5 + (1 if y is not None else 0) + (1 if dres is not None else 0))
        # dx is the producing convolution's dy (data and weight gradient operand) — and, when that convolution said so
        # in forward, nothing else: written packed under a scale bounded before the apply pass (EVK_BN_PACK_DX)
        pack = ctx.pack_dx and _f16x2()
        abits = _amax_zeroed(dev) if pack else _amax_out(dev)
        pack = pack and abits is not None
        mask_bits = rbits if rbits is not None else in_bits
        if in_bits is not None:
            relu_bits_stats['masked_bn'] += 1
        _timed_call('bn', nb, 'evk_bn_bwd_bits', dy.data_ptr(), x.data_ptr(), _ptr(y), _ptr(weight), _ptr(bias),
                    save_mean.data_ptr(), save_invstd.data_ptr(), dx.data_ptr(), _ptr(dres), _ptr(dgamma), _ptr(dbeta), rows, c,
                    (1 if ctx.relu else 0) | (2 if pack else 0),
                    1 if ctx.training else 0, ws.data_ptr(), ws_bytes, _ptr(abits), _ptr(mask_bits), st)
        if lazy:
            # the identity branch's gradient = dy where the block's output was positive: handed on unmasked with the bits
            relu_bits_stats['lazy'] += 1
            dres = dy.view_as(dy)
            dres._evk_relu_bits = (dres._version, dres.data_ptr(), rbits)
            _inherit_amax(dres, dy)
        if pack:
            _mark_packed(dx, abits)
        elif abits is not None:
            _note_amax(dx, abits)
        if ctx.has_res and not need_res:
            dres = None
        return (dx, dres, dgamma if ctx.needs_input_grad[2] else None, dbeta if ctx.needs_input_grad[3] else None,
                None, None, None, None, None, None, None, None, None)


def batch_norm_act(x, weight, bias, running_mean, running_var, training, momentum, eps, residual=None, relu=False,
                   pack_out=False, lazy_res=False):
    """lazy_res: the caller guarantees that the gradient of `residual` reaches only readers of this package that take an
    unmasked gradient with ReLU bits (a `conv2d_fork` identity output, or the BatchNorm of a shortcut convolution): the
    backward then hands the incoming gradient on as it is instead of writing a masked copy (EVK_LAZY_RES).
    pack_out: the caller guarantees that ONE convolution of this package (forward + weight gradient) is the only
    reader of the result; under the f16x2 arithmetic it is then stored packed (include/ever_hip.h: EVK_BN_PACK_Y)."""
    _require_cuda(x, 'batch_norm')
    x = as_nhwc(x, 'batch_norm')
    if x.shape[1] % 4 != 0:
        raise HipPathError(f'batch_norm: channel count {x.shape[1]} must be a multiple of 4')
    if residual is not None:
        residual = as_nhwc(residual, 'batch_norm.residual')
    use_batch_stats = training or running_mean is None
    parts = getattr(x, '_evk_bn_parts', None) if use_batch_stats else None
    if parts is not None:
        del x._evk_bn_parts
    _AMAX_HANDOFF[0] = None
    if use_batch_stats and running_mean is not None:
        weight_planes.note_running_stats_changed()
    y = _BatchNormActFn.apply(x, residual, weight, bias, running_mean, running_var, bool(use_batch_stats),
                              0.0 if momentum is None else momentum, eps, bool(relu), parts, bool(pack_out), bool(lazy_res))
    if _AMAX_HANDOFF[0] is not None:       # the pass left max|y| (or its bound) there: y is the next convolution's operand
        abits, packed = _AMAX_HANDOFF[0]
        if packed:
            _mark_packed(y, abits)
        elif abits is not None:
            _note_amax(y, abits)
        _AMAX_HANDOFF[0] = None
    return y


class _BnReluPoolFn(Function):
    """The stem's BatchNorm (batch statistics from the convolution epilogue's records) + ReLU + MaxPool2d(3, 2, 1) as
    one pass each way (csrc/bn.hip: bn_relu_pool_fwd_kernel): the normalised full-resolution map and its gradient are
    never written.  Replaces bn1 / relu / maxpool of reference ever/module/_resnets.py:150-153."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, parts):
        n, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        dev, st = x.device, _stream()
        ws_bytes = _C.load().evk_bn_workspace_bytes(n * h * w, c)
        ws = workspace(dev, ws_bytes)
        y = empty_nhwc(n, c, ho, wo, dev)
        code = torch.empty((n, ho, wo, c), device=dev, dtype=torch.uint8)
        save_mean = torch.empty((c,), device=dev, dtype=torch.float32)
        save_invstd = torch.empty((c,), device=dev, dtype=torch.float32)
        abits = _amax_out(dev)
        # algorithmic bytes: read x, write the pooled map and the codes
        nb = 4.0 * x.numel() + 5.0 * y.numel()
        _timed_call('bn', nb, 'evk_bn_relu_pool_fwd_train_parts', x.data_ptr(), _ptr(weight), _ptr(bias), _ptr(running_mean),
                    _ptr(running_var), float(momentum), float(eps), y.data_ptr(), code.data_ptr(), save_mean.data_ptr(),
                    save_invstd.data_ptr(), n, h, w, c, parts[0].data_ptr(), parts[1], ws.data_ptr(), ws_bytes, _ptr(abits), st)
        _AMAX_HANDOFF[0] = (abits, False)
        ctx.save_for_backward(x, weight, bias, save_mean, save_invstd, code)
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var) if t is not None])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dp):
        x, weight, bias, save_mean, save_invstd, code = ctx.saved_tensors
        n, c, h, w = x.shape
        dev, st = x.device, _stream()
        dp = as_nhwc(dp, 'bn_relu_pool.backward')
        ws_bytes = _C.load().evk_bn_workspace_bytes(n * h * w, c)
        ws = workspace(dev, ws_bytes)
        dx = torch.empty_like(x)
        has_affine = weight is not None
        dgamma = torch.empty((c,), device=dev, dtype=torch.float32) if has_affine else None
        dbeta = torch.empty((c,), device=dev, dtype=torch.float32) if has_affine else None
        abits = _amax_out(dev)
        # both passes read x, the pooled gradient and the codes; the apply pass writes dx
        nb = 4.0 * x.numel() * 3 + 2 * 5.0 * dp.numel()
        _timed_call('bn', nb, 'evk_bn_relu_pool_bwd', dp.data_ptr(), code.data_ptr(), x.data_ptr(), _ptr(weight), _ptr(bias),
                    save_mean.data_ptr(), save_invstd.data_ptr(), dx.data_ptr(), _ptr(dgamma), _ptr(dbeta), n, h, w, c, 1,
                    ws.data_ptr(), ws_bytes, _ptr(abits), st)
        if abits is not None:
            _note_amax(dx, abits)
        return (dx, dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None,
                None, None, None, None, None)


_STEM_POOL = os.environ.get('EVK_STEM_POOL', '1') != '0'


def batch_norm_relu_max_pool(x, weight, bias, running_mean, running_var, momentum, eps):
    """max_pool3x3s2(relu(batch_norm(x))) in training mode.  One fused pass each way when x carries the statistics
    records of the convolution that produced it (conv2d / stem_conv7x7s2 with bn_stats=True); the two separate passes
    otherwise (EVK_STEM_POOL=0 forces them)."""
    _require_cuda(x, 'batch_norm_relu_max_pool')
    x = as_nhwc(x, 'batch_norm_relu_max_pool')
    parts = getattr(x, '_evk_bn_parts', None)
    n, c, h, w = x.shape
    if parts is None or not _STEM_POOL or c % 4 or n * h * w * (c // 4) >= 2 ** 31:
        from .pointwise import max_pool3x3s2      # (pointwise imports this module's neighbours)
        return max_pool3x3s2(batch_norm_act(x, weight, bias, running_mean, running_var, True, momentum, eps, relu=True))
    del x._evk_bn_parts
    _AMAX_HANDOFF[0] = None
    if running_mean is not None:
        weight_planes.note_running_stats_changed()
    y = _BnReluPoolFn.apply(x, weight, bias, running_mean, running_var, 0.0 if momentum is None else momentum, eps, parts)
    if _AMAX_HANDOFF[0] is not None:
        if _AMAX_HANDOFF[0][0] is not None:
            _note_amax(y, _AMAX_HANDOFF[0][0])
        _AMAX_HANDOFF[0] = None
    return y


# ---------------------------------------------------------------------------------------------------------------------
# The no-grad forward of this family's layers as dispatcher-level operators (hip/oplib.py): what `torch.jit.trace`
# (reference api/infer_tool.py:70-74: export_model) and a compiler's shape pass record instead of an opaque Python call.
# Eager calls keep the direct path; the names below are what the modules (and this file) call from here on.
from . import oplib as _oplib  # noqa: E402


_bn_act_plain = batch_norm_act
batch_norm_act = _oplib.traceable(
    'batch_norm_eval', '(Tensor x, Tensor? weight, Tensor? bias, Tensor running_mean, Tensor running_var, float eps, '
                       'Tensor? residual, bool relu) -> Tensor',
    _bn_act_plain, impl_fn=lambda x, w, b, rm, rv, eps, res, relu: _bn_act_plain(x, w, b, rm, rv, False, 0.1, eps, residual=res, relu=relu),
    adapt=lambda x, weight, bias, running_mean, running_var, training, momentum, eps, residual=None, relu=False, pack_out=False,
        lazy_res=False: (x, weight, bias, running_mean, running_var, float(eps), residual, bool(relu)),
    fake=lambda x, w, b, rm, rv, eps, res, relu: _oplib.nhwc_like(x, *x.shape),
    applies=lambda x, weight, bias, running_mean, running_var, training, *a, **k: not training and running_mean is not None)
