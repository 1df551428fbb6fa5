"""Shared plumbing of the host wrappers (hip/functional.py is the facade that re-exports every family): arithmetic modes of
the convolution GEMMs, operand-scale (absmax) buffers, packed / lazy tensor marks, NHWC helpers, raw stream and pointer
access, the timed C-ABI call.  No kernels of its own."""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from . import timing, weight_planes
from .workspace import workspace


class HipPathError(RuntimeError):
    """Raised when the HIP path is asked to run on something it cannot (CPU tensor, wrong dtype)."""


# Arithmetic of the convolution GEMMs.  Both are fp32 in, fp32 out, fp32 accumulate:
#   'bf16x3' (default) each fp32 operand is split exactly into three bf16 terms and the product is rebuilt from
#            six bf16 MFMA partial products (error per product < 2^-24: below one fp32 rounding), on the
#            v_mfma_f32_32x32x16_bf16 pipe;
#   'f32'    v_mfma_f32_32x32x2_f32, an exact fmaf chain (the parity yardstick for the split kernels).
#   'bf16'   plain bf16 operands (rounded once), ONE bf16 MFMA product per operand pair, fp32 accumulate, fp32 tensors:
#            the counterpart of the reference's `--mixed_precision bf16` (core/launcher.py:40-80).  Opt-in only.
#   'f16x2'  each fp32 operand is divided by a per-tensor power of two and split into two fp16 terms; three partial
#            products on the fp16 matrix pipe, fp32 accumulate, scales multiplied back (csrc/x3_common.hpp).  As accurate
#            against fp64 as 'bf16x3' on every layer shape (tools/check_f16x2.py) at half the matrix work.
_CONV_MATH = os.environ.get('EVK_CONV_MATH', 'f16x2')
_MATH_MODES = ('f16x2', 'bf16x3', 'f32', 'bf16')


def _planes_math():
    """True when the convolutions run on the bf16 matrix pipe from weight planes (exact split or plain bf16)."""
    return _CONV_MATH in ('f16x2', 'bf16x3', 'bf16')


def _f16x2():
    return _CONV_MATH == 'f16x2'


absmax_stats = {'hits': 0, 'standalone': 0, 'fused': 0, 'packed': 0}   # where the operand scales came from (tools / tests)
_FUSED_AMAX = os.environ.get('EVK_FUSED_ABSMAX', '1') != '0'


def _note_amax(t, bits):
    """Record that `bits` holds max|t| (written by the kernel that produced t)."""
    try:
        t._evk_amax = (t._version, t.data_ptr(), bits)
        absmax_stats['fused'] += 1
    except (AttributeError, RuntimeError):
        pass


def _inherit_amax(out, src):
    """`out` is bounded element-wise by max|src| (a convex combination, a selection, a product with a factor in [0, 1]):
    src's word is a valid — at most slightly loose — scale source for out, and saves its read pass.  The f16x2 operand
    keeps its 22 bits for every element within 2^-17 of the bound (csrc/x3_common.hpp)."""
    hit = getattr(src, '_evk_amax', None)
    if _FUSED_AMAX and hit is not None and hit[0] == src._version and hit[1] == src.data_ptr():
        _note_amax(out, hit[2])
    return out


_AMAX_WORDS = [0]


def _amax_buf(dev):
    """An activation scale buffer (64 slots of partial max|x| bit images, include/ever_hip.h: evk_absmax)."""
    if not _AMAX_WORDS[0]:
        _AMAX_WORDS[0] = int(_C.load().evk_absmax_words())
    return torch.empty((_AMAX_WORDS[0],), device=dev, dtype=torch.int32)


def absmax_value(bits):
    """The bit image of max|x| held by an activation scale buffer (tests, tools): the maximum of its slots."""
    n = bits.numel() // 64
    return int(bits.view(64, n)[:, 0].max().item())


_ZERO_POOL = {}


def _amax_zeroed(dev):
    """A scale buffer whose slots are zero (for producers that only RAISE slots: the convolution epilogues), or None.
    Buffers are cut from a pool zeroed 256 at a time: one fill launch per 256 convolution outputs."""
    if not (_FUSED_AMAX and _f16x2()):
        return None
    if not _AMAX_WORDS[0]:
        _AMAX_WORDS[0] = int(_C.load().evk_absmax_words())
    nw = _AMAX_WORDS[0]
    key = (dev, _stream())          # the fill launch and the kernels that raise the slots must share a stream's order
    pool = _ZERO_POOL.get(key)
    if pool is None or pool[1] >= pool[0].shape[0]:
        pool = _ZERO_POOL[key] = [torch.zeros((256, nw), device=dev, dtype=torch.int32), 0]
    buf = pool[0][pool[1]]
    pool[1] += 1
    return buf


def _amax_out(dev):
    """A scale buffer for a producer kernel to leave max|output| in, or None when no consumer will want it."""
    return _amax_buf(dev) if (_FUSED_AMAX and _f16x2()) else None


def absmax_bits(t, st):
    """Activation scale buffer of t (the f16x2 operand scale derives from the maximum of its slots inside the kernels).  Cached on the tensor object: the forward's scale of x serves the weight gradient, the scale of dy serves
    data and weight gradient, a block input serves both convolutions that read it."""
    hit = getattr(t, '_evk_amax', None)
    if hit is not None and hit[0] == t._version and hit[1] == t.data_ptr():
        absmax_stats['hits'] += 1
        return hit[2]
    absmax_stats['standalone'] += 1
    bits = _amax_buf(t.device)
    sp = timing.span('absmax', 0.0, 4.0 * t.numel())
    _C.call('evk_absmax', t.data_ptr(), t.numel(), bits.data_ptr(), weight_planes.absmax_workspace(t.device, st).data_ptr(), st)
    if sp is not None:
        sp.stop()
    try:
        t._evk_amax = (t._version, t.data_ptr(), bits)
    except (AttributeError, RuntimeError):
        pass
    return bits


def _weight_planes(weight, w_dense, w_ptr, d, for_dgrad, st, dev):
    """(planes pointer, weight absmax pointer or None, keep-alive) for the current plane arithmetic: from the cache of
    registered weights, else split into the shared workspace on this call (a transient re-laid-out copy)."""
    h2 = _f16x2()
    hit = weight_planes.planes_for(weight, w_dense, d, for_dgrad, st, f16x2=h2)
    if hit is not None:
        return (hit[0], hit[1], None) if h2 else (hit, None, None)
    planes = workspace(dev, _C.load().evk_conv2d_split_weight_bytes(ctypes.byref(d), for_dgrad))
    if h2:
        wb = _amax_buf(dev)          # (slot 0 = the maximum: a valid single word for the kernels' w_absmax)
        nel = d.Cout * d.kh * d.kw * d.Cin
        _C.call('evk_absmax', w_ptr, nel, wb.data_ptr(), weight_planes.absmax_workspace(dev, st).data_ptr(), st)
        _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), w_ptr, for_dgrad, planes.data_ptr(), wb.data_ptr(), st)
        return planes.data_ptr(), wb.data_ptr(), wb
    _C.call('evk_conv2d_split_weight', ctypes.byref(d), w_ptr, for_dgrad, planes.data_ptr(), st)
    return planes.data_ptr(), None, None


def _entry(x3_name):
    """C-ABI entry point of the current plane arithmetic: evk_*_x3 or its plain-bf16 twin evk_*_bf16."""
    return x3_name if _CONV_MATH != 'bf16' else x3_name.replace('_x3', '_bf16')


def set_conv_math(mode):
    global _CONV_MATH
    if mode not in _MATH_MODES:
        raise ValueError(f"conv math must be one of {_MATH_MODES}, got {mode!r}")
    prev, _CONV_MATH = _CONV_MATH, mode
    return prev


def get_conv_math():
    return _CONV_MATH


# BatchNorm statistics from the producing convolution's epilogue (EVK_BN_EPILOGUE=0: BatchNorm's own statistics pass)
_BN_EPILOGUE = os.environ.get('EVK_BN_EPILOGUE', '1') != '0'
# max|y| (or its bound) of the BatchNorm pass that has just run, for the wrapper that returns y: [(scale buffer, packed)] or [None]
_AMAX_HANDOFF = [None]
# f16x2: activations that only convolutions read are stored already split ("packed", include/ever_hip.h:
# evk_pack_f16x2) by the BatchNorm pass that writes them (EVK_PACKED=0: fp32 everywhere, split while staging)
_PACKED = os.environ.get('EVK_PACKED', '1') != '0'


def _dist_initialized():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def _collectives_world():
    import torch.distributed as dist
    return dist.get_world_size() if _dist_initialized() else 1


def _mark_packed(t, bits):
    """`t` holds packed words of t / s (s from `bits`), not fp32: only the f16x2 convolution kernels may read it."""
    _note_amax(t, bits)
    t._evk_packed = (t._version, t.data_ptr())
    absmax_stats['packed'] += 1
    return t


def _is_packed(t):
    hit = getattr(t, '_evk_packed', None)
    return hit is not None and hit[0] == t._version and hit[1] == t.data_ptr()


# ReLU bits (include/ever_hip.h: evk_bn_fwd_train_parts_bits): the BatchNorm + add + ReLU that ends a residual block keeps one
# bit per output element and its backward reads those instead of the output tensor; with `lazy_res` the identity branch's
# gradient is not written either — the unmasked incoming gradient travels on with the bits (`_evk_relu_bits` on a view of
# it) and is masked by its consumer: the block input's data-gradient launch while it adds, or the shortcut's BatchNorm.
_RELU_BITS = os.environ.get('EVK_RELU_BITS', '1') != '0'
_LAZY_RES = os.environ.get('EVK_LAZY_RES', '1') != '0'
relu_bits_stats = {'forward': 0, 'lazy': 0, 'masked_dgrad': 0, 'masked_bn': 0, 'materialized': 0}   # tests / tools


def _lazy_bits(t):
    """The ReLU bits an unmasked gradient travels with, or None."""
    hit = getattr(t, '_evk_relu_bits', None)
    if hit is not None and hit[0] == t._version and hit[1] == t.data_ptr():
        return hit[2]
    return None


def materialize_lazy(t):
    """`t` itself unless it is an unmasked gradient travelling with ReLU bits: then the masked tensor (one pass)."""
    bits = _lazy_bits(t) if t is not None else None
    if bits is None:
        return t
    out = torch.empty_like(t)
    _C.call('evk_relu_bits_apply', t.data_ptr(), bits.data_ptr(), out.data_ptr(), t.numel(), _stream())
    relu_bits_stats['materialized'] += 1
    return _inherit_amax(out, t)


_top_saved_hooks = getattr(torch._C._autograd, '_top_saved_tensors_default_hooks', None)


def observers_active():
    """True when something other than this package's own kernels may get to see an activation between its producer and
    its consumer: saved-tensor hooks (non-reentrant checkpointing, save_on_cpu: the unpack hook returns a NEW tensor
    object, which would not carry the `_evk_packed` mark) or global module forward hooks.  Packed tensors are raw words
    marked only by a Python attribute, so nothing is stored packed while an observer is installed; per-module
    hooks are checked by the layers (module/layers.py)."""
    if _top_saved_hooks is not None and _top_saved_hooks(True) is not None:
        return True
    from torch.nn.modules import module as _m
    return bool(_m._global_forward_hooks or _m._global_forward_pre_hooks or _m._global_forward_hooks_always_called)


def unpacked(t):
    """`t` as fp32 values: itself unless it holds packed f16x2 words, then (h + l) * s by evk_unpack_f16x2.  For readers
    outside the f16x2 convolution kernels (folded inference convolutions, debugging)."""
    if not _is_packed(t):
        return t
    bits = t._evk_amax[2]
    out = torch.empty_like(t)
    _C.call('evk_unpack_f16x2', t.data_ptr(), t.numel(), bits.data_ptr(), out.data_ptr(), _stream())
    _note_amax(out, bits)
    return out

_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """Raw handle of the current HIP stream of the current device.  ~250 calls per training step: the raw getter skips
    the Stream-object construction of torch.cuda.current_stream() (2 ms of host time per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _timed_call(family, nbytes, name, *args):
    """_C.call bracketed by HIP events when bench.py's KernelTimer is active (HBM-bound families:
    `nbytes` = algorithmic bytes of the call)."""
    sp = timing.span(family, 0.0, nbytes)
    _C.call(name, *args)
    if sp is not None:
        sp.stop()


def _require_cuda(t, what):
    if not t.is_cuda:
        raise HipPathError(
            f'{what}: ever_amd kernels run on MI355X only (got a {t.device} tensor). '
            f'There is no CPU fallback; move the model and data to cuda.')
    if t.dtype != torch.float32:
        raise HipPathError(f'{what}: fp32 tensors required, got {t.dtype}')


def empty_nhwc(n, c, h, w, device, dtype=torch.float32):
    """Logical [n,c,h,w] tensor over dense NHWC memory."""
    return torch.empty((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


def is_nhwc(t):
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def as_nhwc(t, what='tensor'):
    """Return `t` if its memory is already dense NHWC, else transpose it with the HIP kernel."""
    _require_cuda(t, what)
    if is_nhwc(t):
        return t
    return _ToNHWC.apply(t)


class _ToNHWC(Function):
    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        xc = x.contiguous()  # dense NCHW source
        out = empty_nhwc(n, c, h, w, x.device)
        _C.call('evk_nchw_to_nhwc', xc.data_ptr(), out.data_ptr(), n, c, h, w, c, _stream())
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g


def image_to_nhwc(x, cpad):
    """Model-boundary transpose: NCHW image -> NHWC with channels zero-padded to `cpad` (no grad)."""
    _require_cuda(x, 'image_to_nhwc')
    n, c, h, w = x.shape
    out = empty_nhwc(n, cpad, h, w, x.device)
    if is_nhwc(x):
        _C.call('evk_pad_channels', x.data_ptr(), out.data_ptr(), n * h * w, c, cpad, _stream())
    else:
        xc = x.contiguous()
        _C.call('evk_nchw_to_nhwc', xc.data_ptr(), out.data_ptr(), n, c, h, w, cpad, _stream())
    return out


# shape helpers of the dispatcher-level operators (hip/oplib.py) the family modules register
def _conv_out(h, k, s, p, d):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


def _same_shape_fake(x, *rest):
    from . import oplib as _oplib
    return _oplib.nhwc_like(x, *x.shape) if x.dim() == 4 else torch.empty_like(x)
