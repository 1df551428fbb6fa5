"""Autograd wrappers of the SURVEY §8 f2 operators: GroupNorm(+ReLU), channel concat, Dropout2d
(include/ever_hip.h "next rows").  Same conventions as functional.py: logical NCHW, dense NHWC memory, CUDA only."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from .functional import _require_cuda, _stream, _ptr, as_nhwc, empty_nhwc, HipPathError
from .workspace import workspace

__all__ = ['group_norm_act', 'concat_channels', 'dropout2d', 'cross_entropy_per_pixel', 'online_hard_example_mining']


class _GroupNormFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu):
        n, c, h, w = x.shape
        dev = x.device
        lib = _C.load()
        ws_bytes = lib.evk_gn_workspace_bytes(n, h * w, c, groups)
        ws = workspace(dev, ws_bytes)
        y = empty_nhwc(n, c, h, w, dev)
        mean = torch.empty((n * groups,), device=dev, dtype=torch.float32)
        rstd = torch.empty((n * groups,), device=dev, dtype=torch.float32)
        _C.call('evk_gn_fwd', x.data_ptr(), _ptr(weight), _ptr(bias), eps, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                n, h * w, c, groups, 1 if relu else 0, ws.data_ptr(), ws_bytes, _stream())
        ctx.save_for_backward(x, weight, mean, rstd, y if relu else None)
        ctx.cfg = (groups, relu, bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, mean, rstd, y = ctx.saved_tensors
        groups, relu, has_bias = ctx.cfg
        n, c, h, w = x.shape
        dev = x.device
        dy = as_nhwc(dy, 'group_norm.backward')
        lib = _C.load()
        ws_bytes = lib.evk_gn_workspace_bytes(n, h * w, c, groups)
        ws = workspace(dev, ws_bytes)
        dx = empty_nhwc(n, c, h, w, dev)
        dg = torch.empty((c,), device=dev, dtype=torch.float32) if weight is not None else None
        db = torch.empty((c,), device=dev, dtype=torch.float32) if has_bias else None
        _C.call('evk_gn_bwd', dy.data_ptr(), x.data_ptr(), _ptr(y), _ptr(weight), mean.data_ptr(), rstd.data_ptr(),
                dx.data_ptr(), _ptr(dg), _ptr(db), n, h * w, c, groups, 1 if relu else 0, ws.data_ptr(), ws_bytes,
                _stream())
        return dx, dg, db, None, None, None


def group_norm_act(x, num_groups, weight=None, bias=None, eps=1e-5, relu=False):
    """nn.GroupNorm(+ReLU), reference fs_relation.py:88-116."""
    _require_cuda(x, 'group_norm')
    x = as_nhwc(x, 'group_norm')
    c = x.shape[1]
    if c % 4 != 0 or c % num_groups != 0:
        raise HipPathError(f'group_norm: {c} channels must be a multiple of 4 and of num_groups={num_groups}')
    return _GroupNormFn.apply(x, weight, bias, int(num_groups), float(eps), bool(relu))


class _ConcatFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        n, ca, h, w = a.shape
        cb = b.shape[1]
        out = empty_nhwc(n, ca + cb, h, w, a.device)
        _C.call('evk_concat_channels', a.data_ptr(), b.data_ptr(), out.data_ptr(), n * h * w, ca, cb, _stream())
        ctx.shape = (n, ca, cb, h, w)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        n, ca, cb, h, w = ctx.shape
        g = as_nhwc(g, 'concat.backward')
        ga = empty_nhwc(n, ca, h, w, g.device) if ctx.needs_input_grad[0] else None
        gb = empty_nhwc(n, cb, h, w, g.device) if ctx.needs_input_grad[1] else None
        if ga is not None or gb is not None:
            _C.call('evk_split_channels', g.data_ptr(), _ptr(ga), _ptr(gb), n * h * w, ca, cb, _stream())
        return ga, gb


def concat_channels(a, b):
    """torch.cat([a, b], dim=1), reference fs_relation.py:155."""
    _require_cuda(a, 'concat')
    a, b = as_nhwc(a, 'concat'), as_nhwc(b, 'concat')
    if a.shape[0] != b.shape[0] or a.shape[2:] != b.shape[2:]:
        raise ValueError(f'concat: shapes {tuple(a.shape)} and {tuple(b.shape)} differ outside the channel axis')
    if a.shape[1] % 4 or b.shape[1] % 4:
        raise HipPathError('concat: channel counts must be multiples of 4')
    return _ConcatFn.apply(a, b)


class _ChannelScaleFn(Function):
    @staticmethod
    def forward(ctx, x, scale):
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, h, w, x.device)
        _C.call('evk_channel_scale', x.data_ptr(), scale.data_ptr(), y.data_ptr(), n, h * w, c, _stream())
        ctx.save_for_backward(scale)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        scale, = ctx.saved_tensors
        g = as_nhwc(g, 'dropout2d.backward')
        n, c, h, w = g.shape
        dx = empty_nhwc(n, c, h, w, g.device)
        _C.call('evk_channel_scale', g.data_ptr(), scale.data_ptr(), dx.data_ptr(), n, h * w, c, _stream())
        return dx, None


def dropout2d(x, p=0.5, training=True, mask=None):
    """nn.Dropout2d: whole channels of a sample are zeroed with probability p and the rest scaled by 1/(1-p)
    (reference fs_relation.py:104,121).  `mask` ([N, C] of 0/1 keep flags) overrides the random draw."""
    if not training or p == 0.0:
        return x
    _require_cuda(x, 'dropout2d')
    x = as_nhwc(x, 'dropout2d')
    n, c = x.shape[:2]
    if c % 4:
        raise HipPathError('dropout2d: channel count must be a multiple of 4')
    if mask is None:
        mask = torch.bernoulli(torch.full((n, c), 1.0 - p, device=x.device, dtype=torch.float32))
    scale = (mask.to(device=x.device, dtype=torch.float32) / (1.0 - p)).contiguous()
    return _ChannelScaleFn.apply(x, scale)


class _MulScaleFn(Function):
    """out = x * mask * alpha (mask carries no gradient)"""

    @staticmethod
    def forward(ctx, x, mask, alpha):
        out = torch.empty_like(x)
        _C.call('evk_mul_scale', x.data_ptr(), mask.data_ptr(), alpha, out.data_ptr(), x.numel(), _stream())
        ctx.save_for_backward(mask)
        ctx.alpha = alpha
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        if g.stride() != mask.stride():
            g = torch.empty_like(mask).copy_(g)
        dx = torch.empty_like(mask)
        _C.call('evk_mul_scale', g.data_ptr(), mask.data_ptr(), ctx.alpha, dx.data_ptr(), g.numel(), _stream())
        return dx, None, None


def dropout(x, p=0.5, training=True, mask=None):
    """nn.Dropout (element-wise; reference fpn.py:183,190: the decoder's `dropout_rate`).  The 0/1 keep mask comes
    from torch's generator (same stream of random numbers as torch's own dropout would consume is NOT promised — the
    reference's results under dropout are random anyway); masking and the 1/(1-p) scaling are one HIP pass.
    `mask` (a tensor of x's shape) overrides the draw."""
    if not training or p == 0.0:
        return x
    _require_cuda(x, 'dropout')
    if p >= 1.0:
        raise ValueError('dropout: p must be < 1')
    if x.dim() == 4:
        x = as_nhwc(x, 'dropout')
    if mask is None:
        mask = torch.empty_like(x).bernoulli_(1.0 - p)
    else:
        mask = torch.empty_like(x).copy_(mask.to(device=x.device, dtype=torch.float32))
    return _MulScaleFn.apply(x, mask, 1.0 / (1.0 - p))


class _GeluFn(Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        _C.call('evk_gelu_fwd', x.data_ptr(), y.data_ptr(), x.numel(), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        if g.stride() != x.stride():
            g = torch.empty_like(x).copy_(g)
        dx = torch.empty_like(x)
        _C.call('evk_gelu_bwd', g.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), _stream())
        return dx


def gelu(x):
    """nn.GELU() (exact erf form): the decoder's activation when norm_fn is not BatchNorm2d (reference fpn.py:167)."""
    _require_cuda(x, 'gelu')
    if x.dim() == 4:
        x = as_nhwc(x, 'gelu')
    elif not x.is_contiguous():
        x = x.contiguous()
    return _GeluFn.apply(x)


class _CePixelFn(Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        n, c, h, w = logits.shape
        out = torch.empty((n, h, w), device=logits.device, dtype=torch.float32)
        _C.call('evk_ce_pixel_fwd', logits.data_ptr(), labels.data_ptr(), n * h * w, c, ignore_index, out.data_ptr(),
                _stream())
        ctx.save_for_backward(logits, labels)
        ctx.ignore_index = ignore_index
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, labels = ctx.saved_tensors
        n, c, h, w = logits.shape
        g = g.contiguous().float()
        d = torch.empty_like(logits)
        _C.call('evk_ce_pixel_bwd', logits.data_ptr(), labels.data_ptr(), n * h * w, c, ctx.ignore_index, g.data_ptr(),
                d.data_ptr(), _stream())
        return d, None, None


def cross_entropy_per_pixel(y_pred, y_true, ignore_index=255):
    """F.cross_entropy(y_pred, y_true, ignore_index=..., reduction='none') -> [N, H, W]"""
    _require_cuda(y_pred, 'cross_entropy_per_pixel')
    y_pred = as_nhwc(y_pred, 'cross_entropy_per_pixel')
    if y_pred.shape[1] < 2:
        raise HipPathError('cross_entropy_per_pixel: needs at least 2 classes')
    yt = y_true.to(torch.int64).contiguous()
    if yt.numel() != y_pred.numel() // y_pred.shape[1]:
        raise ValueError('cross_entropy_per_pixel: label / logit pixel counts differ')
    return _CePixelFn.apply(y_pred, yt, int(ignore_index))


class _OhemFn(Function):
    @staticmethod
    def forward(ctx, losses, keep):
        lib = _C.load()
        state = torch.empty((lib.evk_ohem_state_bytes(),), device=losses.device, dtype=torch.uint8)
        loss = torch.empty((), device=losses.device, dtype=torch.float32)
        _C.call('evk_ohem_fwd', losses.data_ptr(), losses.numel(), keep, loss.data_ptr(), state.data_ptr(), _stream())
        ctx.save_for_backward(losses, state)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        losses, state = ctx.saved_tensors
        g = g.contiguous().float()
        d = torch.empty_like(losses)
        _C.call('evk_ohem_bwd', losses.data_ptr(), losses.numel(), state.data_ptr(), g.data_ptr(), d.data_ptr(), _stream())
        return d, None


def online_hard_example_mining(losses, keep_ratio):
    """reference ever/module/loss.py:146-155: mean of the non-zero values among the int(keep_ratio * N) largest."""
    assert 0 < keep_ratio < 1, 'The value of keep_ratio must be from 0 to 1.'
    if not losses.is_cuda or losses.dtype != torch.float32:
        raise HipPathError('online_hard_example_mining: fp32 CUDA tensor required')
    flat = losses.contiguous().view(-1) if losses.is_contiguous() else losses.reshape(-1).contiguous()
    keep = int(keep_ratio * flat.numel())
    if keep < 1:
        raise ValueError('online_hard_example_mining: keep_ratio * numel < 1')
    return _OhemFn.apply(flat, keep)
