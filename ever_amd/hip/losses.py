"""Pixel losses and metrics of the host wrappers (reference ever/module/loss.py:10-75,207-235 and the FarSeg++ / ChangeStar
losses; include/ever_hip.h: evk_bce_* / evk_dice_* / evk_ce_* / evk_focal_* / evk_confusion_*).  Part of the
hip/functional.py facade."""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from . import timing, weight_planes
from .workspace import workspace
from ._base import (  # noqa: F401
    _require_cuda, _stream, _timed_call, as_nhwc, is_nhwc,
)


# ------------------------------------------------------------------------------------ losses
def _stats_buf(k, device):
    n = _C.load().evk_loss_stats_doubles(k)
    return torch.empty((n,), device=device, dtype=torch.float64)


def _labels(y_true, npix, what):
    if y_true.dtype != torch.int64:
        y_true = y_true.long()
    y_true = y_true.contiguous()
    if y_true.numel() != npix:
        raise ValueError(f'{what}: labels have {y_true.numel()} elements, logits have {npix} pixels')
    return y_true


class _BceFn(Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index, eps, pos_weight, reduction):
        npix = logits.numel()
        stats = _stats_buf(2, logits.device)
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        _timed_call('resample_loss', 12.0 * npix, 'evk_bce_fwd_ex', logits.data_ptr(), labels.data_ptr(), npix, ignore_index, eps,
                    pos_weight, reduction, loss.data_ptr(), stats.data_ptr(), _stream())
        ctx.save_for_backward(logits, labels, stats)
        ctx.ignore_index = ignore_index
        ctx.eps = eps
        ctx.pw, ctx.red = pos_weight, reduction
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, labels, stats = ctx.saved_tensors
        g = g.contiguous().float()
        d = torch.empty_like(logits)
        _timed_call('resample_loss', 16.0 * logits.numel(), 'evk_bce_bwd_ex', logits.data_ptr(), labels.data_ptr(), logits.numel(), ctx.ignore_index, ctx.eps,
                    ctx.pw, ctx.red, stats.data_ptr(), g.data_ptr(), d.data_ptr(), 0, _stream())
        return d, None, None, None, None, None


def bce_with_logits(y_pred, y_true, ignore_index=255, label_smoothing=0.0, pos_weight=None, reduction='mean'):
    """reference ever/module/loss.py:229-235 (reduction 'mean' | 'sum', optional pos_weight: a number or a one-element
    tensor — the heads on this path have one logit channel); label_smoothing > 0 gives
    label_smoothing_binary_cross_entropy (loss.py:222-226)."""
    if reduction not in ('mean', 'sum', 'none'):
        raise ValueError(f"binary_cross_entropy_with_logits: reduction '{reduction}'")
    if pos_weight is None:
        pw = 1.0
    elif isinstance(pos_weight, torch.Tensor):
        if pos_weight.numel() != 1:
            raise NotImplementedError('binary_cross_entropy_with_logits: pos_weight must have one element (one logit channel)')
        pw = float(pos_weight.reshape(()).item())
    else:
        pw = float(pos_weight)
    _require_cuda(y_pred, 'binary_cross_entropy_with_logits')
    if y_pred.dim() == 4:
        if y_pred.shape[1] != 1:
            raise ValueError('binary_cross_entropy_with_logits: logits must have one channel')
        y_pred = as_nhwc(y_pred, 'bce')
    else:
        y_pred = y_pred.contiguous()
    labels = _labels(y_true, y_pred.numel(), 'bce')
    if reduction == 'none':
        # One value per NON-ignored pixel, in pixel order (the reference compacts logits and targets with masked_select first,
        # loss.py:10-17,229-235): a data-dependent shape, so the compaction is a boolean index (one host round trip, as in the
        # reference).  Per pixel BCE(z, t) = t * CE([0, z], 1) + (1 - t) * CE([0, z], 0): the two-class per-pixel cross entropy
        # kernel on the logit pair (0, z), which is the same softplus arithmetic.
        from . import functional_next as HN
        z = y_pred.reshape(-1, 1, 1, 1)
        z2 = as_nhwc(torch.cat([torch.zeros_like(z), z], dim=1), 'bce.none')
        yl = labels.reshape(-1, 1, 1)
        valid = yl != ignore_index
        if not label_smoothing and pw == 1.0:
            per = HN.cross_entropy_per_pixel(z2, yl, ignore_index)
        else:
            l1 = HN.cross_entropy_per_pixel(z2, torch.where(valid, torch.ones_like(yl), yl), ignore_index)
            l0 = HN.cross_entropy_per_pixel(z2, torch.where(valid, torch.zeros_like(yl), yl), ignore_index)
            t = yl.to(torch.float32)
            if label_smoothing:
                t = torch.where(yl == 0, t + label_smoothing, t - label_smoothing)
            per = pw * t * l1 + (1.0 - t) * l0
        return per.reshape(-1)[valid.reshape(-1)]
    return _BceFn.apply(y_pred, labels, int(ignore_index), float(label_smoothing), pw, 0 if reduction == 'mean' else 1)


_rank_sum_hook = None  # tests: callable(tensor, what) -> world size, summing `tensor` in place over virtual ranks


def _sum_over_ranks(t, what):
    """In-place SUM of a device tensor over the data-parallel ranks (RCCL all-reduce, asynchronous to the host);
    returns the world size.  `_rank_sum_hook` replaces the collective in single-process tests."""
    if _rank_sum_hook is not None:
        return _rank_sum_hook(t, what)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
        return dist.get_world_size()
    return 1


class _DiceFn(Function):
    @staticmethod
    def forward(ctx, logits, labels, smooth, ignore_index, ignore_channel, sync):
        n, c, h, w = logits.shape
        npix = n * h * w
        stats = _stats_buf(2 * c, logits.device)
        st = _stream()
        _timed_call('resample_loss', (4.0 * c + 8.0) * npix, 'evk_dice_stats', logits.data_ptr(), labels.data_ptr(), npix, c, ignore_index, stats.data_ptr(), st)
        # loss.py:46-48: inter / z summed over the ranks before the ratio.  The collective is enqueued behind the
        # statistics kernel and the finishing kernel behind it: the host never waits for it.
        world = _sum_over_ranks(stats[:2 * c], 'dice_stats') if sync else 1
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        _C.call('evk_dice_finish', stats.data_ptr(), c, float(smooth), ignore_channel, loss.data_ptr(), st)
        ctx.save_for_backward(logits, labels, stats)
        ctx.cfg = (float(smooth), ignore_index, ignore_channel, world)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, labels, stats = ctx.saved_tensors
        smooth, ignore_index, ignore_channel, world = ctx.cfg
        n, c, h, w = logits.shape
        g = g.contiguous().float()
        if world > 1:
            # backward of torch.distributed.nn.all_reduce(SUM) is an all_reduce(SUM) of the upstream grads
            g = g.clone()
            _sum_over_ranks(g, 'dice_grad')
        d = torch.empty_like(logits)
        _timed_call('resample_loss', (8.0 * c + 8.0) * n * h * w, 'evk_dice_bwd', logits.data_ptr(), labels.data_ptr(), n * h * w, c, ignore_index, stats.data_ptr(),
                smooth, ignore_channel, g.data_ptr(), d.data_ptr(), 0, _stream())
        return d, None, None, None, None, None


def dice_loss_with_logits(y_pred, y_true, smooth_value=1.0, ignore_index=255, ignore_channel=-1,
                          sync_statistics=True):
    """reference ever/module/loss.py:40-75."""
    _require_cuda(y_pred, 'dice_loss_with_logits')
    if y_pred.dim() != 4 or y_true.dim() != 3:
        raise AssertionError('dice_loss_with_logits expects y_pred [N,C,H,W] and y_true [N,H,W]')
    y_pred = as_nhwc(y_pred, 'dice')
    labels = _labels(y_true, y_pred.numel() // y_pred.shape[1], 'dice')
    return _DiceFn.apply(y_pred, labels, smooth_value, int(ignore_index), int(ignore_channel), bool(sync_statistics))


class _CeFn(Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index, eps):
        n, c, h, w = logits.shape
        stats = _stats_buf(3, logits.device)
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        _C.call('evk_ce_fwd', logits.data_ptr(), labels.data_ptr(), n * h * w, c, ignore_index, eps, loss.data_ptr(),
                stats.data_ptr(), _stream())
        ctx.save_for_backward(logits, labels, stats)
        ctx.cfg = (ignore_index, eps)
        count = stats[1:2].to(torch.float32).reshape(())     # valid pixels (for reduction='sum'); not differentiable
        ctx.mark_non_differentiable(count)
        return loss, count

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _gcount=None):
        logits, labels, stats = ctx.saved_tensors
        ignore_index, eps = ctx.cfg
        n, c, h, w = logits.shape
        g = g.contiguous().float()
        d = torch.empty_like(logits)
        _C.call('evk_ce_bwd', logits.data_ptr(), labels.data_ptr(), n * h * w, c, ignore_index, eps, stats.data_ptr(),
                g.data_ptr(), d.data_ptr(), 0, _stream())
        return d, None, None, None


class _SoftCeFn(Function):
    @staticmethod
    def forward(ctx, logits, target):
        n, c, h, w = logits.shape
        stats = _stats_buf(1, logits.device)
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        _C.call('evk_soft_ce_fwd', logits.data_ptr(), target.data_ptr(), n * h * w, c, loss.data_ptr(),
                stats.data_ptr(), _stream())
        ctx.save_for_backward(logits, target)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        n, c, h, w = logits.shape
        g = g.contiguous().float()
        d = torch.empty_like(logits)
        _C.call('evk_soft_ce_bwd', logits.data_ptr(), target.data_ptr(), n * h * w, c, g.data_ptr(), d.data_ptr(),
                _stream())
        return d, None


def soft_cross_entropy(y_pred, target):
    """reference ever/module/loss.py:238-242 (target is a per-pixel distribution [N,C,H,W]; no gradient to it)."""
    _require_cuda(y_pred, 'soft_cross_entropy')
    assert y_pred.dim() == 4 and target.dim() == 4
    y_pred, target = as_nhwc(y_pred, 'soft_ce'), as_nhwc(target.detach(), 'soft_ce.target')
    return _SoftCeFn.apply(y_pred, target)


def cross_entropy(y_pred, y_true, ignore_index=255, label_smoothing=0.0, reduction='mean'):
    """F.cross_entropy(ignore_index=...) / label_smoothing_cross_entropy (reference loss.py:207-219).
    reduction 'sum' = the mean times the number of valid pixels, a device word of the same kernel (no host round trip; 0 when
    every pixel is ignored, as the reference's sums over nothing).  'none': per pixel, 0 on ignored pixels — the reference's
    own expression (a compacted 1-D term plus an uncompacted one, loss.py:213-219) is defined only where nothing is ignored,
    and equals this there; built from the per-pixel cross entropy: -sum_c q_c log p_c with q = (1 - eps) onehot + eps / C,
    the uniform part as the sum over the C constant-label cross entropies."""
    _require_cuda(y_pred, 'cross_entropy')
    if reduction not in ('mean', 'sum', 'none'):
        raise ValueError(f"cross_entropy: reduction '{reduction}'")
    squeeze = y_pred.dim() == 2          # flat [M, C] logits with [M] targets (the form the reference's 'none' is defined on)
    if squeeze:
        y_pred = y_pred.reshape(y_pred.shape[0], y_pred.shape[1], 1, 1)
    y_pred = as_nhwc(y_pred, 'cross_entropy')
    if reduction == 'none':
        from . import functional_next as HN
        n, c, h, w = y_pred.shape
        yt = y_true.to(torch.int64).reshape(n, h, w)
        out = HN.cross_entropy_per_pixel(y_pred, yt, ignore_index)
        if label_smoothing:
            valid = yt != ignore_index
            uni = None
            for k in range(c):
                t = HN.cross_entropy_per_pixel(y_pred, torch.where(valid, torch.full_like(yt, k), yt), ignore_index)
                uni = t if uni is None else uni + t
            out = out * (1.0 - label_smoothing) + uni * (label_smoothing / c)
        return out.reshape(y_true.shape)
    labels = _labels(y_true, y_pred.numel() // y_pred.shape[1], 'cross_entropy')
    mean, count = _CeFn.apply(y_pred, labels, int(ignore_index), float(label_smoothing))
    if reduction == 'mean':
        return mean
    return torch.where(count > 0, mean * count, torch.zeros_like(mean))


# ------------------------------------------------------------------ SURVEY §8 f2 / f3 rows
class _ProbStatsFn(Function):
    """(tp, sum_p, sum_y) per class over the valid pixels as a float32 [3, C] tensor; backward is the adjoint
    kernel, so any differentiable function of the statistics (tversky, dice variants) trains through it."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        n, c, h, w = logits.shape
        lib = _C.load()
        stats = torch.empty((lib.evk_prob_stats_doubles(c),), device=logits.device, dtype=torch.float64)
        _C.call('evk_prob_stats', logits.data_ptr(), labels.data_ptr(), n * h * w, c, ignore_index, stats.data_ptr(),
                _stream())
        ctx.save_for_backward(logits, labels)
        ctx.ignore_index = ignore_index
        return stats[:3 * c].reshape(3, c)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, labels = ctx.saved_tensors
        n, c, h, w = logits.shape
        g = g.float().contiguous()
        d = torch.empty_like(logits)
        _C.call('evk_prob_stats_bwd', logits.data_ptr(), labels.data_ptr(), n * h * w, c, ctx.ignore_index,
                g[0].data_ptr(), g[1].data_ptr(), d.data_ptr(), 0, _stream())
        return d, None, None


def prob_stats(y_pred, y_true, ignore_index=255):
    _require_cuda(y_pred, 'prob_stats')
    y_pred = as_nhwc(y_pred, 'prob_stats')
    labels = _labels(y_true, y_pred.numel() // y_pred.shape[1], 'prob_stats')
    return _ProbStatsFn.apply(y_pred, labels, int(ignore_index))


def _all_reduce_sum_differentiable(t):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        import torch.distributed.nn as dist_nn
        return dist_nn.all_reduce(t)
    return t


def tversky_loss_with_logits(y_pred, y_true, alpha, beta=None, gamma=1.0, smooth_value=1.0, ignore_index=255,
                             reduction='mean', sync_statistics=True):
    """reference ever/module/loss.py:78-143: statistics by the HIP kernel, the C-element ratio by autograd."""
    st = prob_stats(y_pred, y_true, ignore_index)  # float64 [3, C]
    tp, sp, sy = st[0], st[1], st[2]
    if isinstance(alpha, (list, tuple)):
        alpha = torch.as_tensor(alpha, dtype=st.dtype, device=st.device)
    if beta is None:
        beta = 1. - alpha
    fp, fn = sp - tp, sy - tp
    num, den = tp, tp + alpha * fn + beta * fp
    if sync_statistics:
        num, den = _all_reduce_sum_differentiable(num), _all_reduce_sum_differentiable(den)
    coeff = (num + smooth_value) / (den + smooth_value)
    loss = ((1. - coeff) ** gamma).float()
    if reduction == 'mean':
        return loss.mean()
    if reduction == 'none':
        return loss
    raise ValueError(f'unknown reduction: {reduction}')


class _FocalFn(Function):
    @staticmethod
    def forward(ctx, logits, target, gamma, alpha, mode, mean):
        n = logits.numel()
        stats = torch.empty((1 + 256,), device=logits.device, dtype=torch.float64)
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        _C.call('evk_focal_fwd', logits.data_ptr(), target.data_ptr(), n, gamma, alpha, mode, mean, loss.data_ptr(),
                stats.data_ptr(), _stream())
        ctx.save_for_backward(logits, target)
        ctx.cfg = (gamma, alpha, mode, mean)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        gamma, alpha, mode, mean = ctx.cfg
        g = g.contiguous().float()
        d = torch.empty_like(logits)
        _C.call('evk_focal_bwd', logits.data_ptr(), target.data_ptr(), logits.numel(), gamma, alpha, mode, mean,
                g.data_ptr(), d.data_ptr(), _stream())
        return d, None, None, None, None, None


def _focal(y_pred, y_true, gamma, alpha, mode, mean, what):
    _require_cuda(y_pred, what)
    if y_pred.shape != y_true.shape:
        raise ValueError(f'{what}: logits {tuple(y_pred.shape)} and targets {tuple(y_true.shape)} must have the same shape')
    yp = y_pred if y_pred.is_contiguous() or (y_pred.dim() == 4 and is_nhwc(y_pred)) else y_pred.contiguous()
    # element-wise: any common dense layout works as long as both operands share it
    yt = y_true.detach().float()
    if yt.stride() != yp.stride():
        yt = torch.empty_like(yp).copy_(yt)
    return _FocalFn.apply(yp, yt, float(gamma), float(alpha), int(mode), int(mean))


def focal_loss(y_pred, y_true, gamma=2.0, normalize=False):
    """reference loss.py:158-176"""
    return _focal(y_pred, y_true, gamma, -1.0, 2 if normalize else 0, 0 if normalize else 1, 'focal_loss')


def sigmoid_focal_loss(y_pred, y_true, alpha=-1, gamma=2, reduction='mean'):
    """reference loss.py:179-201 (fvcore form)"""
    if reduction not in ('mean', 'sum'):
        raise NotImplementedError("sigmoid_focal_loss: reduction must be 'mean' or 'sum' on the HIP path")
    return _focal(y_pred, y_true, gamma, alpha, 1, 1 if reduction == 'mean' else 0, 'sigmoid_focal_loss')


def confusion_matrix_update(cm, y_true, y_pred=None, logits=None):
    """cm (int64 [C, C], cuda) += counts.  Either integer predictions or NCHW logits (threshold / argmax fused)."""
    c = cm.shape[0]
    assert cm.is_cuda and cm.dtype == torch.int64 and cm.is_contiguous() and cm.shape == (c, c)
    yt = y_true.to(device=cm.device, dtype=torch.int64).contiguous()
    if logits is not None:
        _require_cuda(logits, 'confusion_matrix')
        lg = as_nhwc(logits.detach(), 'confusion_matrix')
        cl = lg.shape[1]
        if yt.numel() != lg.numel() // cl:
            raise ValueError('confusion_matrix: label / logit pixel counts differ')
        _C.call('evk_confusion_from_logits', lg.data_ptr(), yt.data_ptr(), yt.numel(), cl, c, cm.data_ptr(), _stream())
    else:
        yp = y_pred.to(device=cm.device, dtype=torch.int64).contiguous()
        if yp.numel() != yt.numel():
            raise ValueError('confusion_matrix: y_true and y_pred sizes differ')
        _C.call('evk_confusion_matrix', yt.data_ptr(), yp.data_ptr(), yt.numel(), c, cm.data_ptr(), _stream())
    return cm
