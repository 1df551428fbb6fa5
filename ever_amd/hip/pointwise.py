"""Streaming layers of the host wrappers: ReLU, add, pools, nearest / bilinear resampling, global average pool, FS-Relation
(reference fs_relation.py:56-73), BatchNorm + ReLU + classifier dot (fpn.py:163-193), the decoder's 4-way mean.  Part of
the hip/functional.py facade."""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from . import timing, weight_planes
from .workspace import workspace
from ._base import (  # noqa: F401
    HipPathError, _AMAX_HANDOFF, _amax_out, _amax_zeroed, _conv_out, _f16x2, _inherit_amax, _mark_packed, _note_amax, _ptr,
    _require_cuda, _same_shape_fake, _stream, _timed_call, as_nhwc, empty_nhwc,
)
from .streams import (  # noqa: F401
    _note_param_use,
)
from .conv import (  # noqa: F401
    _weight_ohwi,
)


# ------------------------------------------------------------------------------------ pointwise
class _ReluFn(Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        _C.call('evk_relu_fwd', x.data_ptr(), y.data_ptr(), x.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = as_nhwc(dy, 'relu.backward') if dy.dim() == 4 else dy.contiguous()
        dx = torch.empty_like(y)
        _C.call('evk_relu_bwd', dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), _stream())
        return dx


def relu(x):
    """nn.ReLU (reference fs_relation.py:25)."""
    _require_cuda(x, 'relu')
    if x.dim() == 4:
        x = as_nhwc(x, 'relu')
    return _ReluFn.apply(x)


class _AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        _C.call('evk_add', a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream())
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g, g


def add(a, b):
    _require_cuda(a, 'add')
    a, b = as_nhwc(a, 'add'), as_nhwc(b, 'add')
    if a.shape != b.shape:
        raise ValueError(f'add: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}')
    return _AddFn.apply(a, b)


class _MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = empty_nhwc(n, c, ho, wo, x.device)
        code = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8)
        _C.call('evk_maxpool3x3s2_fwd', x.data_ptr(), y.data_ptr(), code.data_ptr(), n, h, w, c, _stream())
        ctx.save_for_backward(code)
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (code,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        dy = as_nhwc(dy, 'max_pool.backward')
        dx = empty_nhwc(n, c, h, w, dy.device)
        _C.call('evk_maxpool3x3s2_bwd', dy.data_ptr(), code.data_ptr(), dx.data_ptr(), n, h, w, c, _stream())
        return dx


def max_pool3x3s2(x):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (reference _resnets.py:153)."""
    _require_cuda(x, 'max_pool')
    x = as_nhwc(x, 'max_pool')
    if x.shape[1] % 4:
        raise HipPathError('max_pool: channels must be a multiple of 4')
    return _inherit_amax(_MaxPoolFn.apply(x), x)      # a selection of x's elements


class _Nearest2xAddFn(Function):
    @staticmethod
    def forward(ctx, top, lateral):
        n, c, h, w = lateral.shape
        out = empty_nhwc(n, c, h, w, lateral.device)
        bits = _amax_zeroed(lateral.device)      # the sum is the FPN output convolution's operand
        _C.call('evk_upsample_nearest2x_add_fwd', top.data_ptr(), lateral.data_ptr(), out.data_ptr(), n, h, w, c,
                _ptr(bits), _stream())
        if bits is not None:
            _note_amax(out, bits)
        ctx.shape = (n, c, h, w)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        g = as_nhwc(g, 'nearest2x.backward')
        dtop = None
        if ctx.needs_input_grad[0]:
            dtop = empty_nhwc(n, c, h // 2, w // 2, g.device)
            _C.call('evk_upsample_nearest2x_bwd', g.data_ptr(), dtop.data_ptr(), n, h, w, c, _stream())
        return dtop, (g if ctx.needs_input_grad[1] else None)


def upsample_nearest2x_add(top, lateral):
    """`inner_lateral + F.interpolate(last_inner, scale_factor=2, mode="nearest")` (reference fpn.py:100-105)."""
    _require_cuda(top, 'upsample_nearest2x_add')
    top, lateral = as_nhwc(top, 'fpn.top'), as_nhwc(lateral, 'fpn.lateral')
    n, c, h, w = lateral.shape
    if top.shape != (n, c, h // 2, w // 2) or h % 2 or w % 2:
        # same failure mode as the reference's `inner_lateral + inner_top_down` (fpn.py:105)
        raise RuntimeError(f'The size of tensor a ({tuple(lateral.shape)}) must match the size of tensor b '
                           f'(nearest x2 of {tuple(top.shape)})')
    return _Nearest2xAddFn.apply(top, lateral)


class _Subsample2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, x.device)
        _C.call('evk_subsample2_fwd', x.data_ptr(), y.data_ptr(), n, h, w, c, _stream())
        _inherit_amax(y, x)                       # a selection of x's elements: its scale bounds them
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        g = as_nhwc(g, 'subsample2.backward')
        dx = empty_nhwc(n, c, h, w, g.device)
        _C.call('evk_subsample2_bwd', g.data_ptr(), dx.data_ptr(), n, h, w, c, _stream())
        return dx


def max_pool1x1s2(x):
    """F.max_pool2d(x, 1, 2, 0) (reference fpn.py:118-120, LastLevelMaxPool): every second pixel of every second row."""
    _require_cuda(x, 'max_pool1x1s2')
    x = as_nhwc(x, 'max_pool1x1s2')
    if x.shape[1] % 4:
        raise HipPathError('max_pool1x1s2: channels must be a multiple of 4')
    return _Subsample2Fn.apply(x)


class _BilinearFn(Function):
    @staticmethod
    def forward(ctx, x, ho, wo):
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, ho, wo, x.device)
        _timed_call('resample_loss', 4.0 * (x.numel() + y.numel()), 'evk_upsample_bilinear_fwd', x.data_ptr(),
                    y.data_ptr(), n, h, w, ho, wo, c, _stream())
        ctx.dims = (n, c, h, w, ho, wo)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, c, h, w, ho, wo = ctx.dims
        dy = as_nhwc(dy, 'bilinear.backward')
        dx = empty_nhwc(n, c, h, w, dy.device)
        _timed_call('resample_loss', 4.0 * (dy.numel() + dx.numel()), 'evk_upsample_bilinear_bwd', dy.data_ptr(),
                    dx.data_ptr(), n, h, w, ho, wo, c, _stream())
        return dx, None, None


def upsample_bilinear(x, scale_factor):
    """nn.UpsamplingBilinear2d(scale_factor) == bilinear with align_corners=True (reference fpn.py:168,180)."""
    _require_cuda(x, 'upsample_bilinear')
    x = as_nhwc(x, 'upsample_bilinear')
    sh, sw = (scale_factor, scale_factor) if not isinstance(scale_factor, (tuple, list)) else scale_factor
    ho, wo = int(x.shape[2] * sh), int(x.shape[3] * sw)  # floor(in * scale), as aten
    return _inherit_amax(_BilinearFn.apply(x, ho, wo), x)   # convex combinations of x's elements


class _GapFn(Function):
    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, 1, 1, x.device)
        _C.call('evk_gap_fwd', x.data_ptr(), y.data_ptr(), n, h * w, c, _stream())
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        dy = dy.reshape(n, c).contiguous()
        dx = empty_nhwc(n, c, h, w, dy.device)
        _C.call('evk_gap_bwd', dy.data_ptr(), dx.data_ptr(), n, h * w, c, _stream())
        return dx


def global_avg_pool(x):
    """F.adaptive_avg_pool2d(x, 1) (reference fs_relation.py:177)."""
    _require_cuda(x, 'global_avg_pool')
    x = as_nhwc(x, 'global_avg_pool')
    return _GapFn.apply(x)


class _RelationFn(Function):
    @staticmethod
    def forward(ctx, scene, content, feat):
        n, c, h, w = content.shape
        out = empty_nhwc(n, c, h, w, content.device)
        r = torch.empty((n, h * w), device=content.device, dtype=torch.float32)
        _C.call('evk_relation_fwd', scene.data_ptr(), content.data_ptr(), feat.data_ptr(), out.data_ptr(),
                r.data_ptr(), n, h * w, c, _stream())
        ctx.save_for_backward(scene, content, feat, r)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        scene, content, feat, r = ctx.saved_tensors
        n, c, h, w = content.shape
        dev = content.device
        dout = as_nhwc(dout, 'fs_relation.backward')
        lib = _C.load()
        ws_bytes = lib.evk_relation_workspace_bytes(n, h * w, c)
        ws = workspace(dev, ws_bytes)
        dscene = empty_nhwc(n, c, 1, 1, dev)
        dcontent = torch.empty_like(content)
        dfeat = torch.empty_like(feat)
        _C.call('evk_relation_bwd', dout.data_ptr(), scene.data_ptr(), content.data_ptr(), feat.data_ptr(),
                r.data_ptr(), dscene.data_ptr(), dcontent.data_ptr(), dfeat.data_ptr(), n, h * w, c, ws.data_ptr(),
                ws_bytes, _stream())
        return dscene, dcontent, dfeat


def fs_relation(scene, content, feat):
    """`sigmoid((scene * content).sum(dim=1, keepdim=True)) * feat` (reference fs_relation.py:61-71)."""
    _require_cuda(content, 'fs_relation')
    content, feat = as_nhwc(content, 'fs_relation.content'), as_nhwc(feat, 'fs_relation.feat')
    n, c, h, w = content.shape
    scene = as_nhwc(scene.reshape(n, c, 1, 1), 'fs_relation.scene')
    return _inherit_amax(_RelationFn.apply(scene, content, feat), feat)   # sigmoid(.) * feat


class _RelationBnFn(Function):
    """FS-Relation on the two convolution outputs directly: BatchNorm (batch statistics from the convolutions' epilogue
    records) + ReLU of both branches happen inside the relation kernels (include/ever_hip.h: evk_relation_bn_*), their
    backward sums come out of the relation backward.  Replaces content_encoder[1:], feature_reencoder[1:] and the relation
    of reference fs_relation.py:39-53,61-71 as one node."""

    @staticmethod
    def forward(ctx, scene, zc, zf, wc, bc, wf, bf, rmc, rvc, rmf, rvf, cfg):
        (parts_c, parts_f, mom_c, eps_c, mom_f, eps_f) = cfg
        n, c, h, w = zc.shape
        rows, dev, st = n * h * w, zc.device, _stream()
        stats = torch.empty((2, 4, c), device=dev, dtype=torch.float32)   # per BatchNorm: mean, invstd, scale, shift
        for k, (parts, g, b, rm, rv, mom, eps) in enumerate(((parts_c, wc, bc, rmc, rvc, mom_c, eps_c),
                                                               (parts_f, wf, bf, rmf, rvf, mom_f, eps_f))):
            _C.call('evk_bn_finalize_parts', parts[0].data_ptr(), parts[1], c, rows, _ptr(g), _ptr(b), _ptr(rm), _ptr(rv),
                    float(mom), float(eps), stats[k, 0].data_ptr(), stats[k, 1].data_ptr(), stats[k, 2].data_ptr(), st)
        out = empty_nhwc(n, c, h, w, dev)
        r = torch.empty((n, h * w), device=dev, dtype=torch.float32)
        abits = _amax_zeroed(dev)
        _C.call('evk_relation_bn_fwd', scene.data_ptr(), zc.data_ptr(), stats[0, 2].data_ptr(), zf.data_ptr(),
                stats[1, 2].data_ptr(), out.data_ptr(), r.data_ptr(), n, h * w, c, _ptr(abits), st)
        _AMAX_HANDOFF[0] = (abits, False)
        ctx.pack = (bool(len(parts_c) > 2 and parts_c[2]), bool(len(parts_f) > 2 and parts_f[2]))
        ctx.save_for_backward(scene, zc, zf, wc, wf, stats, r)
        ctx.mark_non_differentiable(*[t for t in (rmc, rvc, rmf, rvf) if t is not None])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        scene, zc, zf, wc, wf, stats, r = ctx.saved_tensors
        n, c, h, w = zc.shape
        rows, dev, st = n * h * w, zc.device, _stream()
        dout = as_nhwc(dout, 'fs_relation.backward')
        lib = _C.load()
        nb = int(lib.evk_relation_bn_parts(n, h * w))
        ws_bytes = lib.evk_relation_bn_workspace_bytes(n, h * w, c)
        # (own buffer, not the shared workspace: the BatchNorm backward launches below read it while they use that one)
        ws = torch.empty((ws_bytes // 4,), device=dev, dtype=torch.float32)
        dscene = empty_nhwc(n, c, 1, 1, dev)
        gc, gf = torch.empty_like(zc), torch.empty_like(zf)
        _C.call('evk_relation_bn_bwd', dout.data_ptr(), scene.data_ptr(), zc.data_ptr(), stats[0, 2].data_ptr(),
                stats[0, 0].data_ptr(), zf.data_ptr(), stats[1, 2].data_ptr(), stats[1, 0].data_ptr(), r.data_ptr(),
                dscene.data_ptr(), gc.data_ptr(), gf.data_ptr(), n, h * w, c, ws.data_ptr(), ws_bytes, st)
        coef = workspace(dev, 16 * c * 4)
        grads = []
        for k, (g, z, gamma) in enumerate(((gc, zc, wc), (gf, zf, wf))):
            sums = ws[nb * c * (1 + 4 * k):]
            maxima = ws[nb * c * (3 + 4 * k):]
            pack = ctx.pack[k] and _f16x2()
            abits = _amax_zeroed(dev) if pack else _amax_out(dev)
            pack = pack and abits is not None
            dz = torch.empty_like(z)
            dgamma = torch.empty((c,), device=dev, dtype=torch.float32) if gamma is not None else None
            dbeta = torch.empty((c,), device=dev, dtype=torch.float32) if gamma is not None else None
            # algorithmic bytes: read g, z, write dz
            _timed_call('bn', 12.0 * z.numel(), 'evk_bn_bwd_from_partials', g.data_ptr(), z.data_ptr(), _ptr(gamma),
                        stats[k, 0].data_ptr(), stats[k, 1].data_ptr(), sums.data_ptr(), maxima.data_ptr(), nb, dz.data_ptr(),
                        _ptr(dgamma), _ptr(dbeta), rows, c, 2 if pack else 0, 1, coef.data_ptr(), 16 * c * 4, _ptr(abits), st)
            if pack:
                _mark_packed(dz, abits)
            elif abits is not None:
                _note_amax(dz, abits)
            grads.append((dz, dgamma, dbeta))
        (dzc, dgc, dbc), (dzf, dgf, dbf) = grads
        return dscene, dzc, dzf, dgc, dbc, dgf, dbf, None, None, None, None, None


def fs_relation_bn(scene, zc, zf, bn_c, bn_f):
    """`sigmoid(<scene, relu(bn_c(zc))>) * relu(bn_f(zf))` with both training-mode BatchNorms inside the relation kernels
    (see _RelationBnFn); zc, zf carry their convolutions' statistics records (`_evk_bn_parts`), or None is returned and the
    caller runs the layers one by one."""
    pc, pf = getattr(zc, '_evk_bn_parts', None), getattr(zf, '_evk_bn_parts', None)
    if pc is None or pf is None or pc[1] <= 0 or pf[1] <= 0 or zc.shape != zf.shape or zc.shape[1] % 4 or zc.shape[1] > 448:
        return None
    del zc._evk_bn_parts, zf._evk_bn_parts
    n, c, h, w = zc.shape
    scene = as_nhwc(scene.reshape(n, c, 1, 1), 'fs_relation.scene')
    weight_planes.note_running_stats_changed()
    _AMAX_HANDOFF[0] = None

    def stat(bn, name):
        return getattr(bn, name) if bn.track_running_stats else None
    out = _RelationBnFn.apply(scene, zc, zf, bn_c.weight, bn_c.bias, bn_f.weight, bn_f.bias, stat(bn_c, 'running_mean'),
                              stat(bn_c, 'running_var'), stat(bn_f, 'running_mean'), stat(bn_f, 'running_var'),
                              (pc, pf, bn_c.momentum, bn_c.eps, bn_f.momentum, bn_f.eps))
    if _AMAX_HANDOFF[0] is not None:
        abits, _ = _AMAX_HANDOFF[0]
        if abits is not None:
            _note_amax(out, abits)
        _AMAX_HANDOFF[0] = None
    return out


class _BnReluDotFn(Function):
    """out = conv1x1(relu(bn(z))) for a narrow classifier (K <= 16), BatchNorm with batch statistics from the producing
    convolution's epilogue records, as one consumer of z (include/ever_hip.h: evk_bn_relu_dot_*): the normalised map is
    never written, its K-fold outer-product gradient never formed.  Replaces blocks[i][-1][1:3] + classifier[0] of reference
    fpn.py:163-170,179-193 in the commuted decoder (module/fpn.py)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, w, bias, rm, rv, cfg):
        parts, mom, eps = cfg
        n, c, h, wd = z.shape
        k = w.shape[0]
        rows, dev, st = n * h * wd, z.device, _stream()
        stats = torch.empty((4, c), device=dev, dtype=torch.float32)      # mean, invstd, scale, shift
        _C.call('evk_bn_finalize_parts', parts[0].data_ptr(), parts[1], c, rows, _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv),
                float(mom), float(eps), stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), st)
        w2 = _weight_ohwi(w.detach()).reshape(k, c)
        out = empty_nhwc(n, k, h, wd, dev)
        # algorithmic bytes: read z (the K-channel result is noise beside it)
        _timed_call('bn', 4.0 * z.numel(), 'evk_bn_relu_dot_fwd', z.data_ptr(), stats[2].data_ptr(), w2.data_ptr(), _ptr(bias),
                    out.data_ptr(), rows, c, k, st)
        ctx.pack = bool(len(parts) > 2 and parts[2])
        ctx.has_bias = bias is not None
        ctx.save_for_backward(z, gamma, w, stats)
        ctx.mark_non_differentiable(*[t for t in (rm, rv) if t is not None])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dl):
        z, gamma, w, stats = ctx.saved_tensors
        n, c, h, wd = z.shape
        k = w.shape[0]
        rows, dev, st = n * h * wd, z.device, _stream()
        dl = as_nhwc(dl, 'bn_relu_dot.backward')
        lib = _C.load()
        ws_bytes = lib.evk_bn_relu_dot_workspace_bytes(rows, c, k)
        ws = workspace(dev, ws_bytes)
        pack = ctx.pack and _f16x2()
        abits = _amax_zeroed(dev) if pack else _amax_out(dev)
        pack = pack and abits is not None
        dz = torch.empty_like(z)
        dgamma = torch.empty((c,), device=dev, dtype=torch.float32) if gamma is not None else None
        dbeta = torch.empty((c,), device=dev, dtype=torch.float32) if gamma is not None else None
        # (OHWI memory = [k][c] for a 1x1 kernel, presented with exactly the parameter's strides: the size-1 dims make the
        # stride tuple ambiguous, and a gradient in "another layout" costs AccumulateGrad / the bucket pack a copy)
        dw = torch.empty_strided(w.shape, w.stride(), device=dev, dtype=torch.float32) if (
            w.stride(0) == c and w.stride(1) == 1) else torch.empty_like(w, memory_format=torch.channels_last)
        dbias = torch.empty((k,), device=dev, dtype=torch.float32) if ctx.has_bias else None
        w2 = _weight_ohwi(w.detach()).reshape(k, c)
        # algorithmic bytes: read z twice, write dz
        _timed_call('bn', 12.0 * z.numel(), 'evk_bn_relu_dot_bwd', dl.data_ptr(), z.data_ptr(), stats[2].data_ptr(), _ptr(gamma),
                    stats[0].data_ptr(), stats[1].data_ptr(), w2.data_ptr(), dz.data_ptr(), _ptr(dgamma), _ptr(dbeta),
                    dw.data_ptr(), _ptr(dbias), rows, c, k, 2 if pack else 0, ws.data_ptr(), ws_bytes, _ptr(abits), st)
        if pack:
            _mark_packed(dz, abits)
        elif abits is not None:
            _note_amax(dz, abits)
        return dz, dgamma, dbeta, dw, dbias, None, None, None


def bn_relu_dot(z, bn, conv):
    """`conv(relu(bn(z)))` for a training-mode BatchNorm2d whose statistics records ride on z (`_evk_bn_parts`) and a 1x1
    convolution with at most 16 outputs, as one pass each way (see _BnReluDotFn); None when that form does not apply."""
    parts = getattr(z, '_evk_bn_parts', None)
    k, c = conv.weight.shape[0], z.shape[1]
    if (parts is None or parts[1] <= 0 or k > 16 or c % 4 or c > 1024 or tuple(conv.weight.shape[2:]) != (1, 1)
            or conv.weight.shape[1] != c or (c > 256 and k > 4) or 16 * (4 + k) * c > 65536):
        return None      # (the kernel's register / LDS budget: csrc/bn.hip evk_bn_relu_dot_bwd)
    del z._evk_bn_parts
    weight_planes.note_running_stats_changed()
    if torch.is_grad_enabled():
        _note_param_use(conv.weight, conv.bias)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BnReluDotFn.apply(z, bn.weight, bn.bias, conv.weight, conv.bias, rm, rv, (parts, bn.momentum, bn.eps))


class _Mean4Fn(Function):
    @staticmethod
    def forward(ctx, a, b, c, d):
        out = torch.empty_like(a)
        bits = _amax_zeroed(a.device)            # the mean is the classifier convolution's operand
        _C.call('evk_mean4_fwd', a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), out.data_ptr(), a.numel(),
                _ptr(bits), _stream())
        if bits is not None:
            _note_amax(out, bits)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = as_nhwc(g, 'mean4.backward')
        q = torch.empty_like(g)
        _C.call('evk_scale', g.data_ptr(), 0.25, q.data_ptr(), g.numel(), _stream())
        return q, q, q, q


def mean4(a, b, c, d):
    """`sum(inner_feat_list) / len(inner_feat_list)` for the 4 decoder branches (reference fpn.py:189)."""
    ts = [as_nhwc(t, 'mean4') for t in (a, b, c, d)]
    for t in ts[1:]:
        if t.shape != ts[0].shape:
            raise ValueError('mean4: shape mismatch')
    return _Mean4Fn.apply(*ts)


# ---------------------------------------------------------------------------------------------------------------------
# The no-grad forward of this family's layers as dispatcher-level operators (hip/oplib.py): what `torch.jit.trace`
# (reference api/infer_tool.py:70-74: export_model) and a compiler's shape pass record instead of an opaque Python call.
# Eager calls keep the direct path; the names below are what the modules (and this file) call from here on.
from . import oplib as _oplib  # noqa: E402


_relu_plain, _add_plain, _max_pool_plain, _nearest_add_plain = relu, add, max_pool3x3s2, upsample_nearest2x_add
_bilinear_plain, _gap_plain, _relation_plain, _mean4_plain = upsample_bilinear, global_avg_pool, fs_relation, mean4
relu = _oplib.traceable('relu', '(Tensor x) -> Tensor', _relu_plain, adapt=lambda x: (x,), fake=_same_shape_fake)
add = _oplib.traceable('add', '(Tensor a, Tensor b) -> Tensor', _add_plain, adapt=lambda a, b: (a, b), fake=_same_shape_fake)
max_pool3x3s2 = _oplib.traceable(
    'max_pool3x3s2', '(Tensor x) -> Tensor', _max_pool_plain, adapt=lambda x: (x,),
    fake=lambda x: _oplib.nhwc_like(x, x.shape[0], x.shape[1], _conv_out(x.shape[2], 3, 2, 1, 1), _conv_out(x.shape[3], 3, 2, 1, 1)))
upsample_nearest2x_add = _oplib.traceable(
    'upsample_nearest2x_add', '(Tensor top, Tensor lateral) -> Tensor', _nearest_add_plain, adapt=lambda top, lateral: (top, lateral),
    fake=lambda top, lateral: _oplib.nhwc_like(lateral, *lateral.shape))
upsample_bilinear = _oplib.traceable(
    'upsample_bilinear', '(Tensor x, float scale_h, float scale_w) -> Tensor',
    _bilinear_plain, impl_fn=lambda x, sh, sw: _bilinear_plain(x, (sh, sw)),
    adapt=lambda x, scale_factor: (x,) + tuple(float(s) for s in ((scale_factor, scale_factor) if not isinstance(
        scale_factor, (tuple, list)) else scale_factor)),
    fake=lambda x, sh, sw: _oplib.nhwc_like(x, x.shape[0], x.shape[1], int(x.shape[2] * sh), int(x.shape[3] * sw)))
global_avg_pool = _oplib.traceable('global_avg_pool', '(Tensor x) -> Tensor', _gap_plain, adapt=lambda x: (x,),
                                   fake=lambda x: _oplib.nhwc_like(x, x.shape[0], x.shape[1], 1, 1))
fs_relation = _oplib.traceable('fs_relation', '(Tensor scene, Tensor content, Tensor feat) -> Tensor', _relation_plain,
                               adapt=lambda scene, content, feat: (scene, content, feat),
                               fake=lambda scene, content, feat: _oplib.nhwc_like(feat, *feat.shape))
mean4 = _oplib.traceable('mean4', '(Tensor a, Tensor b, Tensor c, Tensor d) -> Tensor', _mean4_plain,
                         adapt=lambda a, b, c, d: (a, b, c, d), fake=lambda a, b, c, d: _oplib.nhwc_like(a, *a.shape))
