"""Inference-time folding of BatchNorm into the preceding convolution.

In eval mode BatchNorm is the per-channel affine map y = x * s + t with s = gamma / sqrt(running_var + eps),
t = beta - running_mean * s, so conv -> BN (-> + identity) (-> ReLU) collapses into ONE convolution with weight
w * s[:, None, None, None], bias (b_conv * s + t), and the residual add / ReLU in the implicit-GEMM epilogue
(evk_conv2d_fwd_res / _x3_res): the BatchNorm pass (2|x| of HBM traffic per layer) disappears, and the split
weight planes of the bf16x3 kernels are computed once instead of per call.  `fold_batchnorm(model)` prepares the
folded parameters next to the original ones (training is unaffected); the call sites (`conv_bn` in the ResNet
blocks and the Sequential peephole) use them whenever the BatchNorm is in eval mode."""
import ctypes

import collections
import weakref

import torch

from .. import _C
from ..hip import functional as HF

__all__ = ['fold_batchnorm', 'unfold_batchnorm', 'conv_bn']


class _Folded(object):
    __slots__ = ('weight', 'bias', 'planes', 'planes_key', 'cin', 'cin_p', 'stamp', 'wbits', '__weakref__')


def _stamp(conv, bn):
    """What the folded copy was derived from: autograd's version counters of the convolution / BatchNorm parameters
    and running statistics, and the plane-cache epoch (moved by raw-pointer optimiser kernels and by every
    ERModule.apply_gradients).  A folded pair whose stamp no longer matches is re-derived before use."""
    from ..hip import weight_planes
    ts = (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    return tuple(-1 if t is None else (t.data_ptr(), t._version) for t in ts) + (weight_planes._epoch, weight_planes._stats_epoch)


def _fold_pair(conv, bn):
    if not isinstance(bn, torch.nn.BatchNorm2d) or bn.running_mean is None:
        return None
    if conv.groups != 1:      # grouped convolutions (ResNeXt) keep the conv + BatchNorm pair at inference
        return None
    with torch.no_grad():
        dev = conv.weight.device
        s = (bn.weight if bn.affine else torch.ones_like(bn.running_mean)) * torch.rsqrt(bn.running_var + bn.eps)
        t = (bn.bias if bn.affine else torch.zeros_like(bn.running_mean)) - bn.running_mean * s
        w = HF._weight_ohwi(conv.weight.detach())            # logical OIHW over OHWI memory
        wf = (w * s.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
        b = t if conv.bias is None else conv.bias.detach() * s + t
        f = _Folded()
        cout, cin, kh, kw = wf.shape
        f.cin, f.cin_p = cin, HF._pad4(cin)
        if f.cin_p != cin:                                    # 3-band stem: pad the K axis once, here
            f.weight = HF._pad_last(wf.data_ptr(), cout * kh * kw, cin, f.cin_p, dev).reshape(cout, kh, kw, f.cin_p)
        else:
            f.weight = wf
        f.bias = b.contiguous().float()
        f.planes = None
        f.planes_key = None
        f.stamp = _stamp(conv, bn)
        weakref.finalize(f, _drop_planes_of, f.weight)   # a deleted model releases its planes
    return f


def fold_batchnorm(model):
    """Attach folded parameters to every Conv2d that is directly followed by a BatchNorm2d at a known call site
    (ResNet stem / blocks / down-sampling branches, conv-BN neighbours inside Sequential containers) and switch the
    model to eval mode.  Call again after loading new weights."""
    from . import _resnets
    from .layers import Conv2d
    model.eval()
    n = 0

    def pair(conv, bn):
        nonlocal n
        if isinstance(conv, Conv2d) and isinstance(bn, torch.nn.BatchNorm2d) and not isinstance(bn, torch.nn.SyncBatchNorm):
            f = _fold_pair(conv, bn)
            if f is not None:
                if getattr(conv, '_folded', None) is not None:
                    _drop_planes_of(conv._folded.weight)
                conv._folded, bn._folded_into = f, conv
                n += 1
    for m in model.modules():
        if isinstance(m, _resnets.ResNet) and not m.deep_stem:
            pair(m.conv1, m.bn1)
        if isinstance(m, (_resnets.BasicBlock, _resnets.Bottleneck)):
            pair(m.conv1, m.bn1)
            pair(m.conv2, m.bn2)
            if isinstance(m, _resnets.Bottleneck):
                pair(m.conv3, m.bn3)
            if m.downsample is not None:
                pair(m.downsample[0], m.downsample[1])
        if isinstance(m, torch.nn.Sequential):
            kids = list(m)
            for a, b in zip(kids, kids[1:]):
                if getattr(a, '_folded', None) is None:
                    pair(a, b)
    model._folded_pairs = n
    return model


def unfold_batchnorm(model):
    for m in model.modules():
        if hasattr(m, '_folded'):
            _drop_planes_of(getattr(m._folded, 'weight', None))
            del m._folded
        if hasattr(m, '_folded_into'):
            del m._folded_into
    return model


def _use_folded(conv, bn):
    if bn.training or torch.is_grad_enabled() or getattr(conv, '_folded', None) is None \
            or getattr(bn, '_folded_into', None) is not conv:
        return False
    if conv._folded.stamp != _stamp(conv, bn):
        # trained / fine-tuned / re-loaded since fold_batchnorm(): fold again from the current parameters
        f = _fold_pair(conv, bn)
        if f is None:
            return False
        _drop_planes_of(conv._folded.weight)
        conv._folded = f
    return True


def _takes_epilogue_stats(bn):
    """training-mode BatchNorm2d of this package (not SyncBatchNorm: its statistics are exchanged across ranks first)"""
    from .layers import BatchNorm2d
    return type(bn) is BatchNorm2d and (bn.training or bn.running_mean is None)


def conv_bn(conv, bn, x, residual=None, relu=False, conv_only=False, lazy_res=False):
    """bn(conv(x)) (+ residual) (ReLU): one folded convolution at inference, conv + fused BatchNorm pass otherwise.
    conv_only, lazy_res: see layers.BatchNorm2d.forward."""
    if _use_folded(conv, bn):
        return folded_conv2d(x, conv, residual=residual, relu=relu)
    if lazy_res:
        return bn(conv(x, bn_stats=_takes_epilogue_stats(bn)), residual=residual, relu=relu, conv_only=conv_only, lazy_res=True)
    return bn(conv(x, bn_stats=_takes_epilogue_stats(bn)), residual=residual, relu=relu, conv_only=conv_only)


# split planes of folded weights, per (weight memory, input geometry, arithmetic): constant at inference.  Keyed by the
# weight's address and version so that the tensor-level operator below (which a TorchScript trace replays with the folded
# weight as a graph constant) finds them again without a module to hang them on.
# The entry keeps its weight alive (the address is the key: it must not be reused while the entry exists), so the cache is
# a small LRU — every new input geometry of a live weight adds an entry — and the module path evicts a weight's entries
# when it replaces or drops the folded copy (`_use_folded` after training, `fold_batchnorm` again, `unfold_batchnorm`):
# ADVICE r4 — the unbounded dict grew by one entry per folded convolution at every train / eval alternation.
_PLANES = collections.OrderedDict()     # key -> (planes, wbits, weight)
_PLANES_MAX = 256


def _drop_planes_of(weight):
    if weight is None:
        return
    for key in [k for k, v in _PLANES.items() if v[2] is weight]:
        del _PLANES[key]


def _folded_planes(weight, d, n, h, w, h2, st, dev):
    key = (weight.data_ptr(), weight._version, n, h, w, h2)
    hit = _PLANES.get(key)
    if hit is None:
        lib = _C.load()
        planes = torch.empty((lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0),), dtype=torch.uint8, device=dev)
        wbits = None
        if h2:
            wbits = HF.absmax_bits(weight, st)
            _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), weight.data_ptr(), 0, planes.data_ptr(), wbits.data_ptr(), st)
        else:
            _C.call('evk_conv2d_split_weight', ctypes.byref(d), weight.data_ptr(), 0, planes.data_ptr(), st)
        hit = _PLANES[key] = (planes, wbits, weight)
        while len(_PLANES) > _PLANES_MAX:
            _PLANES.popitem(last=False)
    else:
        _PLANES.move_to_end(key)
    return hit[0], hit[1]


def _folded_conv(x, weight, bias, residual, stride, padding, dilation, relu):
    """y = conv(x, weight) + bias (+ residual) (ReLU) with `weight` = the folded [Cout][kh][kw][Cin_p] tensor of _fold_pair
    (channel-padded K axis), tensors only: the body of the `ever_amd::conv2d_folded` operator."""
    x = HF.unpacked(HF.as_nhwc(x, 'folded conv'))
    n, cin, h, w = x.shape
    # the folded weight is either a channels_last OIHW tensor (cin == cin_p) or a plain [cout, kh, kw, cin_p] one (padded K)
    if weight.dim() == 4 and weight.shape[1] == cin and weight.stride(1) == 1:
        cout, kh, kw, cin_p = weight.shape[0], weight.shape[2], weight.shape[3], cin
    else:
        cout, kh, kw, cin_p = weight.shape
    if HF._pad4(cin) != cin_p:
        raise ValueError(f'folded conv: input has {cin} channels, the folded weight expects {cin_p} (padded)')
    dev, st = x.device, HF._stream()
    if cin_p != cin:
        xk = HF._pad_last(x.data_ptr(), n * h * w, cin, cin_p, dev)
        x_ptr = xk.data_ptr()
    else:
        x_ptr = x.data_ptr()
    d = HF._conv_desc(n, h, w, cin_p, cout, kh, kw, HF._pair(stride), HF._pair(padding), HF._pair(dilation))
    y = HF.empty_nhwc(n, cout, d.Ho, d.Wo, dev)
    res_ptr = None
    if residual is not None:
        residual = HF.as_nhwc(residual, 'folded conv residual')
        if residual.shape != y.shape:
            raise ValueError('folded conv: residual shape mismatch')
        res_ptr = residual.data_ptr()
    flags = 1 if relu else 0
    math = HF.get_conv_math()
    if math in ('f16x2', 'bf16x3', 'bf16') and cin_p == cin and cin % 8 == 0:
        # constant at inference: split once per input geometry and arithmetic (the plane layout follows the kernel the
        # descriptor selects)
        h2 = math == 'f16x2'
        planes, wbits = _folded_planes(weight, d, n, h, w, h2, st, dev)
        if h2:
            xbits = HF.absmax_bits(x, st)
            _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), x_ptr, xbits.data_ptr(), planes.data_ptr(), wbits.data_ptr(),
                    bias.data_ptr(), res_ptr, y.data_ptr(), flags, None, 0, ctypes.byref(ctypes.c_int32(0)), None, st)
        else:
            _C.call('evk_conv2d_fwd_x3_res', ctypes.byref(d), x_ptr, planes.data_ptr(), bias.data_ptr(), res_ptr,
                    y.data_ptr(), flags, st)
    else:
        _C.call('evk_conv2d_fwd_res', ctypes.byref(d), x_ptr, weight.data_ptr(), bias.data_ptr(), res_ptr, y.data_ptr(),
                flags, st)
    return y


def _folded_fake(x, weight, bias, residual, stride, padding, dilation, relu):
    if weight.dim() == 4 and weight.shape[1] == x.shape[1] and weight.stride(1) == 1:
        cout, kh, kw = weight.shape[0], weight.shape[2], weight.shape[3]
    else:
        cout, kh, kw, _ = weight.shape
    ho = (x.shape[2] + 2 * padding[0] - dilation[0] * (kh - 1) - 1) // stride[0] + 1
    wo = (x.shape[3] + 2 * padding[1] - dilation[1] * (kw - 1) - 1) // stride[1] + 1
    return HF._oplib.nhwc_like(x, x.shape[0], cout, ho, wo)


_folded_conv_op = HF._oplib.traceable(
    'conv2d_folded', '(Tensor x, Tensor weight, Tensor bias, Tensor? residual, int[] stride, int[] padding, int[] dilation, '
                     'bool relu) -> Tensor',
    _folded_conv, impl_fn=lambda x, w, b, r, s, p, d, relu: _folded_conv(x, w, b, r, tuple(s), tuple(p), tuple(d), relu),
    adapt=lambda x, w, b, r, s, p, d, relu: (x, w, b, r, [int(e) for e in HF._pair(s)], [int(e) for e in HF._pair(p)],
                                             [int(e) for e in HF._pair(d)], bool(relu)),
    fake=_folded_fake)


def folded_conv2d(x, conv, residual=None, relu=False):
    f = conv._folded
    if x.shape[1] != f.cin:
        raise ValueError(f'folded conv: input has {x.shape[1]} channels, expected {f.cin}')
    return _folded_conv_op(x, f.weight, f.bias, residual, conv.stride, conv.padding, conv.dilation, relu)
