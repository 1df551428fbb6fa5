"""HIP-backed counterparts of the stock torch.nn layers the reference's hot path is built from.

Each class SUBCLASSES its torch.nn namesake and keeps its constructor, parameters, buffers and
state-dict keys, so (a) reference checkpoints load unchanged, (b) `isinstance(m, nn.BatchNorm2d)`
style code in user projects (freezing, init loops; e.g. reference ever/module/resnet.py:155-173,
_resnets.py:164-169) keeps working, and (c) `to_hip(model)` can retarget an existing stock model by
swapping classes.  Only `forward` changes: it calls the gfx950 kernels through the C-ABI.
Activations flow as logical-NCHW / memory-NHWC tensors (see hip/functional.py).
"""
import torch
import torch.nn as nn

from ..hip import functional as HF

__all__ = ['Conv2d', 'ConvTranspose2d', 'BatchNorm2d', 'ReLU', 'MaxPool2d', 'UpsamplingBilinear2d', 'AdaptiveAvgPool2d', 'Identity',
           'Dropout', 'GELU', 'HipSequential', 'run_sequence', 'to_hip']

Identity = nn.Identity


class Conv2d(nn.Conv2d):
    """nn.Conv2d (zero padding) on the MFMA implicit-GEMM kernels; weight kept OHWI in memory.  groups > 1 (the ResNeXt
    bodies of reference _resnets.py:291-324) runs as a dense convolution with the block-diagonal weight
    (HF.grouped_dense_weight: exact zeros outside the groups)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        _check_conv(self)
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x, relu=False, bn_stats=False):
        """bn_stats: a training-mode BatchNorm consumes the result next (hip/conv.py:conv2d)"""
        w = self.weight if self.groups == 1 else HF.grouped_dense_weight(self.weight, self.groups)
        y = HF.conv2d(x, w, self.bias, self.stride, self.padding, self.dilation, relu=relu, bn_stats=bn_stats)
        if self._forward_hooks and getattr(y, '_evk_bn_parts', None) is not None and len(y._evk_bn_parts) > 2:
            # a forward hook may hang a tensor hook on y: its gradient (the BatchNorm's dx) must then stay fp32
            y._evk_bn_parts = y._evk_bn_parts[:2] + (False,)
        return y


class ConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d (groups=1, zero padding) on the residue-class data-gradient kernel (BASELINE.json north_star:
    "transposed-conv lowered to MFMA"); weight kept [Cin][kh][kw][Cout] in memory.  The reference's hot path has no
    call site for it; parity is against torch.nn.ConvTranspose2d."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        _check_conv(self)
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x, output_size=None):
        op = self.output_padding
        if output_size is not None:
            op = self._output_padding(x, output_size, self.stride, self.padding, self.kernel_size, 2, self.dilation)
        return HF.conv_transpose2d(x, self.weight, self.bias, self.stride, self.padding, op, self.dilation)


def _check_conv(m):
    if m.groups != 1 and isinstance(m, nn.ConvTranspose2d):
        raise NotImplementedError('ever_amd ConvTranspose2d: groups != 1 has no HIP kernel')
    if m.padding_mode != 'zeros' or isinstance(m.padding, str):
        raise NotImplementedError('ever_amd Conv2d: only explicit zero padding is implemented')


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d with optional fused residual add and ReLU (one apply pass over HBM)."""

    def forward(self, x, residual=None, relu=False, conv_only=False, lazy_res=False):
        """lazy_res: see hip/norm.py:batch_norm_act (residual blocks of this package only).
        conv_only: the result is read by ONE convolution of this package (and that convolution's weight gradient) and
        by nothing else — under the f16x2 arithmetic the pass may then store it already split ("packed",
        hip/norm.py:batch_norm_act); any other reader would see raw words."""
        if self.momentum is None:
            raise NotImplementedError('ever_amd BatchNorm2d: cumulative moving average (momentum=None) unsupported')
        training = self.training or (self.running_mean is None)
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            # counted on the host and folded into the buffer when it is read (state_dict / checkpoint):
            # 68 one-element device increments per step would be 68 extra launches on the stream
            self._nbt_pending = getattr(self, '_nbt_pending', 0) + 1
        rm = self.running_mean if (not self.training or self.track_running_stats) else None
        rv = self.running_var if (not self.training or self.track_running_stats) else None
        # a forward hook on this module would be handed the packed words: store fp32 then
        conv_only = conv_only and not self._forward_hooks
        return HF.batch_norm_act(x, self.weight, self.bias, rm, rv, training, self.momentum, self.eps,
                                 residual=residual, relu=relu, pack_out=conv_only,
                                 lazy_res=lazy_res and not self._forward_hooks)


    def forward_relu_pool(self, x):
        """max_pool3x3s2(relu(self(x))): the ResNet stem's tail; fused into one pass each way in training mode
        (hip/norm.py:batch_norm_relu_max_pool)."""
        training = self.training or (self.running_mean is None)
        if not training or self.momentum is None:
            return HF.max_pool3x3s2(self.forward(x, relu=True))
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self._nbt_pending = getattr(self, '_nbt_pending', 0) + 1
        rm = self.running_mean if self.track_running_stats else None
        rv = self.running_var if self.track_running_stats else None
        return HF.batch_norm_relu_max_pool(x, self.weight, self.bias, rm, rv, self.momentum, self.eps)

    def flush_num_batches_tracked(self):
        pending = getattr(self, '_nbt_pending', 0)
        if pending and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(pending)
        self._nbt_pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_num_batches_tracked()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._nbt_pending = 0
        super()._load_from_state_dict(*args, **kwargs)


class ReLU(nn.ReLU):
    def forward(self, x):
        return HF.relu(x)


class MaxPool2d(nn.MaxPool2d):
    """Only the ResNet stem configuration (kernel 3, stride 2, padding 1) has a kernel."""

    def forward(self, x):
        cfg = (_one(self.kernel_size), _one(self.stride), _one(self.padding), _one(self.dilation), self.ceil_mode)
        if cfg != (3, 2, 1, 1, False):
            raise NotImplementedError(f'ever_amd MaxPool2d: only (k=3,s=2,p=1) is implemented, got {cfg}')
        return HF.max_pool3x3s2(x)


def _one(v):
    if isinstance(v, (tuple, list)):
        if len(set(v)) != 1:
            return tuple(v)
        return v[0]
    return v


class UpsamplingBilinear2d(nn.UpsamplingBilinear2d):
    """bilinear, align_corners=True, by scale_factor."""

    def forward(self, x):
        if self.scale_factor is None:
            raise NotImplementedError('ever_amd UpsamplingBilinear2d: give scale_factor')
        return HF.upsample_bilinear(x, self.scale_factor)


class AdaptiveAvgPool2d(nn.AdaptiveAvgPool2d):
    def forward(self, x):
        if _one(self.output_size) != 1:
            raise NotImplementedError('ever_amd AdaptiveAvgPool2d: only output_size=1 is implemented')
        return HF.global_avg_pool(x)


class GroupNorm(nn.GroupNorm):
    """nn.GroupNorm on evk_gn_fwd / evk_gn_bwd (reference fs_relation.py:88-116); `relu=True` fuses the ReLU."""

    def forward(self, x, relu=False):
        from ..hip import functional_next as HN
        return HN.group_norm_act(x, self.num_groups, self.weight, self.bias, self.eps, relu=relu)


class Dropout2d(nn.Dropout2d):
    """nn.Dropout2d: the [N, C] keep-mask is drawn with torch's generator, the scaling is one HIP pass."""

    def forward(self, x):
        from ..hip import functional_next as HN
        return HN.dropout2d(x, self.p, self.training)


class Dropout(nn.Dropout):
    """nn.Dropout (element-wise): mask from torch's generator, masking + scaling in one HIP pass."""

    def forward(self, x):
        from ..hip import functional_next as HN
        return HN.dropout(x, self.p, self.training)


class GELU(nn.GELU):
    def forward(self, x):
        if self.approximate != 'none':
            raise NotImplementedError("ever_amd GELU: only the exact (erf) form is implemented")
        from ..hip import functional_next as HN
        return HN.gelu(x)


def _unwrap(m):
    # reference ops.Bf16compatible wraps the upsampling module (ever/module/ops.py:152-166); the HIP
    # path is fp32 end to end so the wrapper is transparent.
    inner = getattr(m, '_inner_module', None)
    return inner if inner is not None else m


def _fold():
    from . import fold
    return fold


def run_sequence(mods, x):
    """Run modules in order with peephole fusion: conv->ReLU (epilogue), BN->ReLU (one pass)."""
    mods = [_unwrap(m) for m in mods]
    i, n = 0, len(mods)
    while i < n:
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < n else None
        if isinstance(m, Conv2d) and getattr(m, '_folded', None) is not None and nxt is not None \
                and _fold()._use_folded(m, nxt):
            # inference: conv + folded BatchNorm (+ ReLU) in one launch (module/fold.py)
            folded_conv2d = _fold().folded_conv2d
            relu = i + 2 < n and isinstance(mods[i + 2], nn.ReLU)
            x = folded_conv2d(x, m, relu=relu)
            i += 3 if relu else 2
        elif isinstance(m, Conv2d) and isinstance(nxt, nn.ReLU):
            x = m(x, relu=True)
            i += 2
        elif isinstance(m, Conv2d) and type(nxt) is BatchNorm2d and nxt.training:
            x = m(x, bn_stats=True)     # the BatchNorm that follows takes its statistics from this epilogue
            i += 1
        elif isinstance(m, (BatchNorm2d, GroupNorm, nn.SyncBatchNorm)) and isinstance(nxt, nn.ReLU):
            x = m(x, relu=True)
            i += 2
        elif isinstance(m, nn.Identity):
            i += 1
        else:
            x = m(x)
            i += 1
    return x


class HipSequential(nn.Sequential):
    """nn.Sequential (same child names => same state-dict keys) whose forward fuses neighbours."""

    def forward(self, x):
        return run_sequence(list(self), x)


_SWAP = {
    nn.Conv2d: Conv2d, nn.ConvTranspose2d: ConvTranspose2d, nn.BatchNorm2d: BatchNorm2d, nn.ReLU: ReLU, nn.MaxPool2d: MaxPool2d,
    nn.UpsamplingBilinear2d: UpsamplingBilinear2d, nn.AdaptiveAvgPool2d: AdaptiveAvgPool2d,
    nn.GroupNorm: GroupNorm, nn.Dropout2d: Dropout2d, nn.Dropout: Dropout, nn.GELU: GELU, nn.Sequential: HipSequential,
}


def to_hip(model):
    """Retarget a model built from stock torch.nn layers onto the HIP kernels, in place, by swapping
    the class of every supported layer (parameters, buffers and state-dict keys are untouched).
    Unsupported configurations raise when first executed, never silently fall back."""
    for m in model.modules():
        hip_cls = _SWAP.get(type(m))
        if hip_cls is None:
            continue
        m.__class__ = hip_cls
        if hip_cls in (Conv2d, ConvTranspose2d):
            _check_conv(m)
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return model
