"""FreeNet-style patch-free hyperspectral segmentation network on the HIP layers (SURVEY §8 f4, config C5:
200-band input, GroupNorm convolution blocks, nearest-x2 top-down path, global — whole-image — forward).

The reference tree carries no definition of this model (it lives in the external FreeNet project); the structure
below follows the published architecture (Zheng et al., "FPGA: Fast Patch-Free Global Learning Framework for
Fully End-to-End Hyperspectral Image Classification", TGRS 2020): a plain encoder of 3x3 conv - GroupNorm - ReLU
blocks at four resolutions joined by stride-2 3x3 convolutions + ReLU, 1x1 lateral reductions, nearest-x2 top-down
fusion with 3x3 fuse convolutions, a 1x1 classifier at full resolution.  Parity is pinned to the stock-torch
restatement in oracle/freenet_ref.py (same state-dict keys), not to reference outputs ("parity unpinned").
What it exercises on the kernels: Cin = 200 in the split-MFMA implicit GEMM, spatial sizes that are not powers of
two (610 x 340 padded to a multiple of 8), GroupNorm on full maps, stride-2 3x3 data gradients, nearest top-down."""
import math

import torch
import torch.nn as nn

from ..core import registry
from ..hip import functional as HF
from ..interface import ERModule
from .layers import Conv2d, GroupNorm, HipSequential, ReLU

__all__ = ['FreeNet', 'divisible_pad']


def conv3x3_gn_relu(cin, cout, groups):
    return HipSequential(Conv2d(cin, cout, 3, 1, 1), GroupNorm(groups, cout), ReLU(inplace=True))


def downsample2x(cin, cout):
    return HipSequential(Conv2d(cin, cout, 3, 2, 1), ReLU(inplace=True))


def repeat_block(channels, groups, n):
    return HipSequential(*[conv3x3_gn_relu(channels, channels, groups) for _ in range(n)])


def divisible_pad(x, size_divisor, value=0.0):
    """zero-pad H, W at the bottom / right to multiples of `size_divisor` (reference preprocess/function.py:35-64)"""
    h, w = x.shape[-2:]
    nh, nw = math.ceil(h / size_divisor) * size_divisor, math.ceil(w / size_divisor) * size_divisor
    if (nh, nw) == (h, w):
        return x
    return torch.nn.functional.pad(x, (0, nw - w, 0, nh - h), value=value)


@registry.MODEL.register(verbose=False)
class FreeNet(ERModule):
    def __init__(self, config):
        super().__init__(config)
        r = int(16 * self.config.reduction_ratio)
        chans = [int(c * self.config.reduction_ratio / r) * r for c in (96, 128, 192, 256)]
        nb = self.config.num_blocks
        ops = [conv3x3_gn_relu(self.config.in_channels, chans[0], r), repeat_block(chans[0], r, nb[0]), nn.Identity()]
        for i in range(1, 4):
            ops += [downsample2x(chans[i - 1], chans[i]), repeat_block(chans[i], r, nb[i]), nn.Identity()]
        self.feature_ops = nn.ModuleList(ops)
        inner = int(128 * self.config.reduction_ratio)
        self.reduce_1x1convs = nn.ModuleList([Conv2d(c, inner, 1) for c in chans])
        self.fuse_3x3convs = nn.ModuleList([Conv2d(inner, inner, 3, 1, 1) for _ in range(4)])
        self.cls_pred_conv = Conv2d(inner, self.config.num_classes, 1)

    def features(self, x):
        x = HF.as_nhwc(x, 'FreeNet input')
        feats = []
        for op in self.feature_ops:
            x = op(x)
            if isinstance(op, nn.Identity):
                feats.append(x)
        inner = [conv(f) for conv, f in zip(self.reduce_1x1convs, feats)]
        inner.reverse()
        out = inner[0]
        for i in range(len(inner) - 1):
            out = self.fuse_3x3convs[i](HF.upsample_nearest2x_add(out, inner[i + 1]))
        return out

    def forward(self, x, y=None, w=None, **kwargs):
        logit = self.cls_pred_conv(self.features(x))
        if self.training:
            if isinstance(y, dict):
                y, w = y['mask'], y.get('weight', w)
            return dict(cls_loss=self.loss(logit, y, w))
        return logit

    def loss(self, logit, y, weight=None):
        """cross entropy over the labelled pixels; labels are 1-based with 0 = unlabelled in the HSI datasets, i.e.
        class = y - 1 and ignore_index = -1; an optional 0/1 `weight` map (the stratified training mask of the global
        learning scheme) removes further pixels."""
        t = y.long() - 1
        if weight is not None:
            t = torch.where(weight > 0, t, torch.full_like(t, -1))
        from . import loss as L
        return L.cross_entropy(logit, t, ignore_index=-1)

    def set_default_config(self):
        self.config.update(dict(in_channels=200, num_classes=16, num_blocks=(1, 1, 1, 1), reduction_ratio=1.0))
