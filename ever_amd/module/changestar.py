"""ChangeStar-style bitemporal change detection on the HIP layers (configuration C4: two 3x512x512 dates
per sample): any per-pixel segmentation feature extractor + ChangeMixin.

The reference tree carries no definition of ChangeMixin / ChangeStar (they live in the external ChangeStar project,
Zheng et al., "Change is Everywhere: Single-Temporal Supervised Object Change Detection in Remote Sensing Imagery",
ICCV 2021); the structure below follows the published one: the two dates' feature maps are concatenated along the
channels in BOTH orders, the two orders are stacked along the batch and run through `num_convs` 3x3 conv-BN-ReLU
layers of `inner_channels`, a 3x3 one-channel classifier and a bilinear x`scale_factor` up-sampling.  Parity is pinned
to the stock-torch restatement in oracle/changestar_ref.py (same state-dict keys): "parity unpinned" by the reference.
Every operator is one the path already has: channel concat (evk_concat_channels), the split-MFMA 3x3 convolutions
(Cin = 512 -> 16: a 16-wide GEMM N), BatchNorm+ReLU, align-corners bilinear; the batch stacking is a device copy."""
import torch
import torch.nn as nn

from ..core import registry
from ..hip import functional as HF
from ..hip import functional_next as HN
from ..interface import ERModule
from . import loss as L
from .fs_relation import FarSegHead
from .layers import BatchNorm2d, Conv2d, HipSequential, ReLU, UpsamplingBilinear2d
from .resnet import ResNetEncoder

__all__ = ['ChangeMixin', 'ChangeStarFarSeg']


class ChangeMixin(nn.Module):
    def __init__(self, in_channels=256 * 2, inner_channels=16, num_convs=4, scale_factor=4.0):
        super().__init__()
        layers = [HipSequential(Conv2d(in_channels, inner_channels, 3, 1, 1), BatchNorm2d(inner_channels), ReLU(True))]
        layers += [HipSequential(Conv2d(inner_channels, inner_channels, 3, 1, 1), BatchNorm2d(inner_channels), ReLU(True))
                   for _ in range(num_convs - 1)]
        layers.append(Conv2d(inner_channels, 1, 3, 1, 1))
        layers.append(UpsamplingBilinear2d(scale_factor=scale_factor))
        self.convs = HipSequential(*layers)

    def forward(self, t1, t2):
        """t1, t2: [N, C, h, w] feature maps of the two dates -> (change logits t1->t2, t2->t1), each [N, 1, H, W]"""
        n = t1.shape[0]
        both = torch.cat([HN.concat_channels(t1, t2), HN.concat_channels(t2, t1)], dim=0)   # NHWC: a plain row append
        out = self.convs(HF.as_nhwc(both, 'ChangeMixin'))
        return out[:n], out[n:]


@registry.MODEL.register(verbose=False)
class ChangeStarFarSeg(ERModule):
    """FarSeg (ResNet + FPN + FS-Relation + decoder) as the feature extractor of ChangeStar.

    forward(x[N, 2*C, H, W], y): the two dates are stacked along the channels (date 1 first).  Training returns the
    losses of whatever labels `y` holds: `cls` / `cls2` (semantic masks of date 1 / 2 -> BCE + dice on the semantic
    head) and `change` (binary change mask -> BCE on both orders, averaged).  Eval returns probabilities
    dict(t1, t2, change) with change = sigmoid of the mean of the two orders' logits."""

    def __init__(self, config):
        super().__init__(config)
        self.en = ResNetEncoder(self.config.encoder)
        self.head = FarSegHead(self.config.head)
        dec = self.head.fpn_decoder
        feat_c = dec.blocks[0][0][0].out_channels
        self.change = ChangeMixin(in_channels=2 * feat_c, **self.config.change_mixin)

    def forward(self, x, y=None):
        n, c2 = x.shape[:2]
        c = c2 // 2
        dates = torch.cat([x[:, :c], x[:, c:]], dim=0)               # (t b): the first N rows are date 1
        feat = self.head.features(self.en(HF.as_nhwc(dates.contiguous(), 'ChangeStar input')))
        dec = self.head.fpn_decoder
        sem = dec.classifier(dec.dropout(feat))                      # [2N, 1, H, W]
        c12, c21 = self.change(feat[:n], feat[n:])
        if self.training:
            return self.loss(sem[:n], sem[n:], c12, c21, y)
        with torch.no_grad():
            return dict(t1=torch.sigmoid(sem[:n]), t2=torch.sigmoid(sem[n:]), change=torch.sigmoid(0.5 * (c12 + c21)))

    def loss(self, s1, s2, c12, c21, y):
        ig = self.config.loss.ignore_index
        out = dict()
        if 'cls' in y:
            out['t1_bce_loss'] = L.binary_cross_entropy_with_logits(s1, y['cls'], ignore_index=ig)
            out['t1_dice_loss'] = L.dice_loss_with_logits(s1, y['cls'], ignore_index=ig)
        if 'cls2' in y:
            out['t2_bce_loss'] = L.binary_cross_entropy_with_logits(s2, y['cls2'], ignore_index=ig)
            out['t2_dice_loss'] = L.dice_loss_with_logits(s2, y['cls2'], ignore_index=ig)
        if 'change' in y:
            out['change12_bce_loss'] = 0.5 * L.binary_cross_entropy_with_logits(c12, y['change'], ignore_index=ig)
            out['change21_bce_loss'] = 0.5 * L.binary_cross_entropy_with_logits(c21, y['change'], ignore_index=ig)
        return out

    def set_default_config(self):
        self.config.update(dict(
            encoder=dict(resnet_type='resnet50', include_conv5=True, batchnorm_trainable=True, pretrained=False,
                         freeze_at=0, output_stride=32, with_cp=(False, False, False, False), in_channels=3),
            head=dict(),
            change_mixin=dict(inner_channels=16, num_convs=4, scale_factor=4.0),
            loss=dict(ignore_index=255),
        ))
