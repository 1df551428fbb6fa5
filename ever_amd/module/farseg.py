"""FarSeg segmentation model = ResNetEncoder + FarSegHead + pixel loss, as an ERModule.

The reference ships the blocks (ever/module/resnet.py, fs_relation.py, loss.py) and leaves the
composition to user projects (docs/ERModule.md:25-47: `forward(x, y)` returns a dict of `*_loss`
in training and the prediction in eval).  This is that composition for the configs BASELINE.json
names; state-dict prefixes are `en.` (encoder) and `head.`.
"""
import torch

from ..core import registry
from ..hip import functional as HF
from ..hip import timing
from ..interface import ERModule
from . import loss as L
from .fs_relation import FarSegHead, FarSegPPHead
from .resnet import ResNetEncoder

__all__ = ['FarSeg', 'FarSegPP']


@registry.MODEL.register(verbose=False)
class FarSeg(ERModule):
    HEAD = FarSegHead

    def __init__(self, config):
        super().__init__(config)
        self.en = ResNetEncoder(self.config.encoder)
        self.head = self.HEAD(self.config.head)

    def forward(self, x, y=None):
        HF._require_cuda(x, 'FarSeg input')   # boundary: the stem takes the NCHW (or NHWC) image as it is
        with timing.scope('encoder'):   # bench.py prices the encoder conv stack on its own (north_star target)
            feats = self.en(x)
        logits = self.head(feats)
        if self.training:
            if isinstance(y, dict):
                y = y[self.config.loss.get('label_key', 'cls')]
            return self.loss(logits, y)
        if logits.shape[1] == 1:
            return _SigmoidNoGrad(logits)
        return logits

    def loss(self, logits, y):
        cfg = self.config.loss
        out = dict()
        if logits.shape[1] == 1:
            if cfg.get('bce', True):
                out['bce_loss'] = L.binary_cross_entropy_with_logits(logits, y, ignore_index=cfg.ignore_index)
            if cfg.get('dice', True):
                out['dice_loss'] = L.dice_loss_with_logits(logits, y, ignore_index=cfg.ignore_index)
        else:
            out['cls_loss'] = L.cross_entropy(logits, y, ignore_index=cfg.ignore_index)
            if cfg.get('multiclass_dice', False):
                out['dice_loss'] = L.dice_loss_with_logits(logits, y, ignore_index=cfg.ignore_index)
        return out

    def set_default_config(self):
        self.config.update(dict(
            encoder=dict(resnet_type='resnet50', include_conv5=True, batchnorm_trainable=True, pretrained=False,
                         freeze_at=0, output_stride=32, with_cp=(False, False, False, False), in_channels=3),
            head=dict(),
            loss=dict(ignore_index=255, bce=True, dice=True),
        ))


@registry.MODEL.register(verbose=False)
class FarSegPP(FarSeg):
    """FarSeg++ (BASELINE config C3): the FarSeg composition with the FSRelationV2 head."""
    HEAD = FarSegPPHead


def _SigmoidNoGrad(logits):
    with torch.no_grad():
        return torch.sigmoid(logits)
