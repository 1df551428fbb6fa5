"""ResNetEncoder ERModule (API of reference ever/module/resnet.py:72-265): returns [c2, c3, c4(, c5)].

Config keys and their meaning are the reference's: resnet_type, include_conv5, batchnorm_trainable,
pretrained, freeze_at, output_stride (8/16/32 via stride->dilation), with_cp (activation
checkpointing per stage), norm_layer, in_channels.  The image enters as NCHW fp32; one boundary
kernel lays it out NHWC (channels padded to a multiple of 4 for the 16-byte im2col gathers).
"""
from functools import partial

import torch
import torch.nn as nn
from torch.utils import checkpoint as cp

from ..core import logger, registry
from ..interface import ERModule
from ..util import param_util
from . import _resnets
from ..hip import functional as HF
from .layers import Conv2d

_logger = logger.get_logger()
__all__ = ['ResNetEncoder']

for _name in ('resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152', 'resnext50_32x4d', 'resnext101_32x4d',
              'resnext101_32x8d', 'resnet50_v1c', 'resnet101_v1c'):
    registry.MODEL.register(_name, getattr(_resnets, _name), verbose=False)


def patch_first_conv(module, new_in_channels, default_in_channels=3):
    """Pretrained RGB stem -> n bands: tile the RGB filters cyclically, rescale by 3/n (resnet.py:55-69)."""
    import torch
    w = module.weight.detach()
    new_w = torch.empty(module.out_channels, new_in_channels // module.groups, *module.kernel_size)
    for i in range(new_in_channels):
        new_w[:, i] = w[:, i % default_in_channels]
    new_w = new_w * (default_in_channels / new_in_channels)
    module.in_channels = new_in_channels
    module.weight = nn.parameter.Parameter(new_w.contiguous(memory_format=torch.channels_last))


@registry.MODEL.register(verbose=False)
class ResNetEncoder(ERModule):
    def __init__(self, config):
        super().__init__(config)
        if self.config.output_stride not in (8, 16, 32):
            raise ValueError('output_stride must be 8, 16 or 32.')
        self.resnet = registry.MODEL[self.config.resnet_type](pretrained=self.config.pretrained,
                                                              norm_layer=self.config.norm_layer)
        _logger.info('ResNetEncoder: pretrained = {}'.format(self.config.pretrained))
        self.resnet._modules.pop('fc')
        if not self.config.batchnorm_trainable:
            self._frozen_res_bn()
        self._freeze_at(at=self.config.freeze_at)
        if self.config.output_stride == 16:
            self.resnet.layer4.apply(partial(self._nostride_dilate, dilate=2))
        elif self.config.output_stride == 8:
            self.resnet.layer3.apply(partial(self._nostride_dilate, dilate=2))
            self.resnet.layer4.apply(partial(self._nostride_dilate, dilate=4))
        if self.config.in_channels != 3:
            self.reset_in_channels(self.config.in_channels)

    def reset_in_channels(self, in_channels):
        if in_channels == 3:
            return
        if self.resnet.deep_stem:
            if not self.config.pretrained:
                # a fresh, default-initialised conv (NOT kaiming_normal) as in the reference (resnet.py:104-106)
                self.resnet.stem.add_module('0', Conv2d(in_channels, 32, 3, 2, 1, bias=False))
            else:
                patch_first_conv(self.resnet.stem[0], in_channels)
        else:
            if not self.config.pretrained:
                self.resnet.add_module('conv1', Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3,
                                                       bias=False))
            else:
                patch_first_conv(self.resnet.conv1, in_channels)
        _logger.info(f'ResNetEncoder: in_channels = {in_channels}')

    # layer1..layer4 are exposed read/write so plug-ins can wrap stages (reference resnet.py:119-153)
    def _stage(name):  # noqa: N805
        def getter(self):
            return getattr(self.resnet, name)

        def setter(self, value):
            delattr(self.resnet, name)
            setattr(self.resnet, name, value)

        return property(getter, setter)

    layer1, layer2, layer3, layer4 = _stage('layer1'), _stage('layer2'), _stage('layer3'), _stage('layer4')
    del _stage

    def _frozen_res_bn(self):
        _logger.info('ResNetEncoder: freeze all BN layers')
        param_util.freeze_modules(self.resnet, nn.modules.batchnorm._BatchNorm)
        for m in self.resnet.modules():
            if isinstance(m, nn.modules.batchnorm._BatchNorm):
                m.eval()

    def _freeze_at(self, at=2):
        r = self.resnet
        stem = [r.stem] if r.deep_stem else [r.conv1, r.bn1]
        for level, mods in enumerate([stem, [r.layer1], [r.layer2], [r.layer3], [r.layer4]], start=1):
            if at >= level:
                for m in mods:
                    param_util.freeze_params(m)

    def _run_stage(self, stage, x, use_cp):
        if use_cp and x.requires_grad:
            return cp.checkpoint(stage, x, use_reentrant=False)
        return stage(x)

    def forward(self, inputs):
        x = inputs
        r = self.resnet
        x = r.stem_pool_forward(x)
        wcp = self.config.with_cp
        # A stage output feeds the next stage AND (later) the caller.  With gradient slots the caller's gradient is
        # added inside the next stage's first data-gradient launch (hip/conv.py:GradSlot) instead of by an
        # autograd add pass over the whole map; off under activation checkpointing (the fork nodes are rebuilt then).
        slots = HF.grad_slots_enabled() and not any(wcp) and torch.is_grad_enabled()
        outs = []

        def stage(layer, t, use_cp):
            slot = None
            if slots and t.requires_grad:
                slot = HF.GradSlot()
                t._evk_grad_slot = slot
            y = self._run_stage(layer, t, use_cp)
            if slot is not None:
                del t._evk_grad_slot
                if slot.claimed:
                    outs[-1] = HF.slot_output(t, slot)
            return y
        c2 = self._run_stage(r.layer1, x, wcp[0])    # os 4 : 64 (r18/34) / 256 ch
        outs.append(c2)
        c3 = stage(r.layer2, c2, wcp[1])             # os 8 : 128 / 512
        outs.append(c3)
        c4 = stage(r.layer3, c3, wcp[2])             # os 16: 256 / 1024
        outs.append(c4)
        if self.config.include_conv5:
            outs.append(stage(r.layer4, c4, wcp[3]))  # os 32: 512 / 2048
        return outs

    def set_default_config(self):
        self.config.update(dict(
            resnet_type='resnet50',
            include_conv5=True,
            batchnorm_trainable=True,
            pretrained=False,
            freeze_at=0,
            output_stride=32,  # 8, 16 or 32
            with_cp=(False, False, False, False),
            norm_layer=nn.BatchNorm2d,
            in_channels=3,
        ))

    def train(self, mode=True):
        super().train(mode)
        self._freeze_at(self.config.freeze_at)
        if mode and not self.config.batchnorm_trainable:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()  # frozen BN keeps using its running statistics
        return self

    @staticmethod
    def _nostride_dilate(m, dilate):
        """Turn the stage's stride-2 convs into stride 1 and dilate the 3x3s (reference resnet.py:236-251)."""
        if not isinstance(m, nn.Conv2d):
            return
        if m.stride == (2, 2):
            m.stride = (1, 1)
            if m.kernel_size == (3, 3):
                m.dilation = (dilate // 2, dilate // 2)
                m.padding = (dilate // 2, dilate // 2)
        elif m.kernel_size == (3, 3):
            m.dilation = (dilate, dilate)
            m.padding = (dilate, dilate)
