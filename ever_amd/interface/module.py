"""ERModule: the plugin base class of EVer models (API of reference ever/interface/module.py:12-108).

Training protocol kept verbatim: `forward` returns a dict, every entry whose key ends in `loss` is
summed and differentiated by `backward`; `apply_gradients` = (unscale) -> clip -> step -> zero_grad.
"""
import re

import torch
import torch.nn as nn
from torch.nn.utils import clip_grad

from ..core import checkpoint
from ..core.logger import info
from .configurable import ConfigurableMixin


class ERModule(nn.Module, ConfigurableMixin):
    __Keys__ = ['GLOBAL', ]

    def __init__(self, config=None):
        nn.Module.__init__(self)
        ConfigurableMixin.__init__(self, dict() if config is None else config)
        for key in ERModule.__Keys__:
            if key not in self.config:
                self.config[key] = dict()

    def forward(self, *input):
        raise NotImplementedError

    def set_default_config(self):
        raise NotImplementedError('The default config should be overridden.')

    def init_from_weight_file(self):
        """config.GLOBAL.weight = dict(path=..., excepts=<regex>) -> non-strict load, stripping the
        `module.` (DDP) and `_orig_mod.` (torch.compile) prefixes; reference module.py:31-68."""
        weight = self.config.GLOBAL.get('weight') if isinstance(self.config.GLOBAL, dict) else None
        if not isinstance(weight, dict) or weight.get('path') is None:
            return
        state = torch.load(weight['path'], map_location='cpu', weights_only=False)
        if checkpoint.is_checkpoint(state):
            state = state[checkpoint.CheckPoint.MODEL]
        pattern = re.compile(weight['excepts']) if weight.get('excepts') is not None else None
        picked = {}
        for k, v in state.items():
            if k.startswith('module.'):
                k = k.replace('module.', '')
            if '_orig_mod.' in k:
                k = k.replace('_orig_mod.', '')
            if pattern is not None and pattern.match(k):
                continue
            picked[k] = v
        res = self.load_state_dict(picked, strict=False)
        info('Load weights from: {}'.format(weight['path']))
        info(f'missing_keys ({len(res.missing_keys)}): {res.missing_keys}')
        info(f'unexpected_keys ({len(res.unexpected_keys)}): {res.unexpected_keys}')

    def log_info(self):
        return dict()

    def custom_param_groups(self):
        return [{'params': self.parameters()}, ]

    def backward(self, loss_dict, amp, scaler, **kwargs):
        total_loss = sum(loss_dict.values())
        if amp and scaler is not None:
            scaler.scale(total_loss).backward()
        else:
            total_loss.backward()

    def apply_gradients(self, optimizer, amp, scaler, **kwargs):
        if amp and scaler is not None:
            scaler.unscale_(optimizer)
            grad_info = self.clip_grad(optimizer)
            scaler.step(optimizer)
            scaler.update()
        else:
            grad_info = self.clip_grad(optimizer)
            optimizer.step()
        optimizer.zero_grad()
        # whatever optimizer ran (a custom one may write through `.data`, which moves no version counter the plane
        # cache can see): the cached bf16 weight planes of the split-arithmetic convolutions are stale from here on
        from ..hip import weight_planes
        weight_planes.note_weights_changed()
        return grad_info

    def clip_grad(self, optimizer):
        """Clips only when the optimizer config carries `grad_clip` (reference module.py:96-108).
        A fused optimizer (ever_amd.opt.FusedSGD) clips inside its own step and reports the norm."""
        grad_info = dict()
        er_config = getattr(optimizer, 'er_config', {})
        if 'grad_clip' in er_config:
            cfg = er_config.get('grad_clip', dict(max_norm=35, norm_type=2))
            if hasattr(optimizer, 'fused_clip'):
                optimizer.fused_clip(**cfg)
                grad_info['grad_norm'] = optimizer.last_grad_norm
            else:
                grad_info['grad_norm'] = clip_grad.clip_grad_norm_(
                    [p for p in self.parameters() if p.requires_grad], **cfg)
        return grad_info
