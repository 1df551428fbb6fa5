"""Checkpoint files compatible with the reference (ever/core/checkpoint.py:21-141):
`checkpoint-{step}.pth` = OrderedDict{model, opt, global_step} with UNWRAPPED model keys, plus the
index `checkpoint_info.json` = {"last": {"step", "name"}, "<step>": name}."""
import json
import os
from collections import OrderedDict

import torch


def is_checkpoint(obj):
    if isinstance(obj, CheckPoint):
        return True
    return isinstance(obj, OrderedDict) and all(
        k in obj for k in (CheckPoint.MODEL, CheckPoint.OPTIMIZER, CheckPoint.GLOBALSTEP))


class CheckPoint:
    MODEL = 'model'
    OPTIMIZER = 'opt'
    GLOBALSTEP = 'global_step'
    LASTCHECKPOINT = 'last'
    CHECKPOINT_NAME = 'checkpoint_info.json'

    def __init__(self, launcher=None):
        self._launcher = launcher
        self._global_step = 0
        self._index = {CheckPoint.LASTCHECKPOINT: dict(step=0, name='')}
        self._read_index()

    @property
    def global_step(self):
        return self._global_step

    def set_global_step(self, value):
        if value < 0:
            raise ValueError('The global step must be larger than zero.')
        self._global_step = value

    def step(self):
        self._global_step += 1

    def set_launcher(self, launcher):
        self._launcher = launcher
        self._read_index()

    @staticmethod
    def get_checkpoint_name(global_step):
        return 'checkpoint-{}.pth'.format(global_step)

    @staticmethod
    def load(filepath):
        return torch.load(filepath, map_location=torch.device('cpu'), weights_only=False)

    @staticmethod
    def load_checkpoint_info(model_dir):
        path = os.path.join(model_dir, CheckPoint.CHECKPOINT_NAME)
        if not os.path.exists(path):
            return None
        with open(path, 'r') as f:
            return json.load(f)

    def _read_index(self):
        if self._launcher is None:
            return
        idx = self.load_checkpoint_info(self._launcher.model_dir)
        if idx is not None:
            self._index = idx

    def save(self, filename=None):
        la = self._launcher
        opt = la.optimizer
        ckpt = OrderedDict()
        ckpt[CheckPoint.MODEL] = la.unwrapped_model.state_dict()
        ckpt[CheckPoint.GLOBALSTEP] = self.global_step
        ckpt[CheckPoint.OPTIMIZER] = ({k: o.state_dict() for k, o in opt.items()} if isinstance(opt, dict)
                                      else opt.state_dict())
        filename = filename or self.get_checkpoint_name(self.global_step)
        torch.save(ckpt, os.path.join(la.model_dir, filename))
        self._index[self.global_step] = filename
        last = self._index[CheckPoint.LASTCHECKPOINT]
        if self.global_step > last['step']:
            last['step'], last['name'] = self.global_step, filename
        with open(os.path.join(la.model_dir, CheckPoint.CHECKPOINT_NAME), 'w') as f:
            json.dump(self._index, f)
        la.logger.info(f'{filename} has been saved.')

    def try_resume(self):
        """index json -> last checkpoint -> model / optimizer / global_step."""
        la = self._launcher
        if la is None:
            return
        idx = self.load_checkpoint_info(la.model_dir)
        if idx is None:
            return
        path = os.path.join(la.model_dir, idx[CheckPoint.LASTCHECKPOINT]['name'])
        ckpt = self.load(path)
        la.unwrapped_model.load_state_dict(ckpt[CheckPoint.MODEL])
        opt = la.optimizer
        if opt is not None:
            if isinstance(opt, dict):
                for k, o in opt.items():
                    o.load_state_dict(ckpt[CheckPoint.OPTIMIZER][k])
            else:
                opt.load_state_dict(ckpt[CheckPoint.OPTIMIZER])
        if la.checkpoint is not None:
            la.checkpoint.set_global_step(ckpt[CheckPoint.GLOBALSTEP])
        la.logger.info(f'{path} has been restored.')


def remove_module_prefix(state):
    """strip leading `module.` / `_orig_mod.` when EVERY key carries a wrapper prefix."""
    if any(('module.' not in k) and ('_orig_mod.' not in k) for k in state):
        return state
    out = {}
    for k, v in state.items():
        if k.startswith('module.'):
            k = k[len('module.'):]
        if k.startswith('_orig_mod.'):
            k = k[len('_orig_mod.'):]
        out[k] = v
    return out


def load_model_state_dict_from_ckpt(filepath):
    ckpt = torch.load(filepath, map_location='cpu', weights_only=False)
    return remove_module_prefix(ckpt[CheckPoint.MODEL])
