"""Learning-rate schedules registered under the reference's names (ever/opt/learning_rate.py:41-157).

`step(global_step, optimizer)` SETS the lr of every param group; the Launcher calls it after the
optimizer step with the pre-increment step counter (the one-step lag of SURVEY §3.1 is the
Launcher's, not the schedule's).  Values are pinned by tests/golden/op_kats.json.
"""
import math

import numpy as np

from ..core import registry
from ..interface import LearningRateBase

__all__ = ['set_lr', 'MultiStepLearningRate', 'PolyLearningRate', 'CosineAnnealingLearningRate',
           'ConstantLearningRate', 'SearchLearningRate']


def set_lr(optimizer, lr):
    for group in optimizer.param_groups:
        group['lr'] = lr


class _Warmup:
    """linear: base*(1-(1-t)(1-r)) ; exp: base*r^(1-t) ; constant: base*r   with t = step/warmup_step."""

    def _init_warmup(self, warmup):
        self.warmup = warmup
        if warmup is None:
            self.warmup_type, self.warmup_step, self.warmup_ratio = None, 0, None
        else:
            self.warmup_type, self.warmup_step, self.warmup_ratio = warmup['type'], warmup['step'], warmup['ratio']

    def get_warmup_lr(self, cur_step, base_lr):
        t = cur_step / self.warmup_step
        if self.warmup_type == 'linear':
            return base_lr * (1 - (1 - t) * (1 - self.warmup_ratio))
        if self.warmup_type == 'exp':
            return base_lr * self.warmup_ratio ** (1 - t)
        if self.warmup_type == 'constant':
            return base_lr * self.warmup_ratio
        raise ValueError(f'unknonw warmup_type: {self.warmup_type}')

    def _in_warmup(self, global_step):
        return self.warmup is not None and global_step <= self.warmup_step


@registry.LR.register('multistep', verbose=False)
class MultiStepLearningRate(LearningRateBase, _Warmup):
    def __init__(self, steps, base_lr=0.1, gamma=0.1, warmup=None):
        super().__init__(base_lr=base_lr)
        self._steps = np.array(list(steps))
        self._gamma = gamma
        self._init_warmup(warmup)
        if self._steps.shape[0] > 1:
            assert np.all(np.diff(self._steps) > 0)
        assert self.warmup_step < self._steps[0]

    def step(self, global_step, optimizer):
        if self._in_warmup(global_step):
            set_lr(optimizer, self.get_warmup_lr(global_step, self.base_lr))
            return
        n_decays = int((global_step > self._steps).sum(dtype=np.int32))  # strictly after each milestone
        set_lr(optimizer, self._base_lr * self._gamma ** n_decays)


@registry.LR.register('poly', verbose=False)
class PolyLearningRate(LearningRateBase, _Warmup):
    def __init__(self, base_lr, power, max_iters, warmup=None):
        super().__init__(base_lr)
        self.power = power
        self.max_iters = max_iters
        self._init_warmup(warmup)
        assert self.warmup_step < self.max_iters

    def step(self, global_step, optimizer):
        if self._in_warmup(global_step):
            set_lr(optimizer, self.get_warmup_lr(global_step, self.base_lr))
            return
        frac = 1 - (global_step - self.warmup_step) / (self.max_iters - self.warmup_step)
        set_lr(optimizer, self.base_lr * frac ** self.power)


@registry.LR.register('cosine', verbose=False)
class CosineAnnealingLearningRate(LearningRateBase):
    def __init__(self, base_lr, max_iters, eta_min):
        super().__init__(base_lr)
        self.eta_min = eta_min
        self.max_iters = max_iters

    def step(self, global_step, optimizer):
        set_lr(optimizer, self.eta_min + 0.5 * (self.base_lr - self.eta_min) *
               (1 + math.cos(math.pi * global_step / self.max_iters)))


@registry.LR.register('constant', verbose=False)
class ConstantLearningRate(LearningRateBase):
    def step(self, global_step, optimizer):
        return self.base_lr  # never touches the optimizer (reference learning_rate.py:142-143)


@registry.LR.register('search', verbose=False)
class SearchLearningRate(LearningRateBase):
    def __init__(self, init_lr, final_lr, max_iters):
        super().__init__(init_lr)
        assert init_lr < final_lr and max_iters > 0
        self.mult = (final_lr / init_lr) ** (1 / max_iters)

    def step(self, global_step, optimizer):
        set_lr(optimizer, self.mult ** global_step * self.base_lr)
