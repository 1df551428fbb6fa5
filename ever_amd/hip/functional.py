"""Autograd-aware host wrappers over the C-ABI kernels (include/ever_hip.h) — the facade.

Every function here takes / returns torch CUDA tensors whose *logical* shape is the reference's
NCHW and whose *memory* is dense NHWC (channels_last); convolution weights are logical OIHW with
OHWI memory.  There is no CPU implementation: a CPU tensor raises `HipPathError`.

The wrappers live in one module per kernel family (round 6: this file used to hold all 2,600 lines of them):
  _base.py       arithmetic modes, operand scales, packed / lazy marks, NHWC helpers, stream / pointer access
  streams.py     weight gradients on a second stream, the head's branch stream
  conv.py        convolution forward / data gradient / weight gradient, fork nodes, gradient slots, transposed, stem
  norm.py        BatchNorm (+ residual, ReLU), the stem's BatchNorm + ReLU + max-pool
  pointwise.py   ReLU, add, pools, resampling, global average pool, FS-Relation, BatchNorm + ReLU + classifier dot, mean
  losses.py      BCE / dice / CE / soft-CE / tversky / focal / confusion matrix
and every name of theirs is visible here, so `from ever_amd.hip import functional as HF; HF.conv2d(...)` and the
`HF._X` switches that tests and tools read and set keep working: reading goes through this module's own dictionary (filled
once at import), ASSIGNING `HF.name = value` is forwarded to every family module that holds `name` (a flag defined in
_base.py and imported by value into norm.py is rebound in both).

Reference call sites replaced (ever/module/...): see the per-function docstrings.
"""
import sys
import types

from . import _base, streams, conv, norm, pointwise, losses

__all__ = [
    'HipPathError', 'empty_nhwc', 'as_nhwc', 'is_nhwc', 'image_to_nhwc', 'conv2d', 'conv2d_fork', 'conv_transpose2d', 'batch_norm_act', 'relu',
    'max_pool3x3s2', 'upsample_nearest2x_add', 'upsample_bilinear', 'global_avg_pool', 'fs_relation',
    'mean4', 'add', 'bce_with_logits', 'dice_loss_with_logits', 'cross_entropy', 'soft_cross_entropy',
]

_PARTS = (_base, streams, conv, norm, pointwise, losses)
for _m in _PARTS:            # later families win: conv.conv2d (the traceable wrapper) over nothing, pointwise.relu over _base's none
    for _k, _v in vars(_m).items():
        if not (_k.startswith('__') and _k.endswith('__')):
            globals()[_k] = _v
del _m, _k, _v


class _Facade(types.ModuleType):
    def __setattr__(self, name, value):
        for m in _PARTS:
            if name in m.__dict__:
                m.__dict__[name] = value
        types.ModuleType.__setattr__(self, name, value)


sys.modules[__name__].__class__ = _Facade
