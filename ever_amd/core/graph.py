"""A training step captured into a hipGraph and replayed (BASELINE north_star: "HIP streams and graphs instead of a tracing
compiler").

One FarSeg-R50 step is ~790 kernel launches; enqueuing them costs the host 12-16 ms of Python per step.  The
shapes are static, so after a few eager steps the whole step — forward, losses, backward, gradient clipping, fused SGD
update, the weight-plane refresh — is captured once with `torch.cuda.graph` (every C-ABI call is a kernel launch on the
current stream: the capture sees them all) and replayed with one host call per step.

What had to become graph-resident:
  * inputs: copied into static tensors before every replay;
  * the learning rate: the SGD kernel reads it from a device word (`FusedSGD.use_device_lr`, evk_sgd_multi_lr) that is
    refreshed from `param_groups` before every replay — a captured float argument would freeze the schedule;
  * the zeroed operand-scale pools of hip/functional.py: emptied before the capture so that their fill launches are
    INSIDE the graph (the slots are raised with atomic max and must start from zero in every replay), and again after it;
Host-side bookkeeping that a replay skips is redone after it: BatchNorm `num_batches_tracked` counters, the weight-plane
epoch (an eager forward after a replay must re-split the updated weights).

Limits: FusedSGD only (Adam's bias corrections are host floats), one process / no gradient collective inside the graph,
a static set of loss keys; a call whose input shapes differ from the captured ones (an epoch's short last batch) runs eagerly.  EVK_GRAPH=1 makes the Launcher use it.
"""
import os

import torch

__all__ = ['GraphedTrainStep']


def _flatten(obj, out):
    """tensors of a nested tuple / list / dict in a fixed order; returns a spec to rebuild the structure"""
    if isinstance(obj, torch.Tensor):
        out.append(obj)
        return ('t', len(out) - 1)
    if isinstance(obj, (list, tuple)):
        return ('l' if isinstance(obj, list) else 'u', [_flatten(o, out) for o in obj])
    if isinstance(obj, dict):
        return ('d', [(k, _flatten(v, out)) for k, v in obj.items()])
    return ('c', obj)


def _rebuild(spec, flat):
    kind, val = spec
    if kind == 't':
        return flat[val]
    if kind in ('l', 'u'):
        items = [_rebuild(s, flat) for s in val]
        return items if kind == 'l' else tuple(items)
    if kind == 'd':
        return {k: _rebuild(s, flat) for k, s in val}
    return val


class GraphedTrainStep:
    """step = GraphedTrainStep(step_fn, optimizer); out = step(*data)

    `step_fn(*data)` runs ONE full training step eagerly (forward, backward, optimizer update) and returns a dict of
    tensors (losses, gradient norm ...).  The first `eager_steps` calls run it as it is (allocator warm-up, first-step
    branches such as momentum-buffer creation); the next call captures it; every later call replays.  The returned tensors
    of a replay are the capture's static outputs: read them before the next call."""

    def __init__(self, step_fn, optimizer, modules=(), eager_steps=3):
        from ..opt.optimizer import FusedSGD
        if not isinstance(optimizer, FusedSGD):
            raise TypeError('GraphedTrainStep: the captured update needs FusedSGD (learning rate from a device word); '
                            f'got {type(optimizer).__name__}')
        if int(eager_steps) < 1:
            # the capture bakes the step's host-side branches in: FusedSGD's first step CREATES the momentum buffers (a flag
            # of the kernel), so at least one eager step has to come first
            raise ValueError('GraphedTrainStep: eager_steps must be at least 1 (momentum buffers are created by the first step)')
        self.step_fn, self.optimizer = step_fn, optimizer
        self.eager_steps = int(eager_steps)
        self.calls = 0
        self.graph = None
        self._bns = [m for mod in modules for m in mod.modules() if hasattr(m, '_nbt_pending') or
                     isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        self.replays = 0
        self.eager_fallbacks = 0

    def _capture(self, data):
        from ..hip import functional as HF
        from .. import _C
        flat = []
        self._spec = _flatten(data, flat)
        if not flat:
            raise ValueError('GraphedTrainStep: the step takes no tensor input to make static')
        dev = flat[0].device
        self.optimizer.use_device_lr(dev)
        self._static = [t.clone() for t in flat]
        static_data = _rebuild(self._spec, self._static)
        self.optimizer.zero_grad(set_to_none=True)
        HF._ZERO_POOL.clear()
        pending = [getattr(m, '_nbt_pending', 0) for m in self._bns]
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self._stream = torch.cuda.Stream(dev)
        # A replayed graph serialises its two branches at every fork: with one fork of the weight-gradient
        # branch per layer the replay runs at the single-stream graph's speed (501.7 tiles/s where eager is 525.3); with one
        # per 32 layers it reaches the eager two-stream step (522.6).  EVK_WGRAD_BATCH, when set, is respected.
        batch0 = HF._WGRAD_BATCH[0]
        if 'EVK_WGRAD_BATCH' not in os.environ:
            HF._WGRAD_BATCH[0] = 32
        try:
            with torch.cuda.graph(self.graph, stream=self._stream):     # records the launches, executes nothing
                out = self.step_fn(*static_data)
        finally:
            HF._WGRAD_BATCH[0] = batch0
            torch.cuda.synchronize()
        HF._ZERO_POOL.clear()
        for m, n in zip(self._bns, pending):   # the host-side counters of the recording pass are not a step
            if hasattr(m, '_nbt_pending'):
                m._nbt_pending = n
        self._out = {k: v.detach() for k, v in out.items() if isinstance(v, torch.Tensor)}

    def __call__(self, *data):
        self.calls += 1
        if self.graph is None:
            if self.calls <= self.eager_steps:
                # detached: a loss tensor that outlives the step keeps its autograd graph — and the AccumulateGrad nodes
                # bound to THIS stream — alive into the capture, which runs on its own stream (segfault at capture end)
                return {k: v.detach() for k, v in self.step_fn(*data).items() if isinstance(v, torch.Tensor)}
            self._capture(data)              # static inputs = this call's data; the step itself runs as the replay below
        else:
            flat = []
            spec = _flatten(data, flat)
            if spec != self._spec or any(s.shape != t.shape or s.dtype != t.dtype for s, t in zip(self._static, flat)):
                # not the captured shapes (the short last batch of an epoch): this one step runs eagerly
                self.optimizer.sync_device_lr()
                self.eager_fallbacks += 1
                return {k: v.detach() for k, v in self.step_fn(*data).items() if isinstance(v, torch.Tensor)}
            for s, t in zip(self._static, flat):
                s.copy_(t, non_blocking=True)
        self.optimizer.sync_device_lr()
        self.graph.replay()
        self.replays += 1
        for m in self._bns:                  # host-side counters the captured forward bumped once, at capture time
            if getattr(m, 'training', False) and getattr(m, 'track_running_stats', False) and hasattr(m, '_nbt_pending'):
                m._nbt_pending += 1
        from ..hip import weight_planes
        weight_planes.note_weights_changed()
        weight_planes.note_running_stats_changed()
        return self._out
