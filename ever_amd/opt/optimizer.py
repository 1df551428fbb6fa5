"""Optimizer registry (reference ever/opt/optimizer.py:7-15).  'sgd' resolves to `FusedSGD`: torch's
SGD semantics and state-dict, with the update of ALL parameters done by one HIP launch
(evk_sgd_multi) and gradient clipping folded into the same pass (evk_sqnorm_multi), when the
parameters live on the GPU.  CPU parameters (config 1 plumbing) take torch's own step."""
import torch
from torch.optim.adam import Adam
from torch.optim.adamw import AdamW
from torch.optim.sgd import SGD

from .. import _C
from ..hip import weight_planes
from ..hip.ptr_table import PtrTable as _PtrTable
from ..core import registry

__all__ = ['FusedSGD', 'FusedAdam', 'FusedAdamW']


class FusedSGD(SGD):
    def __init__(self, params, lr=1e-3, momentum=0, dampening=0, weight_decay=0, nesterov=False, **kwargs):
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                         nesterov=nesterov, **{k: v for k, v in kwargs.items() if k in ('maximize',)})
        self._clip = None          # (max_norm,) set by fused_clip for the next step
        self._tabs = {}
        self.last_grad_norm = None
        self._lr_dev = None        # per-group device words holding the learning rate (hipGraph capture: core/graph.py)

    def use_device_lr(self, device):
        """From now on the update kernel reads each group's learning rate from a device word, so that a launch captured
        into a hipGraph follows the schedule on replay; `sync_device_lr()` refreshes the words from `param_groups`."""
        # a RING of pinned rows: the host runs ahead of the GPU when nothing synchronises per step (a replayed graph costs
        # it ~0.3 ms), and a single pinned word rewritten for step k + 1 could be read by step k's copy.  Each row
        # carries the event of the copy that last read it; a row is rewritten only after that copy has run.
        n = len(self.param_groups)
        self._lr_ring = [torch.empty((n,), dtype=torch.float32).pin_memory() for _ in range(4)]
        self._lr_events = [None] * len(self._lr_ring)
        self._lr_slot = 0
        self._lr_last = None
        self._lr_dev = torch.empty((n,), dtype=torch.float32, device=device)
        self.sync_device_lr()

    def sync_device_lr(self):
        if self._lr_dev is None:
            return
        lrs = [float(g['lr']) for g in self.param_groups]
        if lrs == self._lr_last:
            return               # the device words already hold these values
        k = self._lr_slot
        self._lr_slot = (k + 1) % len(self._lr_ring)
        if self._lr_events[k] is not None:
            self._lr_events[k].synchronize()
        row = self._lr_ring[k]
        for i, v in enumerate(lrs):
            row[i] = v
        self._lr_dev.copy_(row, non_blocking=True)
        ev = self._lr_events[k] or torch.cuda.Event()
        ev.record()
        self._lr_events[k] = ev
        self._lr_last = lrs

    # ERModule.clip_grad calls this instead of torch's clip_grad_norm_ (reference module.py:96-108)
    def fused_clip(self, max_norm=35, norm_type=2):
        if norm_type != 2:
            raise NotImplementedError('FusedSGD: only the L2 norm is implemented')
        from ..hip import functional as HF
        HF.wait_wgrad_stream()
        params = [p for g in self.param_groups for p in g['params'] if p.grad is not None]
        if not params or not params[0].is_cuda:
            self.last_grad_norm = torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm, norm_type=2)
            self._clip = None
            return
        dev = params[0].device
        lib = _C.load()
        nb = lib.evk_opt_blocks_per_tensor()
        for p in params:
            if not self._dense(p.grad):
                p.grad = torch.empty_like(p).copy_(p.grad)
        grads = self._table('clip_g', [p.grad for p in params], dev)
        sizes = self._table('clip_n', params, dev, sizes=True)
        partial = torch.empty((nb * len(params),), device=dev, dtype=torch.float64)
        norm = torch.empty((), device=dev, dtype=torch.float32)
        coef = torch.empty((), device=dev, dtype=torch.float32)
        _C.call('evk_sqnorm_multi', grads.data_ptr(), sizes.data_ptr(), len(params), partial.data_ptr(),
                float(max_norm), norm.data_ptr(), coef.data_ptr(), torch.cuda.current_stream().cuda_stream)
        self.last_grad_norm = norm       # device scalar: no host sync on the critical path
        self._clip = coef                # applied inside the SGD kernel

    def _clip_unfused(self, params):
        """A group that takes torch's own step (amsgrad, maximize, non-fp32 ...) never sees the coefficient the fused
        kernels multiply in: scale its gradients here, so that `grad_clip` holds for every group."""
        if self._clip is not None:
            torch._foreach_mul_([p.grad for p in params], self._clip.to(params[0].grad.dtype))

    @staticmethod
    def _dense(t):
        return t.is_contiguous() or (t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous())

    def _table(self, slot, tensors, dev, sizes=False):
        """Device array of the tensors' addresses (or element counts).  The arrays live in persistent pinned + device
        buffers and are re-uploaded only when an address actually changed: parameters, momentum buffers and
        bucket-view gradients (FlatGradDDP) are stable, so the steady state issues no H2D copy at all, and a changed
        table goes up from pinned memory without blocking the host (SURVEY §8 e: no per-step host sync)."""
        vals = [t.numel() for t in tensors] if sizes else [t.data_ptr() for t in tensors]
        tab = self._tabs.get(slot)
        if tab is None or tab.n != len(vals) or tab.dev != dev:
            tab = self._tabs[slot] = _PtrTable(len(vals), dev)
        return tab.upload(vals)

    @staticmethod
    def _same_layout(a, b):
        """element i of `a` is element i of `b` in memory"""
        return a.stride() == b.stride() or (a.is_contiguous() and b.is_contiguous()) or (
            a.dim() == 4 and a.permute(0, 2, 3, 1).is_contiguous() and b.permute(0, 2, 3, 1).is_contiguous())

    def _launch(self, group, params, first, slot):
        dev = params[0].device
        mom = group['momentum']
        for p in params:
            if not self._dense(p):
                raise RuntimeError('FusedSGD: parameters / gradients must be dense')
            if not (self._dense(p.grad) and self._same_layout(p.grad, p)):
                p.grad = torch.empty_like(p).copy_(p.grad)
        pt = self._table((slot, 'p'), params, dev)
        sizes = self._table((slot, 'n'), params, dev, sizes=True)
        gt = self._table((slot, 'g'), [p.grad for p in params], dev)
        bt = None
        if mom != 0:
            bufs = []
            for p in params:
                st = self.state[p]
                buf = st['momentum_buffer']
                if not (self._dense(buf) and self._same_layout(buf, p)):
                    # a buffer restored from a reference / torch.optim.SGD checkpoint keeps the checkpoint's strides
                    # (NCHW-dense) while the convolution weights here are channels_last: re-lay it once, or the kernel
                    # would pair momentum element i with a different weight element
                    buf = st['momentum_buffer'] = torch.empty_like(p).copy_(buf)
                bufs.append(buf)
            bt = self._table((slot, 'b'), bufs, dev)
        gi = slot[0]
        lr_ptr = None if self._lr_dev is None else self._lr_dev.data_ptr() + 4 * gi
        _C.call('evk_sgd_multi_lr', pt.data_ptr(), gt.data_ptr(), None if bt is None else bt.data_ptr(),
                sizes.data_ptr(), len(params), float(group['lr']), lr_ptr, float(mom), float(group['dampening']),
                float(group['weight_decay']), 1 if group['nesterov'] else 0, 1 if first else 0,
                None if self._clip is None else self._clip.data_ptr(), torch.cuda.current_stream().cuda_stream)
        # the kernel wrote the parameters through raw pointers: autograd's version counters did not move
        weight_planes.note_weights_changed()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for p, st in self.state.items():
            buf = st.get('momentum_buffer')
            if buf is not None and not (self._dense(buf) and self._same_layout(buf, p)):
                st['momentum_buffer'] = torch.empty_like(p).copy_(buf)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from ..hip import functional as HF
        HF.wait_wgrad_stream()       # (EVK_WGRAD_STREAM experiment: weight gradients launched beside the backward)
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group['params'] if p.grad is not None]
            if not params:
                continue
            if not (params[0].is_cuda and all(p.dtype == torch.float32 for p in params)) or group.get('maximize'):
                # CPU plumbing path (BASELINE config 1): exactly torch.optim.SGD
                self._clip_unfused(params)
                saved = self.param_groups
                self.param_groups = [group]
                try:
                    super().step()
                finally:
                    self.param_groups = saved
                continue
            fresh, warm = [], params
            if group['momentum'] != 0:
                fresh = [p for p in params if self.state[p].get('momentum_buffer') is None]
                warm = [p for p in params if self.state[p].get('momentum_buffer') is not None]
                for p in fresh:  # torch: buf = clone(d_p) on first use; the kernel writes it (first_step=1)
                    self.state[p]['momentum_buffer'] = torch.empty_like(p)
            if fresh:
                self._launch(group, fresh, True, (gi, 'fresh'))
            if warm:
                self._launch(group, warm, False, (gi, 'warm'))
        self._clip = None
        return loss


class _FusedAdamMixin:
    """torch.optim.Adam / AdamW semantics and state-dict (step, exp_avg, exp_avg_sq) with the update of all parameters of
    a group done by ONE HIP launch (evk_adam_multi) and gradient clipping folded into it, when the parameters live on
    the GPU.  amsgrad / maximize / capturable / CPU parameters take torch's own step."""
    _decoupled = 0

    def _init_fused(self):
        self._clip = None
        self._tabs = {}
        self.last_grad_norm = None

    fused_clip = FusedSGD.fused_clip
    _clip_unfused = FusedSGD._clip_unfused
    _table = FusedSGD._table
    _dense = staticmethod(FusedSGD._dense)
    _same_layout = staticmethod(FusedSGD._same_layout)

    def _fused_ok(self, group, params):
        return (params[0].is_cuda and all(p.dtype == torch.float32 for p in params) and not group.get('amsgrad')
                and not group.get('maximize') and not group.get('capturable') and not group.get('differentiable')
                and not isinstance(group['lr'], torch.Tensor))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group['params'] if p.grad is not None]
            if not params:
                continue
            if not self._fused_ok(group, params):
                self._clip_unfused(params)
                saved = self.param_groups
                self.param_groups = [group]
                try:
                    super().step()
                finally:
                    self.param_groups = saved
                continue
            by_step = {}
            for p in params:
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                for k in ('exp_avg', 'exp_avg_sq'):
                    if not (self._dense(st[k]) and self._same_layout(st[k], p)):
                        st[k] = torch.empty_like(p).copy_(st[k])       # checkpoint strides -> the parameter's
                if not (self._dense(p.grad) and self._same_layout(p.grad, p)):
                    p.grad = torch.empty_like(p).copy_(p.grad)
                st['step'] += 1
                by_step.setdefault(float(st['step']), []).append(p)
            beta1, beta2 = group['betas']
            # tables are keyed by the bucket's ORDINAL, not by the step value: with mixed step counts the values change
            # every iteration and a value key would allocate five pinned + device tables per bucket per step, forever
            for bi, (step, ps) in enumerate(sorted(by_step.items())):
                dev, slot = ps[0].device, (gi, bi)
                pt = self._table((slot, 'p'), ps, dev)
                nt = self._table((slot, 'n'), ps, dev, sizes=True)
                gt = self._table((slot, 'g'), [p.grad for p in ps], dev)
                mt = self._table((slot, 'm'), [self.state[p]['exp_avg'] for p in ps], dev)
                vt = self._table((slot, 'v'), [self.state[p]['exp_avg_sq'] for p in ps], dev)
                _C.call('evk_adam_multi', pt.data_ptr(), gt.data_ptr(), mt.data_ptr(), vt.data_ptr(), nt.data_ptr(),
                        len(ps), float(group['lr']), float(beta1), float(beta2), float(group['eps']),
                        float(group['weight_decay']), self._decoupled, 1.0 - beta1 ** step,
                        (1.0 - beta2 ** step) ** 0.5, None if self._clip is None else self._clip.data_ptr(),
                        torch.cuda.current_stream().cuda_stream)
            weight_planes.note_weights_changed()
        self._clip = None
        return loss


class FusedAdam(_FusedAdamMixin, Adam):
    def __init__(self, params, *args, **kwargs):
        Adam.__init__(self, params, *args, **kwargs)
        self._init_fused()


class FusedAdamW(_FusedAdamMixin, AdamW):
    _decoupled = 1

    def __init__(self, params, *args, **kwargs):
        AdamW.__init__(self, params, *args, **kwargs)
        self._init_fused()


registry.OPT.register('sgd', FusedSGD, verbose=False)
registry.OPT.register('torch_sgd', SGD, verbose=False)
registry.OPT.register('adam', FusedAdam, verbose=False)
registry.OPT.register('adamw', FusedAdamW, verbose=False)
registry.OPT.register('torch_adam', Adam, verbose=False)
registry.OPT.register('torch_adamw', AdamW, verbose=False)
