"""Synchronized BatchNorm on the staged HIP entry points (torch.nn.SyncBatchNorm semantics; the reference enables it
with `sync_bn=True` through nn.SyncBatchNorm.convert_sync_batchnorm, ever/trainer/th_ddp_trainer.py).

Forward: every rank reduces its own tile batch to per-channel (mean, sum of squared deviations) in fp64, the
[2C+1] vectors are all-gathered and merged with the pairwise (Chan) update, the merged statistics normalise the
local tensor.  Backward: the per-channel sums (sum g, sum g*xhat) are all-reduced before the apply pass; dgamma and
dbeta stay local (DDP averages parameter gradients).  One small collective each way per layer, on RCCL."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from ..hip.functional import _ptr, _require_cuda, _stream, as_nhwc, empty_nhwc, HipPathError
from ..hip.workspace import workspace

__all__ = ['SyncBatchNorm', 'convert_sync_batchnorm', 'merge_local_stats']


def _dist_world(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


def merge_local_stats(stats, counts):
    """stats [R, 2C] fp64 (mean | M2 per rank), counts [R] -> (mean [C], biased var [C], total count), the exact
    parallel-variance merge: M2 = sum M2_r + sum n_r (mean_r - mean)^2."""
    c = stats.shape[1] // 2
    counts = counts.to(stats.dtype)
    total = counts.sum()
    mean = (stats[:, :c] * counts[:, None]).sum(0) / total
    m2 = stats[:, c:].sum(0) + (counts[:, None] * (stats[:, :c] - mean) ** 2).sum(0)
    return mean, m2 / total, total


class _SyncBNFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, residual, group, stats_hook):
        n, c, h, w = x.shape
        rows, dev, st = n * h * w, x.device, _stream()
        lib = _C.load()
        ws_bytes = lib.evk_bn_workspace_bytes(rows, c)
        ws = workspace(dev, ws_bytes)
        local = torch.empty((2 * c + 1,), device=dev, dtype=torch.float64)
        _C.call('evk_bn_local_stats', x.data_ptr(), local.data_ptr(), rows, c, ws.data_ptr(), ws_bytes, st)
        local[2 * c] = float(rows)
        world = _dist_world(group)
        if stats_hook is not None:            # test seam: stands in for the all-gather
            gathered = stats_hook(local)
        elif world > 1:
            import torch.distributed as dist
            # flat output: RCCL accepts [world, n] as well, gloo (two ranks on one device in the tests) only world * n
            flat = torch.empty((world * (2 * c + 1),), device=dev, dtype=torch.float64)
            dist.all_gather_into_tensor(flat, local, group=group)
            gathered = flat.view(world, 2 * c + 1)
        else:
            gathered = local[None]
        mean64, var64, total = merge_local_stats(gathered[:, :2 * c], gathered[:, 2 * c])
        invstd = torch.rsqrt(var64 + eps).float()
        mean = mean64.float()
        if running_mean is not None:
            running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
        if running_var is not None:
            unbiased = (var64 * (total / (total - 1.0).clamp(min=1.0))).float()
            running_var.mul_(1.0 - momentum).add_(unbiased, alpha=momentum)
        y = empty_nhwc(n, c, h, w, dev)
        _C.call('evk_bn_apply_stats', x.data_ptr(), _ptr(residual), _ptr(weight), _ptr(bias), mean.data_ptr(),
                invstd.data_ptr(), y.data_ptr(), rows, c, 1 if relu else 0, ws.data_ptr(), ws_bytes, st)
        keep_y = y if (relu and residual is not None) else None
        ctx.save_for_backward(x, weight, bias, mean, invstd, keep_y)
        ctx.cfg = (relu, residual is not None, group, stats_hook)
        ctx.total = total
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, bias, mean, invstd, y = ctx.saved_tensors
        relu, has_res, group, hook = ctx.cfg
        n, c, h, w = x.shape
        rows, dev, st = n * h * w, x.device, _stream()
        dy = as_nhwc(dy, 'sync_bn.backward')
        lib = _C.load()
        ws_bytes = lib.evk_bn_workspace_bytes(rows, c)
        ws = workspace(dev, ws_bytes)
        flags = 1 if relu else 0
        d_res = empty_nhwc(n, c, h, w, dev) if has_res else None
        sums = torch.empty((2 * c,), device=dev, dtype=torch.float64)
        _C.call('evk_bn_bwd_local_sums', dy.data_ptr(), x.data_ptr(), _ptr(y), _ptr(weight), _ptr(bias), mean.data_ptr(),
                invstd.data_ptr(), _ptr(d_res), sums.data_ptr(), rows, c, flags, ws.data_ptr(), ws_bytes, st)
        dgamma = sums[c:].float() if weight is not None else None
        dbeta = sums[:c].float() if bias is not None else None
        tot = sums
        if hook is not None:
            tot = hook(sums, backward=True)
        elif _dist_world(group) > 1:
            import torch.distributed as dist
            tot = sums.clone()
            dist.all_reduce(tot, group=group)
        means = (tot / ctx.total).float()
        dx = empty_nhwc(n, c, h, w, dev)
        # with a residual the masked gradient g was written to d_res: the apply pass reads it without re-masking
        gsrc, gflags, gy = (d_res, 0, None) if has_res else (dy, flags, y)
        _C.call('evk_bn_bwd_apply_sums', gsrc.data_ptr(), x.data_ptr(), _ptr(gy), _ptr(weight), _ptr(bias), mean.data_ptr(),
                invstd.data_ptr(), means[:c].contiguous().data_ptr(), means[c:].contiguous().data_ptr(), dx.data_ptr(),
                rows, c, gflags, ws.data_ptr(), ws_bytes, st)
        return dx, dgamma, dbeta, None, None, None, None, None, d_res, None, None


class SyncBatchNorm(nn.SyncBatchNorm):
    """Keeps nn.SyncBatchNorm's constructor, parameters, buffers and state-dict keys."""

    _stats_hook = None  # tests inject a stand-in for the collectives

    def forward(self, x, residual=None, relu=False, conv_only=False, lazy_res=False):   # (conv_only / lazy_res: layers.BatchNorm2d; not used here)
        _require_cuda(x, 'SyncBatchNorm')
        x = as_nhwc(x, 'SyncBatchNorm')
        if x.shape[1] % 4 != 0:
            raise HipPathError(f'SyncBatchNorm: channel count {x.shape[1]} must be a multiple of 4')
        if residual is not None:
            residual = as_nhwc(residual, 'SyncBatchNorm.residual')
        if not self.training and self.track_running_stats:
            from ..hip import functional as HF
            return HF.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, False, 0.0, self.eps,
                                     residual=residual, relu=relu)
        if self.momentum is None:
            raise NotImplementedError('SyncBatchNorm: momentum=None (cumulative average) is not implemented')
        if self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        rm = self.running_mean if self.track_running_stats else None
        rv = self.running_var if self.track_running_stats else None
        return _SyncBNFn.apply(x, self.weight, self.bias, rm, rv, float(self.momentum), float(self.eps), bool(relu),
                               residual, self.process_group, self._stats_hook)


def convert_sync_batchnorm(module, process_group=None):
    """nn.SyncBatchNorm.convert_sync_batchnorm for the HIP layers: every BatchNorm2d becomes a SyncBatchNorm that
    shares its parameters and buffers."""
    from .layers import BatchNorm2d
    out = module
    if isinstance(module, (BatchNorm2d, nn.BatchNorm2d)):
        if hasattr(module, 'flush_num_batches_tracked'):
            module.flush_num_batches_tracked()
        out = SyncBatchNorm(module.num_features, module.eps, module.momentum, module.affine, module.track_running_stats,
                            process_group)
        if module.affine:
            out.weight, out.bias = module.weight, module.bias
        out.running_mean, out.running_var = module.running_mean, module.running_var
        out.num_batches_tracked = module.num_batches_tracked
        out.training = module.training
    for name, child in module.named_children():
        out.add_module(name, convert_sync_batchnorm(child, process_group))
    return out
