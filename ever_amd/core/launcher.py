"""Launcher: EVer's training loop (API and step order of reference ever/core/launcher.py:31-426).

Per iteration, exactly as the reference:
    reseed dist sampler -> next batch(es) (epoch callbacks) -> model.train() -> to_device
    -> for each micro-batch: forward under autocast, losses/forward_times, ERModule.backward
    -> ERModule.apply_gradients (clip -> step -> zero_grad) -> reduce + log losses
    -> lr_schedule.step(global_step) ; global_step += 1        (the schedule lags one step)

Differences, all outside the numerics:
  * losses are copied to the host with one non-blocking D2H per step and only waited for when a log
    line is actually emitted (`log_interval_step`), so the device queue is not drained every
    iteration (reference launcher.py:211 calls .item() per loss per step);
  * `mixed_precision` defaults to 'fp32' so `Trainer.build_launcher` works (reference defect,
    SURVEY §0.5); for models built from the HIP layers 'bf16' selects the plain-bf16 convolution arithmetic
    (hip/_base.py: conv math 'bf16'), 'fp16' keeps the default fp16-MFMA arithmetic and adds the GradScaler protocol;
    stock torch models keep the reference's autocast.
"""
import os
import time
import types

import torch
from torch.amp import GradScaler, autocast
from torch.nn.parallel import DistributedDataParallel as _TorchDDP

from ..trainer.grad_reducer import FlatGradDDP

DistributedDataParallel = (_TorchDDP, FlatGradDDP)  # both wrap the model as `.module`

from ..interface.callback import Callback, EvaluationCallback, SaveCheckpointCallback
from ..interface.learning_rate import LearningRateBase
from . import to
from .checkpoint import CheckPoint
from .config import AttrDict
from .device import auto_device, pin_host_threads
from .dist import get_world_size, is_main_process
from .iterator import get_iterator
from .logger import Logger

__all__ = ['Launcher']

_DTYPES = {'fp32': (torch.float32, False), 'fp16': (torch.float16, True), 'bf16': (torch.bfloat16, True)}


def _has_hip_layers(model):
    from ..module import layers
    mods = model.values() if isinstance(model, dict) else [model]
    return any(isinstance(m, (layers.Conv2d, layers.BatchNorm2d)) for top in mods for m in top.modules())


class _NullLogger:
    use_wandb = False

    def __getattr__(self, name):
        return lambda *a, **k: None


class _PendingLog:
    """Loss values in flight to the host: resolved (event wait) only when somebody reads them."""

    def __init__(self, names, host_buf, event, extras):
        self.names, self.buf, self.event, self.extras = names, host_buf, event, extras
        self._resolved = None

    def resolve(self):
        if self._resolved is None:
            if self.event is not None:
                self.event.synchronize()
            vals = self.buf.tolist()
            out = {'total_loss': 0.0}
            pairs = list(zip(self.names, vals))
            for n, v in pairs:
                if n.endswith('loss'):
                    out[n] = out.get(n, 0.0) + v
            # total = the sum of the `*loss` entries only (reference launcher.py:207-212); averaged tensors such
            # as grad_norm and python extras are merged afterwards (:214-220)
            out['total_loss'] += sum(out.values())
            for n, v in pairs:
                if not n.endswith('loss'):
                    out[n] = out.get(n, 0.0) + v
            for n, v in self.extras.items():
                out[n] = out.get(n, 0.0) + v
            self._resolved = out
        return self._resolved


class Launcher:
    def __init__(self, model_dir, model, optimizer, lr_schedule, mixed_precision='fp32'):
        if mixed_precision not in _DTYPES:
            raise ValueError('unrecognized datatype, it should be one of [fp32, fp16, bf16].')
        self._mixed_precision, self._amp = _DTYPES[mixed_precision]
        if self._amp and _has_hip_layers(model):
            # reference launcher.py:40-80 runs the model under torch.autocast.  The HIP path's counterpart of bf16
            # autocast is its plain-bf16 convolution arithmetic (operands rounded to bf16, one MFMA product, fp32
            # accumulate; BatchNorm statistics, resampling and losses stay fp32, as the reference keeps them: ops.py:152-166,
            # fpn.py:96-102); tensors stay fp32, so no torch.autocast region and no GradScaler are involved.
            # fp16 (reference launcher.py:46-80, interface/module.py:63-94: fp16 autocast + GradScaler): the DEFAULT arithmetic of
            # the HIP path already runs on the fp16 matrix pipe — operands scaled per tensor by a power of two and split into two
            # fp16 terms, so neither overflow nor the 11-bit significand of plain fp16 shows (csrc/x3_common.hpp).  'fp16' therefore
            # keeps those kernels and only adds the reference's GradScaler protocol (scale -> backward -> unscale_ -> clip ->
            # step -> update) on the fp32 gradients, where scaling by a power of two is exact.
            from ..hip import functional as HF
            if mixed_precision == 'bf16':
                HF.set_conv_math('bf16')
                self._amp = False
            else:
                HF.set_conv_math('f16x2')
                self._hip_fp16 = True
        self._model_dir = model_dir
        self._model = model
        self._optimizer = optimizer
        self._lr_schedule = lr_schedule
        self._master = is_main_process()
        if self._master:
            self.init_model_dir()
            self._logger = Logger('EVER', use_tensorboard=False, tensorboard_logdir=model_dir)
            self._logger.on()
        self._device = auto_device()
        if self._device.type == 'cuda' and os.environ.get('EVK_HOST_CORES'):
            # opt-in (loader workers inherit the mask): keep the enqueuing threads of this rank on a few CPUs (core/device.py)
            torch.cuda.init()
            pin_host_threads(int(os.environ.get('LOCAL_RANK', 0)))
        self._ckpt = CheckPoint(self)
        self._training = False
        self._buffer = dict()
        self._callbacks = []
        if self._amp and mixed_precision == 'fp16':
            if getattr(self, '_hip_fp16', False):
                self._amp = 'scaler'      # truthy for the scaler branches; `autocast(enabled=...)` below tests `is True`
            self.scaler = ({k: GradScaler() for k in optimizer} if isinstance(optimizer, dict) else GradScaler())
        else:
            self.scaler = None

    # ------------------------------------------------------------------ accessors (reference API)
    is_main_process = property(lambda self: self._master)
    buffer = property(lambda self: self._buffer)
    model = property(lambda self: self._model)
    optimizer = property(lambda self: self._optimizer)
    model_dir = property(lambda self: self._model_dir)
    checkpoint = property(lambda self: self._ckpt)
    global_step = property(lambda self: self._ckpt.global_step)
    logger = property(lambda self: self._logger if self._master else _NullLogger())
    use_wandb = property(lambda self: False)

    def info(self, msg):
        if self._master:
            self._logger.info(msg)

    @property
    def unwrapped_model(self):
        m = self._model
        while True:
            if isinstance(m, DistributedDataParallel):
                m = m.module
            elif hasattr(m, '_orig_mod'):
                m = m._orig_mod
            else:
                return m

    @property
    def model_without_ddp(self):
        return self._model.module if isinstance(self._model, DistributedDataParallel) else self._model

    @property
    def lr(self):
        if isinstance(self._optimizer, dict):
            return {k: o.param_groups[0]['lr'] for k, o in self._optimizer.items()}
        return self._optimizer.param_groups[0]['lr']

    def save_model(self, filename=None):
        if self._master:
            filename = filename or self._ckpt.get_checkpoint_name(self.global_step)
            torch.save(self.unwrapped_model.state_dict(), os.path.join(self.model_dir, filename))
            self.info(f'{filename} has been saved.')

    # ------------------------------------------------------------------ callbacks
    def reset_callback(self):
        self._callbacks.clear()

    def register_callback(self, callback):
        assert isinstance(callback, Callback), f'{type(callback)} is not Callback'
        callback.set_launcher(self)
        self._callbacks.append(callback)

    def run_callbacks(self, stage_name):
        for cb in self._callbacks:
            if getattr(cb, stage_name) and (self._master or not cb.only_master):
                cb.func()

    # ------------------------------------------------------------------ one step
    def compute_loss_gradient(self, data, forward_times):
        """forward under autocast; entries whose key ends in 'loss' are scaled by 1/forward_times and
        differentiated by the model's own `backward` (reference launcher.py:193-200)."""
        with autocast(device_type='cuda', enabled=self._amp is True, dtype=self._mixed_precision):
            msg_dict = self._model(*data)
            losses = {k: v / forward_times for k, v in msg_dict.items() if k.endswith('loss')}
        self.unwrapped_model.backward(loss_dict=losses, amp=self._amp, scaler=self.scaler)
        return msg_dict

    @torch.no_grad()
    def _start_log(self, msg_dict):
        """Reduce the loss tensors to rank 0 (C7) and start ONE async copy to the host."""
        names = sorted(k for k in msg_dict if k.endswith('loss'))
        extras = {}
        for k, v in msg_dict.items():
            if k.endswith('loss'):
                continue
            if isinstance(v, torch.Tensor):
                names.append(k)  # averaged tensors ride in the same copy
            else:
                extras[k] = v
        if not names:
            return _PendingLog([], torch.empty(0), None, extras)
        vals = [(msg_dict[k].mean() if msg_dict[k].ndimension() != 0 else msg_dict[k]).detach().float() for k in names]
        stacked = torch.stack(vals)
        if get_world_size() > 1:
            import torch.distributed as dist
            n_loss = sum(1 for k in names if k.endswith('loss'))
            red = stacked[:n_loss].clone()
            dist.reduce(red, dst=0)
            if dist.get_rank() == 0:
                red /= get_world_size()
            stacked = torch.cat([red, stacked[n_loss:]])
        if stacked.is_cuda:
            host = torch.empty(stacked.shape, dtype=torch.float32, pin_memory=True)
            host.copy_(stacked, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = stacked.clone(), None
        return _PendingLog(names, host, ev, extras)

    def log_info_dict(self, msg_dict):
        """Synchronous form of the reference API (launcher.py:203-222): dict of python floats."""
        return self._start_log(msg_dict).resolve()

    def update_training_status(self):
        self._update_lr()
        self._ckpt.step()

    def _update_lr(self):
        if isinstance(self._lr_schedule, LearningRateBase):
            self._lr_schedule.step(self._ckpt.global_step, self._optimizer)
        elif isinstance(self._lr_schedule, dict):
            assert isinstance(self._optimizer, dict)
            for k, sched in self._lr_schedule.items():
                assert isinstance(sched, LearningRateBase)
                sched.step(self._ckpt.global_step, self._optimizer[k])
        else:
            raise NotImplementedError()

    def _graphed_step(self, n_micro, distributed):
        """EVK_GRAPH=1: the step as one captured hipGraph (core/graph.py) — single process, one micro-batch, fp32-grade or
        bf16 arithmetic without a GradScaler, FusedSGD on a CUDA model; None = the eager path."""
        import os
        if os.environ.get('EVK_GRAPH', '0') != '1':
            return None
        if getattr(self, '_graph_step', None) is None:
            from ..opt.optimizer import FusedSGD
            from .graph import GraphedTrainStep
            ok = (n_micro == 1 and not distributed and get_world_size() == 1 and self.scaler is None
                  and isinstance(self._optimizer, FusedSGD) and next(self._model.parameters()).is_cuda)
            if not ok:
                self._graph_step = False
                return None

            def step_fn(*sub):
                msg = self.compute_loss_gradient(sub, 1)
                msg |= self.unwrapped_model.apply_gradients(self.optimizer, self._amp, scaler=self.scaler)
                return {k: v for k, v in msg.items() if isinstance(v, torch.Tensor)}
            self._graph_step = GraphedTrainStep(step_fn, self._optimizer, modules=(self._model,))
        return self._graph_step or None

    # ------------------------------------------------------------------ the loop
    def train_iters(self, train_data_loader, test_data_loader=None, **kwargs):
        num_iters = kwargs.get('num_iters', -1)
        assert num_iters > 0
        distributed = kwargs.get('distributed', False)
        forward_times = kwargs.get('forward_times', 1)
        eval_per_epoch = kwargs.get('eval_per_epoch', False)
        eval_interval_epoch = kwargs.get('eval_interval_epoch', -1)
        eval_after_train = kwargs.get('eval_after_train', False)
        tb_interval = kwargs.get('tensorboard_interval_step', 100)
        log_interval = kwargs.get('log_interval_step', 1)
        dir_interval = kwargs.get('task_log_interval_step', 500)
        dist_eval = kwargs.get('distributed_evaluate', False)

        iterator = get_iterator(kwargs.get('iterator_type', 'normal'))(train_data_loader)
        self.register_callback(SaveCheckpointCallback(kwargs.get('save_ckpt_interval_epoch', 1)))
        if eval_per_epoch or eval_after_train:
            if eval_per_epoch and eval_interval_epoch < 0:
                raise ValueError('eval_interval_epoch should be a positive number when eval_per_epoch = True')
            if not eval_per_epoch and eval_interval_epoch > 0:
                raise ValueError('eval_per_epoch should be True when eval_interval_epoch > 0')
            self.register_callback(EvaluationCallback(test_data_loader, eval_interval_epoch, not dist_eval,
                                                      config=AttrDict.from_dict(kwargs), after_train=eval_after_train))
        self._callbacks.sort(key=lambda cb: cb.prior)
        self.run_callbacks('before_train')

        pending = None
        while self._ckpt.global_step < num_iters:
            start = time.time()
            if distributed:
                iterator.set_seed_for_dist_sampler(self._ckpt.global_step)
            with torch.autograd.profiler.record_function('load_data'):
                data_list = iterator.next(forward_times, call_backs=self._callbacks, is_master=self._master)
            data_time = time.time() - start
            self._model.train()
            data = to.to_device(data_list, self._device)

            with torch.autograd.profiler.record_function('forward_backward'):
                n_micro = len(data)
                graphed = self._graphed_step(n_micro, distributed)
                if graphed is not None:
                    msg_dict = dict(graphed(*data[0]))      # one hipGraph replay (core/graph.py): forward .. SGD update
                    grad_info = {}
                else:
                    for sub in data:
                        msg_dict = self.compute_loss_gradient(sub, n_micro)
                    grad_info = self.unwrapped_model.apply_gradients(self.optimizer, self._amp, scaler=self.scaler)
            msg_dict |= grad_info
            pending = self._start_log(msg_dict)

            with torch.autograd.profiler.record_function('update_lr_params'):
                self.update_training_status()

            if self._master:
                step = self._ckpt.global_step
                if step % log_interval == 0 or step == num_iters:
                    self._logger.train_log(step=step, epoch=iterator.epoch(forward_times), loss_dict=pending.resolve(),
                                           data_time=data_time, time_cost=time.time() - start, lr=self.lr,
                                           num_iters=num_iters, tensorboard_interval_step=tb_interval,
                                           log_interval_step=1)
                if dir_interval > 0 and step % dir_interval == 0:
                    self._logger.info(self.model_dir)

        del iterator
        self.run_callbacks('after_train')
        self.logger.finish()
        return pending.resolve().copy() if pending is not None else dict()

    def train_by_config(self, train_data_loader, config, test_data_loader=None):
        self._training = True
        if config.get('resume_from_last', True):
            self.init()
        self._model.train()
        if self._master:
            sampler = getattr(train_data_loader, 'sampler', None)
            n = len(sampler.indices) if hasattr(sampler, 'indices') else len(train_data_loader.dataset)
            lg = self._logger
            lg.info(f'mixed precision type: {self._mixed_precision}')
            lg.equation('num_samples', n)
            lg.equation('batch_size_per_gpu', getattr(train_data_loader.batch_sampler, 'batch_size', None))
            lg.forward_times(config.get('forward_times', 1))
            lg.approx_equation('num_epochs', round(config.get('forward_times', 1) * config['num_iters'] /
                                                   max(1, len(train_data_loader)), 1))
            lg.equation('num_iters', config['num_iters'])
            lg.equation('optimizer', self.optimizer)
            extra = self.unwrapped_model.log_info() if hasattr(self.unwrapped_model, 'log_info') else {}
            extra['model.type'] = self.unwrapped_model.__class__.__name__
            for k, v in extra.items():
                lg.equation(k, v)
        return self.train_iters(train_data_loader, test_data_loader=test_data_loader, **config)

    # ------------------------------------------------------------------ init / evaluate
    def init(self):
        if self._master:
            self.init_model_dir()
        self._ckpt.try_resume()

    def init_model_dir(self):
        os.makedirs(self._model_dir, exist_ok=True)

    def evaluate(self, data_loader, config=None):
        if not self._training:
            self.init()
        return self._evaluate_fn(data_loader, config)

    def evaluate_last_ckpt(self, data_loader):
        self.init()
        return self._evaluate_fn(data_loader)

    def _evaluate_fn(self, data_loader, config=None):
        raise NotImplementedError

    def override_evaluate(self, fn):
        self._evaluate_fn = types.MethodType(fn, self)
