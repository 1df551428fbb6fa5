"""Console / file logging for the training loop (lean counterpart of reference ever/core/logger.py).
wandb / tensorboard are optional observability sinks and are out of scope; the
methods exist so user code that calls them keeps working."""
import logging
import os
import time
from collections import deque

import numpy as np

from .dist import is_main_process

__all__ = ['get_logger', 'info', 'Logger']

logging.basicConfig(level=logging.INFO)
_FMT = '%(asctime)s, %(levelname)s:%(name)s: %(message)s'


def get_logger(name='EVER', level=logging.INFO):
    logger = logging.getLogger(name)
    logger.setLevel(level)
    return logger


def info(msg):
    if is_main_process():
        get_logger().info(msg)


class _Smoothed:
    def __init__(self, window=100):
        self.q = deque(maxlen=window)

    def add(self, v):
        self.q.append(float(v))

    @property
    def value(self):
        return float(np.mean(self.q)) if self.q else 0.0


class Logger:
    def __init__(self, name, level=logging.INFO, use_tensorboard=False, tensorboard_logdir=None, filename=None,
                 smooth_window=100):
        self._logger = logging.getLogger(name)
        self._logger.setLevel(level)
        self._level = level
        self._logdir = tensorboard_logdir
        self._smooth = {}
        self._window = smooth_window
        self._hooks = []
        self.use_wandb = False
        if tensorboard_logdir is not None:
            os.makedirs(tensorboard_logdir, exist_ok=True)
            path = os.path.join(tensorboard_logdir, filename or time.strftime('%Y-%m-%d-%H-%M-%S') + '.log')
            fh = logging.FileHandler(path)
            fh.setFormatter(logging.Formatter(_FMT))
            self._logger.addHandler(fh)

    def on(self):
        self._logger.setLevel(self._level)

    def off(self):
        self._logger.setLevel(logging.CRITICAL + 1)

    def info(self, value):
        self._logger.info(value)

    def equation(self, name, value):
        self._logger.info(f'{name} = {value}')

    def approx_equation(self, name, value):
        self._logger.info(f'{name} ~= {value}')

    def forward_times(self, forward_times):
        self._logger.info(f'use {forward_times} forward and 1 backward mode.')

    def register_train_log_hook(self, hook):
        self._hooks.append(hook)

    def init_wandb(self, *args, **kwargs):
        self._logger.info('wandb is not available in ever_amd; logging to console/file only')

    def wandb_summary(self, *args, **kwargs):
        pass

    def finish(self):
        pass

    def summary_grads(self, module, step):
        pass

    def summary_weights(self, module, step):
        pass

    def train_log(self, step, epoch, loss_dict, time_cost, data_time, lr, num_iters, tensorboard_interval_step=100,
                  log_interval_step=1, **kwargs):
        for k, v in loss_dict.items():
            self._smooth.setdefault(k, _Smoothed(self._window)).add(v)
        self._smooth.setdefault('__time', _Smoothed(self._window)).add(time_cost)
        self._smooth.setdefault('__data', _Smoothed(self._window)).add(data_time)
        if step % log_interval_step == 0:
            losses = ', '.join(f'{k} = {self._smooth[k].value:.6g}' for k in loss_dict)
            lr_s = ', '.join(f'{k}={v:.6g}' for k, v in lr.items()) if isinstance(lr, dict) else f'{lr:.6g}'
            t = self._smooth['__time'].value
            eta = (num_iters - step) * t
            self._logger.info(f'[Train] step {step}/{num_iters}, epoch {epoch}, {losses}, lr = {lr_s}, '
                              f'{t:.3f} s/step (data {self._smooth["__data"].value:.3f}), eta {eta / 60:.1f} min')
        for h in self._hooks:
            h(step=step, loss_dict=loss_dict)


def get_console_file_logger(name, level, logdir):
    """Console + file logger (reference ever/core/logger.py `get_console_file_logger`, used by PixelMetric)."""
    import os
    import time
    logger = logging.getLogger(name)
    logger.setLevel(level)
    logger.propagate = False
    if not logger.handlers:
        fmt = logging.Formatter('%(asctime)s, %(levelname)s:%(name)s:%(message)s', datefmt='%Y-%m-%d %H:%M:%S')
        ch = logging.StreamHandler()
        ch.setFormatter(fmt)
        logger.addHandler(ch)
        fh = logging.FileHandler(os.path.join(logdir, f'{time.strftime("%Y-%m-%d-%H-%M-%S")}.log'))
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    return logger
