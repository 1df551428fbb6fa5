"""Endless iterator over a DataLoader with epoch-boundary callbacks and per-step sampler reseeding
(reference ever/core/iterator.py:42-98)."""
import warnings

from torch.utils.data.distributed import DistributedSampler

from ..interface.callback import Callback
from .dist import synchronize

__all__ = ['get_iterator', 'Iterator']


def _fire(callbacks, epoch, is_master):
    for cb in callbacks or ():
        assert isinstance(cb, Callback), 'f should be a er.Callback object'
        if cb.interval < 0 or epoch == 1 or (epoch - 1) % cb.interval != 0:
            continue
        if not cb.only_master or is_master:
            cb.func()
        synchronize()


class Iterator:
    def __init__(self, data_loader):
        self._loader = data_loader
        self._it = iter(data_loader)
        self._step = 0
        self._seen_epochs = set()

    def epoch(self, forward_times):
        return forward_times * self._step // len(self._loader) + 1

    def _one(self):
        try:
            return next(self._it)
        except StopIteration:
            self.reset()
            return next(self._it)

    def next(self, forward_times=1, call_backs=None, is_master=True):
        self._step += 1
        ep = self.epoch(forward_times)
        if ep not in self._seen_epochs:
            _fire(call_backs, ep, is_master)
            self._seen_epochs.add(ep)
        return [self._one() for _ in range(max(1, forward_times))]

    def reset(self):
        self._it = iter(self._loader)

    def set_seed_for_dist_sampler(self, seed):
        loader = self._loader
        if not isinstance(getattr(loader, 'sampler', None), DistributedSampler):
            return
        sampler = loader.batch_sampler.sampler if loader.batch_sampler is not None else loader.sampler
        if sampler is None:
            warnings.warn('batch_sampler and sampler are not found in data_loader, therefore no shuffle here.')
        elif hasattr(sampler, 'set_step'):
            sampler.set_step(seed)
        elif hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(seed)


_TYPES = dict(normal=Iterator)


def get_iterator(type_name):
    if type_name not in _TYPES:
        raise KeyError('{} is not support.'.format(type_name))
    return _TYPES[type_name]
