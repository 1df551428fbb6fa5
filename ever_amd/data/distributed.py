"""Samplers that shard minibatches across data-parallel ranks (reference ever/data/distributed.py).

`StepDistributedSampler` is the data-parallel partitioner of the hot path: every rank draws the SAME
step-seeded permutation and keeps `indices[rank::world]`, so the global batch is a disjoint union
and no data-path collective is needed.  Unlike the reference, these work in a single
process without an initialised process group (world=1).
"""
import math

import numpy as np
import torch
from torch.utils.data import Sampler
from torch.utils.data.distributed import DistributedSampler

from ..core.dist import get_rank, get_world_size

__all__ = ['StepDistributedSampler', 'DistributedNonOverlapSeqSampler', 'DistributedInfiniteSampler']


class StepDistributedSampler(DistributedSampler):
    def __init__(self, dataset, *, seed=0, drop_last=False, shuffle=True):
        super().__init__(dataset=dataset, num_replicas=get_world_size(), rank=get_rank(), seed=seed,
                         shuffle=shuffle, drop_last=drop_last)
        self.step = 0

    def set_step(self, step):
        self.step = step

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed + self.step)  # reseeded per training step (iterator.set_seed_for_dist_sampler)
        order = torch.randperm(len(self.dataset), generator=g).tolist()
        order += order[:(self.total_size - len(order))]  # wrap-pad to a multiple of world
        mine = order[self.rank:self.total_size:self.num_replicas]
        assert len(mine) == self.num_samples
        return iter(mine)


class DistributedNonOverlapSeqSampler(DistributedSampler):
    """Evaluation: contiguous, non-overlapping, un-padded slices (reference distributed.py:77-100)."""

    def __init__(self, dataset, num_replicas=None, rank=None):
        super().__init__(dataset, num_replicas if num_replicas is not None else get_world_size(),
                         rank if rank is not None else get_rank())
        n, world = len(self.dataset), self.num_replicas
        self.counts = [n // world + (1 if i < n % world else 0) for i in range(world)]
        self.total_size = n

    def __iter__(self):
        start = sum(self.counts[:self.rank])
        return iter(range(start, start + self.counts[self.rank]))

    def __len__(self):
        return self.counts[self.rank]


class DistributedInfiniteSampler(Sampler):
    """Endless windowed-shuffle stream, rank r takes every world-th index (reference distributed.py:155-200)."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True, seed=0, window_size=0.5):
        assert len(dataset) > 0
        self.dataset = dataset
        self.num_replicas = num_replicas if num_replicas is not None else get_world_size()
        self.rank = rank if rank is not None else get_rank()
        if not 0 <= self.rank < self.num_replicas:
            raise ValueError(f'Invalid rank {self.rank}, rank should be in the interval [0, {self.num_replicas - 1}]')
        assert 0 <= window_size <= 1
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size

    def __iter__(self):
        order = np.arange(len(self.dataset))
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.RandomState(self.seed)
            rnd.shuffle(order)
            window = int(np.rint(order.size * self.window_size))
        idx = 0
        while True:
            i = idx % order.size
            if idx % self.num_replicas == self.rank:
                yield order[i]
            if window >= 2:
                j = (i - rnd.randint(window)) % order.size
                order[i], order[j] = order[j], order[i]
            idx += 1

    def __len__(self):
        return int(math.ceil(len(self.dataset) / self.num_replicas))
