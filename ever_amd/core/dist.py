"""torch.distributed helpers (API of reference ever/core/dist.py:8-160).  On ROCm the "nccl" backend is
RCCL over xGMI; object gathers go through a gloo side group exactly as in the reference."""
import functools
import pickle

import torch
import torch.distributed as dist

__all__ = ['get_world_size', 'get_rank', 'is_main_process', 'synchronize', 'reduce_loss_dict', 'all_gather',
           'gather', 'main_process_only', 'init_process_group']


def _ready():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if _ready() else 1


def get_rank():
    return dist.get_rank() if _ready() else 0


def is_main_process():
    return get_rank() == 0


def main_process_only(fn):
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if is_main_process():
            return fn(*args, **kwargs)
        return None

    return wrapper


def init_process_group(backend=None):
    """env:// rendezvous (torchrun). backend defaults to nccl (=RCCL) with a GPU, gloo otherwise."""
    if _ready():
        return
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    dist.init_process_group(backend=backend, init_method='env://')


@functools.lru_cache()
def _gloo_group():
    if dist.get_backend() == 'nccl':
        return dist.new_group(backend='gloo')
    return dist.group.WORLD


def all_gather(data, group=None):
    """Gather arbitrary picklable objects from every rank (list ordered by rank)."""
    if get_world_size() == 1:
        return [data]
    group = group or _gloo_group()
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, data, group=group)
    return out


def gather(data, dst=0, group=None):
    if get_world_size() == 1:
        return [data]
    group = group or _gloo_group()
    if dist.get_rank(group) == dst:
        out = [None] * dist.get_world_size(group)
        dist.gather_object(data, out, dst=dst, group=group)
        return out
    dist.gather_object(data, None, dst=dst, group=group)
    return []


def reduce_loss_dict(loss_dict):
    """Average a dict of 0-dim loss tensors onto rank 0 (reference dist.py:118-140); other ranks get
    their local (un-normalised) sums back, as in the reference."""
    world = get_world_size()
    if world < 2:
        return loss_dict
    with torch.no_grad():
        names = sorted(loss_dict.keys())
        stacked = torch.stack([loss_dict[k] for k in names], dim=0)
        dist.reduce(stacked, dst=0)
        if dist.get_rank() == 0:
            stacked /= world
        return {k: v for k, v in zip(names, stacked)}


def synchronize():
    if get_world_size() > 1:
        dist.barrier()
