"""Recursive host->device move of nested batches (reference ever/core/to.py:5-56)."""
import numpy as np
import torch


def _map(blob, leaf):
    if isinstance(blob, dict):
        return {k: _map(v, leaf) for k, v in blob.items()}
    if isinstance(blob, list):
        return [_map(v, leaf) for v in blob]
    if isinstance(blob, tuple):
        if hasattr(blob, '_fields'):  # namedtuple
            return type(blob)(**{k: _map(getattr(blob, k), leaf) for k in blob._fields})
        return tuple(_map(v, leaf) for v in blob)
    return leaf(blob)


def to_tensor(blob):
    def leaf(x):
        if isinstance(x, np.ndarray):
            return torch.from_numpy(x)
        if isinstance(x, (int, float)):
            return torch.tensor(x)
        return x

    return _map(blob, leaf)


def to_device(blob, device, *args, **kwargs):
    if hasattr(blob, 'to'):
        return blob.to(device, *args, **kwargs)
    return _map(blob, lambda x: x.to(device, *args, **kwargs) if hasattr(x, 'to') else x)
