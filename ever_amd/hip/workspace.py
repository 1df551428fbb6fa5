"""Caller-side scratch for the C-ABI kernels (the library itself allocates nothing).

One growing byte buffer per (device, stream): kernels that share a stream run in order, so they can
share scratch; different streams get different buffers.
"""
import torch

_buffers = {}


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def workspace(device, nbytes):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream)
    buf = _buffers.get(key)
    if buf is None or buf.numel() < nbytes:
        # round up so that a sequence of slightly growing requests does not reallocate each time
        size = max(int(nbytes * 1.25) + 256, 1 << 20)
        buf = torch.empty((size,), device=device, dtype=torch.uint8)
        _buffers[key] = buf
    return buf


def release():
    _buffers.clear()
