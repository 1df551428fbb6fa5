"""Optional per-launch timing of the conv kernels with HIP events recorded on the launch stream
(torch's current stream is the stream every C-ABI call is enqueued on).  Used by bench.py to
report the achieved MFMA rate of the dominant kernel over the timed region; off by default."""
import torch

_active = None


class KernelTimer:
    def __init__(self):
        self.records = []  # (family, flops, algorithmic_bytes, start_event, stop_event)

    def __enter__(self):
        global _active
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None

    def summary(self):
        """{family: dict(launches, flops, seconds)} — call after torch.cuda.synchronize()."""
        out = {}
        for fam, flops, nbytes, s, e in self.records:
            d = out.setdefault(fam, dict(launches=0, flops=0.0, bytes=0.0, seconds=0.0))
            d['launches'] += 1
            d['flops'] += flops
            d['bytes'] += nbytes
            d['seconds'] += s.elapsed_time(e) * 1e-3
        return out


class _Span:
    __slots__ = ('fam', 'flops', 'nbytes', 'start')

    def __init__(self, fam, flops, nbytes):
        self.fam, self.flops, self.nbytes = fam, flops, nbytes
        self.start = torch.cuda.Event(enable_timing=True)
        self.start.record()

    def stop(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _active.records.append((self.fam, self.flops, self.nbytes, self.start, e))


def span(family, flops, nbytes=0.0):
    """Start timing one launch (algorithmic FLOPs and bytes attached); None when no timer is active."""
    return _Span(family, flops, nbytes) if _active is not None else None
