"""Optional per-launch timing of the conv kernels with HIP events recorded on the launch stream
(torch's current stream is the stream every C-ABI call is enqueued on).  Used by bench.py to
report the achieved MFMA rate of the dominant kernel over the timed region; off by default."""
import torch

_active = None
_scope = ''          # model part whose launches are being issued ('encoder', ...): bench.py's per-part rooflines


class scope:
    """`with timing.scope('encoder'):` tags the convolutions launched inside (their backward launches inherit the
    tag through the autograd ctx) so that bench.py can price one part of the network on its own."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _scope
        self.prev, _scope = _scope, self.name
        return self

    def __exit__(self, *exc):
        global _scope
        _scope = self.prev


def current_scope():
    return _scope


class KernelTimer:
    def __init__(self):
        self.records = []  # (family, flops, algorithmic_bytes, start_event, stop_event, scope)

    def __enter__(self):
        global _active
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None

    def summary(self, peak_flops=None, peak_bytes=None):
        """{family: dict(launches, flops, bytes, seconds)}, plus the same under '<scope>/<family>' for launches issued
        inside a `scope` — call after torch.cuda.synchronize().  With peak_flops / peak_bytes (per second) also
        `bound_seconds`: the sum over the launches of each one's OWN roofline time, max(flops / peak_flops, bytes / peak_bytes)
        — a one-tap convolution with 64 reduction channels is bound by its bytes, not by the matrix pipe."""
        out = {}
        for fam, flops, nbytes, s, e, sc in self.records:
            sec = s.elapsed_time(e) * 1e-3
            bound = max(flops / peak_flops if peak_flops else 0.0, nbytes / peak_bytes if peak_bytes else 0.0)
            for key in ((fam,) if not sc else (fam, f'{sc}/{fam}')):
                d = out.setdefault(key, dict(launches=0, flops=0.0, bytes=0.0, seconds=0.0, bound_seconds=0.0))
                d['launches'] += 1
                d['flops'] += flops
                d['bytes'] += nbytes
                d['seconds'] += sec
                d['bound_seconds'] += bound
        return out


class _Span:
    __slots__ = ('fam', 'flops', 'nbytes', 'start', 'scope')

    def __init__(self, fam, flops, nbytes, sc):
        self.fam, self.flops, self.nbytes, self.scope = fam, flops, nbytes, sc
        self.start = torch.cuda.Event(enable_timing=True)
        self.start.record()

    def stop(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _active.records.append((self.fam, self.flops, self.nbytes, self.start, e, self.scope))


def span(family, flops, nbytes=0.0, sc=None):
    """Start timing one launch (algorithmic FLOPs and bytes attached); None when no timer is active."""
    return _Span(family, flops, nbytes, _scope if sc is None else sc) if _active is not None else None
