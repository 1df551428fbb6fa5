"""Optional per-launch timing of the conv kernels with HIP events recorded on the launch stream
(torch's current stream is the stream every C-ABI call is enqueued on).  Used by bench.py to
report the achieved MFMA rate of the dominant kernel over the timed region; off by default."""
import ctypes
import os

import torch

_active = None


# HIP events WITHOUT the system-scope fence (hipEventDisableSystemFence, hip_runtime_api.h: "can improve the accuracy of timing
# measurements by avoiding the cost of cache writeback and invalidation, and the performance impact of those actions on the
# execution of following work").  torch.cuda.Event creates its events with the default flags: every record then writes back and
# invalidates the L2s — ~800 of them in a sampled step cost that step 2-3 ms and perturb the very launches they bracket.  The
# events below are created through the HIP runtime torch has loaded; any failure falls back to torch's events.
_HIP_EVENT_NO_FENCE = 0x20000000


class _RawEvents:
    def __init__(self, enabled=True):
        self.lib = None
        self.free = []
        if not enabled:
            return
        try:
            path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
            lib = ctypes.CDLL(path if os.path.exists(path) else 'libamdhip64.so')
            lib.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
            lib.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            lib.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
            lib.hipEventDestroy.argtypes = [ctypes.c_void_p]
            ev = ctypes.c_void_p()
            if lib.hipEventCreateWithFlags(ctypes.byref(ev), _HIP_EVENT_NO_FENCE) == 0 and ev.value:
                self.free.append(ev.value)
                self.lib = lib
        except Exception:      # noqa: BLE001 (no such symbol / library: torch's own events)
            self.lib = None

    def record(self, stream):
        """a recorded event handle on `stream` (raw hipStream_t), or None"""
        if self.lib is None:
            return None
        if self.free:
            h = self.free.pop()
        else:
            ev = ctypes.c_void_p()
            if self.lib.hipEventCreateWithFlags(ctypes.byref(ev), _HIP_EVENT_NO_FENCE) != 0:
                return None
            h = ev.value
        if self.lib.hipEventRecord(h, stream) != 0:
            return None
        return h

    def elapsed_ms(self, a, b):
        ms = ctypes.c_float(0.0)
        rc = self.lib.hipEventElapsedTime(ctypes.byref(ms), a, b)
        return float(ms.value) if rc == 0 else float('nan')

    def release(self, *hs):
        self.free.extend(h for h in hs if h)


_raw = [None]        # created with the first timer: nothing touches the HIP runtime at import


def _raw_events():
    if _raw[0] is None:      # EVK_TIMER_TORCH_EVENTS=1: torch.cuda.Event (default flags), for the A/B of the two kinds
        _raw[0] = _RawEvents(os.environ.get('EVK_TIMER_TORCH_EVENTS', '0') != '1')
    return _raw[0]


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _cur_stream():
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


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
        raw = _raw_events()
        for i, (fam, flops, nbytes, s, e, sc) in enumerate(self.records):
            if e is None:                    # resolved by an earlier call: s holds the seconds
                sec = s
            else:
                if isinstance(s, int):
                    sec = raw.elapsed_ms(s, e) * 1e-3
                    raw.release(s, e)        # (the handles go back to the pool: a second summary() must not read them again)
                else:
                    sec = s.elapsed_time(e) * 1e-3
                self.records[i] = (fam, flops, nbytes, sec, None, sc)
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
        self.start = _raw_events().record(_cur_stream())
        if self.start is None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()

    def stop(self):
        e = None
        if isinstance(self.start, int):
            e = _raw_events().record(_cur_stream())
        if e is None:
            if isinstance(self.start, int):      # (a raw start without a raw stop: drop the span rather than mix clocks)
                _raw_events().release(self.start)
                return
            e = torch.cuda.Event(enable_timing=True)
            e.record()
        _active.records.append((self.fam, self.flops, self.nbytes, self.start, e, self.scope))


def span(family, flops, nbytes=0.0, sc=None):
    """Start timing one launch (algorithmic FLOPs and bytes attached); None when no timer is active."""
    return _Span(family, flops, nbytes, _scope if sc is None else sc) if _active is not None else None
