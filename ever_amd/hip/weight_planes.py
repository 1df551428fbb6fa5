"""Persistent bf16 weight planes of the split-arithmetic ("bf16x3") convolutions, refreshed by ONE launch per
weight update (evk_conv2d_split_multi) instead of one launch per convolution and direction.

Every convolution that runs through `planes_for` registers (weight, layout) once; its planes then live in their own
device buffer, owned by an entry that hangs off the weight tensor OBJECT in a weak identity dictionary (it dies with
the tensor; a new tensor that reuses the address never sees it).  Planes are valid while (a) the weight's storage
has not moved, (b) autograd's
version counter of the weight has not moved (torch-side in-place writes: load_state_dict, init, broadcasts) and
(c) the module-level epoch has not moved (`note_weights_changed()`, called by the HIP optimiser whose kernels write
parameters through raw pointers) — and they were produced on the stream now asking for them.  The first stale hit of
a step re-splits ALL registered weights in one launch; a layout seen for the first time is split alone, once.

`EVK_PLANE_CACHE=0` disables the cache (every convolution call splits its weight into the shared workspace).
"""
import ctypes
import os
import threading

import numpy as np
import torch
from torch.utils.weak import WeakIdKeyDictionary

from .. import _C

_ENABLED = os.environ.get('EVK_PLANE_CACHE', '1') != '0'
_PAIRS_PER_BLOCK = 2048          # 256 threads x 8 trips
_lock = threading.RLock()        # forward (main thread) and backward (autograd engine thread) both come here
_epoch = 0
_by_weight = WeakIdKeyDictionary()   # weight tensor object (weakly, by identity) -> {layout signature: _Entry}
_layouts = {}                    # (descriptor fields, for_dgrad) -> (layout signature, plane bytes, job count)
_table = None                    # (jobs_dev, map_dev, nblocks) for the current set of entries
stats = {'single': 0, 'multi': 0, 'hits': 0}


class _Entry:
    __slots__ = ('ptr', 'planes', 'jobs', 'jobs_array', 'version', 'epoch', 'stream')


def enabled():
    return _ENABLED


def note_weights_changed():
    """Parameters were written behind autograd's back (raw-pointer kernels): every cached plane is stale."""
    global _epoch
    _epoch += 1


def clear():
    global _table
    with _lock:
        _by_weight.clear()
        _table = None


def _desc_key(d, for_dgrad):
    return (d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.kh, d.kw, d.stride_h, d.stride_w, d.pad_h, d.pad_w,
            d.dil_h, d.dil_w, for_dgrad)


def _layout(d, for_dgrad):
    """(signature, plane bytes, job-array capacity) of the planes this descriptor's kernels read.  The signature is the jobs'
    own parameters: two geometries that lead to the same planes (same kernels' layouts) share an entry."""
    key = _desc_key(d, for_dgrad)
    lay = _layouts.get(key)
    if lay is None:
        lib = _C.load()
        nj = lib.evk_conv2d_split_job_count(ctypes.byref(d), for_dgrad)
        jobs = (_C.SplitJob * nj)()
        # pointers are placeholders here (non-null); only kind / arg identify the layout
        n = lib.evk_conv2d_split_jobs(ctypes.byref(d), 16, for_dgrad, 16, jobs, nj)
        if n < 0:
            _C.check(n, 'evk_conv2d_split_jobs')
        sig = tuple((jobs[i].kind,) + tuple(jobs[i].arg[:12]) + (jobs[i].out - 16,) for i in range(n))
        lay = (sig, int(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), for_dgrad)), nj)
        _layouts[key] = lay
    return lay


def _valid(e, weight, stream):
    return (e.ptr == weight.data_ptr() and e.version == weight._version and e.epoch == _epoch
            and e.stream == stream)


def _build_table(device):
    """Job table + (job, block-in-job) map of every live entry, uploaded once per change of the entry set."""
    global _table
    lib = _C.load()
    live = [e for per in _by_weight.values() for e in per.values()]
    njobs = sum(len(e.jobs) for e in live)
    jobs = (_C.SplitJob * max(njobs, 1))()
    rows = []
    k = 0
    for e in live:
        for j in e.jobs:
            ctypes.memmove(ctypes.byref(jobs[k]), ctypes.byref(j), ctypes.sizeof(_C.SplitJob))
            pairs = int(lib.evk_split_job_pairs(ctypes.byref(jobs[k])))
            nb = max(1, (pairs + _PAIRS_PER_BLOCK - 1) // _PAIRS_PER_BLOCK)
            jobs[k].arg[12] = nb
            rows.append(np.stack([np.full(nb, k, dtype=np.int32), np.arange(nb, dtype=np.int32)], axis=1))
            k += 1
    bmap = np.concatenate(rows, axis=0) if rows else np.zeros((0, 2), dtype=np.int32)
    raw = np.frombuffer(bytes(jobs), dtype=np.uint8)[:njobs * ctypes.sizeof(_C.SplitJob)].copy()
    jobs_dev = torch.from_numpy(raw).to(device)
    map_dev = torch.from_numpy(np.ascontiguousarray(bmap)).to(device)
    _table = (jobs_dev, map_dev, int(bmap.shape[0]), len(live))


def _refresh_all(device, stream):
    """One launch for every registered weight (entries of dead tensors left the weak dictionary by themselves;
    those of tensors that moved are dropped here)."""
    global _table
    alive = 0
    for w, per in list(_by_weight.items()):
        moved = [k for k, e in per.items() if e.ptr != w.data_ptr()]
        for k in moved:
            del per[k]
            _table = None
        alive += len(per)
    if _table is None or _table[3] != alive:
        _build_table(device)
    jobs_dev, map_dev, nblocks, _ = _table
    _C.call('evk_conv2d_split_multi', jobs_dev.data_ptr(), map_dev.data_ptr(), nblocks, stream)
    stats['multi'] += 1
    for w, per in _by_weight.items():
        for e in per.values():
            e.version, e.epoch, e.stream = w._version, _epoch, stream


def planes_for(weight, w_dense, d, for_dgrad, stream):
    """Device pointer of up-to-date planes of `weight` for descriptor `d`, or None when the cache does not apply
    (disabled, or `w_dense` — the OHWI memory the kernels read — is a transient re-laid-out copy of `weight`)."""
    global _table
    if not _ENABLED or w_dense.data_ptr() != weight.data_ptr():
        return None
    with _lock:
        sig, nbytes, njobs = _layout(d, for_dgrad)
        per = _by_weight.get(weight)
        if per is None:
            per = _by_weight[weight] = {}
        e = per.get(sig)
        if e is not None and _valid(e, weight, stream):
            stats['hits'] += 1
            return e.planes.data_ptr()
        if e is not None and e.ptr == weight.data_ptr():
            _refresh_all(weight.device, stream)       # stale: the step's one launch, for every weight
            e = per.get(sig)
            if e is not None and _valid(e, weight, stream):
                return e.planes.data_ptr()
        # first sight of this (weight, layout): own buffer, own jobs, split alone this once
        lib = _C.load()
        e = _Entry()
        e.ptr = weight.data_ptr()
        e.planes = torch.empty((nbytes,), device=weight.device, dtype=torch.uint8)
        jobs = (_C.SplitJob * njobs)()
        n = lib.evk_conv2d_split_jobs(ctypes.byref(d), e.ptr, for_dgrad, e.planes.data_ptr(), jobs, njobs)
        if n < 0:
            _C.check(n, 'evk_conv2d_split_jobs')
        e.jobs_array = jobs                      # owns the memory the per-job views below alias
        e.jobs = [jobs[i] for i in range(n)]
        _C.call('evk_conv2d_split_weight', ctypes.byref(d), e.ptr, for_dgrad, e.planes.data_ptr(), stream)
        stats['single'] += 1
        e.version, e.epoch, e.stream = weight._version, _epoch, stream
        per[sig] = e
        _table = None
        return e.planes.data_ptr()
