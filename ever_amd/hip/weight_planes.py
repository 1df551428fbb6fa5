"""Persistent weight planes of the split-arithmetic convolutions ("f16x2": two fp16 planes of w / s + the bit image of
max|w| the scale s derives from; "bf16x3" / "bf16": three bf16 planes), refreshed by ONE launch per weight update and
arithmetic (evk_conv2d_split_multi[_f16x2], preceded by one evk_absmax_multi) instead of one per convolution and direction.

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
_PAIRS_PER_BLOCK = int(os.environ.get('EVK_SPLIT_PAIRS', '2048'))          # 256 threads x 8 trips (EVK_SPLIT_PAIRS: A/B)
_lock = threading.RLock()        # forward (main thread) and backward (autograd engine thread) both come here
_epoch = 0
_by_weight = WeakIdKeyDictionary()   # weight tensor object (weakly, by identity) -> {layout signature: _Entry}
_layouts = {}                    # (descriptor fields, for_dgrad) -> (layout signature, plane bytes, job count)
_table = None                    # per arithmetic kind: (jobs_dev, map_dev, nblocks) for the current set of entries
stats = {'single': 0, 'multi': 0, 'hits': 0}
_SLOTS = 8192                    # capacity of the weights' absmax array (one word per weight tensor, f16x2 planes)
_wabs = {}                       # device -> int32[_SLOTS]: bit image of max|w| per slot
_slot_of = WeakIdKeyDictionary() # weight tensor object -> slot
_next_slot = [0]
_free_slots = []
_abs_ws = {}                     # (device, stream) -> zeroed workspace of evk_absmax


class _Entry:
    __slots__ = ('ptr', 'planes', 'jobs', 'jobs_array', 'version', 'epoch', 'stream', 'kind', 'slot', 'numel')


def absmax_workspace(device, stream):
    """The (partials, ticket) scratch of evk_absmax: zero before its first use, one per stream."""
    key = (device, stream)
    ws = _abs_ws.get(key)
    if ws is None:
        ws = _abs_ws[key] = torch.zeros((_C.load().evk_absmax_workspace_bytes(),), dtype=torch.uint8, device=device)
    return ws


def _wabs_for(device):
    t = _wabs.get(device)
    if t is None:
        t = _wabs[device] = torch.zeros((_SLOTS,), dtype=torch.int32, device=device)
    return t


def _slot(weight):
    """Index of `weight`'s word in the absmax array; indices of dead weight tensors are handed out again."""
    sl = _slot_of.get(weight)
    if sl is None:
        if not _free_slots:
            if _next_slot[0] < _SLOTS:
                _free_slots.append(_next_slot[0])
                _next_slot[0] += 1
            else:       # the range is used up: collect the indices whose tensors have died since
                live = set(_slot_of.values())
                _free_slots.extend(i for i in range(_SLOTS) if i not in live)
                if not _free_slots:
                    raise RuntimeError('weight_planes: more than %d live weight tensors registered' % _SLOTS)
        sl = _slot_of[weight] = _free_slots.pop()
    return sl


def enabled():
    return _ENABLED


_stats_epoch = 0


def note_running_stats_changed():
    """A training-mode BatchNorm kernel updated running_mean / running_var through raw pointers (no autograd version
    bump): whatever was derived from them — folded inference weights (module/fold.py) — is stale."""
    global _stats_epoch
    _stats_epoch += 1


def note_weights_changed():
    """Parameters were written behind autograd's back (raw-pointer kernels): every cached plane is stale."""
    global _epoch
    _epoch += 1


def clear():
    global _table
    with _lock:
        _by_weight.clear()
        _table = None
        _slot_of.clear()
        _next_slot[0] = 0
        del _free_slots[:]


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


_alias = {}                      # raw stream -> the stream it forks from and joins again within a step


def alias_stream(branch, origin):
    """`branch` (raw handle) only ever runs work that is ordered behind `origin`'s by a fork — the head's branch stream
    (functional.HeadBranches), forward and backward: planes produced on `origin` are valid there, and a request from it must
    not re-split every registered weight under the kernels of `origin` that read them."""
    if branch != origin:
        _alias[branch] = _alias.get(origin, origin)


def _canon(stream):
    return _alias.get(stream, stream)


def _valid(e, weight, stream):
    return (e.ptr == weight.data_ptr() and e.version == weight._version and e.epoch == _epoch
            and e.stream == _canon(stream))


def _build_table(device):
    """Per arithmetic kind: job table + (job, block-in-job) map of every live entry, uploaded once per change of the
    entry set; for the f16x2 kind also the (pointer, size) table evk_absmax_multi walks (indexed by slot)."""
    global _table
    lib = _C.load()
    tables = {}
    nlive = 0
    for kind in ('b', 'h'):
        live = [e for per in _by_weight.values() for e in per.values() if e.kind == kind]
        nlive += len(live)
        njobs = sum(len(e.jobs) for e in live)
        jobs = (_C.SplitJob * max(njobs, 1))()
        rows = []
        k = 0
        for e in live:
            for j in e.jobs:
                ctypes.memmove(ctypes.byref(jobs[k]), ctypes.byref(j), ctypes.sizeof(_C.SplitJob))
                pairs = int(lib.evk_split_job_pairs(ctypes.byref(jobs[k])))
                nb = max(1, (pairs + _PAIRS_PER_BLOCK - 1) // _PAIRS_PER_BLOCK)
                jobs[k].arg[11] = e.slot if kind == 'h' else 0
                jobs[k].arg[12] = nb
                rows.append(np.stack([np.full(nb, k, dtype=np.int32), np.arange(nb, dtype=np.int32)], axis=1))
                k += 1
        bmap = np.concatenate(rows, axis=0) if rows else np.zeros((0, 2), dtype=np.int32)
        raw = np.frombuffer(bytes(jobs), dtype=np.uint8)[:njobs * ctypes.sizeof(_C.SplitJob)].copy()
        tab = [torch.from_numpy(raw).to(device), torch.from_numpy(np.ascontiguousarray(bmap)).to(device), int(bmap.shape[0])]
        if kind == 'h':
            nslots = max([e.slot for e in live], default=-1) + 1
            ptrs, sizes = np.zeros(max(nslots, 1), dtype=np.int64), np.zeros(max(nslots, 1), dtype=np.int64)
            for e in live:
                ptrs[e.slot], sizes[e.slot] = e.ptr, e.numel
            tab += [torch.from_numpy(ptrs).to(device), torch.from_numpy(sizes).to(device), nslots]
        tables[kind] = tab
    _table = (tables, nlive)


def _refresh_all(device, stream):
    """One launch per arithmetic kind for every registered weight (entries of dead tensors left the weak dictionary by
    themselves; those of tensors that moved are dropped here)."""
    global _table
    alive = 0
    for w, per in list(_by_weight.items()):
        moved = [k for k, e in per.items() if e.ptr != w.data_ptr()]
        for k in moved:
            del per[k]
            _table = None
        alive += len(per)
    if _table is None or _table[1] != alive:
        _build_table(device)
    tables = _table[0]
    jobs_dev, map_dev, nblocks = tables['b'][:3]
    if nblocks:
        _C.call('evk_conv2d_split_multi', jobs_dev.data_ptr(), map_dev.data_ptr(), nblocks, stream)
    jobs_dev, map_dev, nblocks, ptrs_dev, sizes_dev, nslots = tables['h']
    if nblocks:
        wabs = _wabs_for(device)
        _C.call('evk_absmax_multi', ptrs_dev.data_ptr(), sizes_dev.data_ptr(), nslots, wabs.data_ptr(), stream)
        _C.call('evk_conv2d_split_multi_f16x2', jobs_dev.data_ptr(), map_dev.data_ptr(), nblocks, wabs.data_ptr(), stream)
    stats['multi'] += 1
    for w, per in _by_weight.items():
        for e in per.values():
            e.version, e.epoch, e.stream = w._version, _epoch, _canon(stream)


def planes_for(weight, w_dense, d, for_dgrad, stream, f16x2=False):
    """Device pointer of up-to-date planes of `weight` for descriptor `d` — with `f16x2` the pair (planes, word holding
    the bit image of max|w|) — or None when the cache does not apply (disabled, or `w_dense` — the OHWI memory the
    kernels read — is a transient re-laid-out copy of `weight`)."""
    global _table
    if not _ENABLED or w_dense.data_ptr() != weight.data_ptr() or getattr(weight, '_evk_transient', False):
        return None
    kind = 'h' if f16x2 else 'b'
    with _lock:
        lay, nbytes, njobs = _layout(d, for_dgrad)
        sig = (kind,) + lay

        def result(e):
            if kind == 'b':
                return e.planes.data_ptr()
            return e.planes.data_ptr(), _wabs_for(weight.device).data_ptr() + 4 * e.slot
        per = _by_weight.get(weight)
        if per is None:
            per = _by_weight[weight] = {}
        e = per.get(sig)
        if e is not None and _valid(e, weight, stream):
            stats['hits'] += 1
            return result(e)
        if e is not None and e.ptr == weight.data_ptr():
            _refresh_all(weight.device, stream)       # stale: the step's one launch, for every weight
            e = per.get(sig)
            if e is not None and _valid(e, weight, stream):
                return result(e)
        # first sight of this (weight, layout): own buffer, own jobs, split alone this once
        lib = _C.load()
        e = _Entry()
        e.kind = kind
        e.ptr = weight.data_ptr()
        e.numel = weight.numel()
        e.slot = _slot(weight) if kind == 'h' else -1
        e.planes = torch.empty((nbytes,), device=weight.device, dtype=torch.uint8)
        jobs = (_C.SplitJob * njobs)()
        n = lib.evk_conv2d_split_jobs(ctypes.byref(d), e.ptr, for_dgrad, e.planes.data_ptr(), jobs, njobs)
        if n < 0:
            _C.check(n, 'evk_conv2d_split_jobs')
        e.jobs_array = jobs                      # owns the memory the per-job views below alias
        e.jobs = [jobs[i] for i in range(n)]
        if kind == 'h':
            wabs_ptr = _wabs_for(weight.device).data_ptr() + 4 * e.slot
            one = torch.tensor([e.ptr, e.numel], dtype=torch.int64, device=weight.device)   # one-entry (pointer, size) table
            _C.call('evk_absmax_multi', one.data_ptr(), one.data_ptr() + 8, 1, wabs_ptr, stream)
            _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), e.ptr, for_dgrad, e.planes.data_ptr(), wabs_ptr, stream)
        else:
            _C.call('evk_conv2d_split_weight', ctypes.byref(d), e.ptr, for_dgrad, e.planes.data_ptr(), stream)
        stats['single'] += 1
        e.version, e.epoch, e.stream = weight._version, _epoch, _canon(stream)
        per[sig] = e
        _table = None
        return result(e)
