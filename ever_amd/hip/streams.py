"""Second streams of a training step: weight gradients beside the backward chain and — measured, off by
default — the head's pyramid levels on a branch stream.  Part of the hip/functional.py facade."""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from . import timing, weight_planes
from .workspace import workspace
from ._base import (  # noqa: F401
    _dist_initialized, _stream, observers_active,
)


# Weight gradients on a second stream.  Nothing downstream in a backward pass depends on dw, and a weight gradient is
# MFMA-bound where the chain it would otherwise interrupt (BatchNorm backward, one-tap data gradients, pointwise passes) is
# HBM-bound: launched beside that chain it fills the matrix pipe while the chain fills the memory system (+4.7 % on the
# FarSeg-R50 step).  Rules that keep it invisible:
#  * only for LEAF weight (and bias) whose .grad is None, that nobody hooks, and whose memory order is the one the gradient
#    comes in (AccumulateGrad then only STORES the tensor; an in-place accumulation, a hook, or the deep copy it makes of a
#    gradient that breaks the layout contract would read dw on the main stream) — FlatGradDDP opts its parameters in and
#    packs the bucket on this stream (trainer/grad_reducer.py);
#  * only for parameters used ONCE in the forward of this pass (_note_param_use): the engine sums the gradients of a
#    multiply used leaf in its own input buffer, on the main stream, before any hook runs.  A second consumer OUTSIDE this
#    package (an L2 term built from the weights in the loss) is invisible to that count; the end-of-pass check
#    (_wgrad_pass_done) sees that .grad is not the tensor the weight gradient was written to and raises;
#  * operands and results (allocated on the main stream) are kept alive in _WGRAD_HOLD until the join — cheaper on the
#    host than record_stream (an event per block when it is freed: 5 ms per step) at the price of saved activations and
#    output gradients living to the end of the backward pass (bounded by EVK_WGRAD_HOLD_GB: a join in mid-pass beyond it);
#    the weight gradient's own temporaries belong to the side stream;
#  * the main stream waits for the side stream at the END of the backward pass (autograd final callback), so everything
#    after backward() — optimiser, clipping, .grad readers — is ordered as before;
#  * under a hipGraph capture the side stream forks from the capturing stream by the same event and joins it again in the
#    end-of-backward callback, so a replay holds the same two branches as the eager step.
_WGRAD_STREAM = [os.environ.get('EVK_WGRAD_STREAM', '1') != '0']
_WGRAD_SIDE = {}
_WGRAD_PASS = {'pending': False, 'gid': None}     # gid: the backward pass (graph task) whose end-of-pass join is queued
_WGRAD_HOLD = []                 # tensors of the main stream's pool that a pending weight gradient reads or writes
_WGRAD_OWNED = {}                # id(leaf) -> (leaf, storage address of the gradient the side stream wrote), this pass
_WGRAD_HOLD_BYTES = [0]
# operand bytes held for pending weight gradients beyond which the backward joins the side stream in mid-pass:
# EVK_WGRAD_HOLD_GB, default a quarter of the device's memory (a fixed 64 GB was the whole of a smaller part)
_WGRAD_HOLD_CAP = [int(float(os.environ['EVK_WGRAD_HOLD_GB']) * 2 ** 30) if 'EVK_WGRAD_HOLD_GB' in os.environ else None]
_WGRAD_MAIN = {}                 # device -> {stream id: stream} the weight gradients forked from (joins go to each)
_cuda_get_stream = getattr(torch._C, '_cuda_getCurrentStream', None)
_cuda_set_stream = getattr(torch._C, '_cuda_setStream', None)


_WGRAD_QUEUE = {}                # device -> launch closures of weight gradients not issued yet (one list per
                                 # device — autograd runs one engine thread per device, and a shared list lost appends)
_WGRAD_EARLY_MODE = int(os.environ.get('EVK_WGRAD_EARLY', '0'))
_WGRAD_EARLY = _WGRAD_EARLY_MODE == 1
_WGRAD_SHARED = 32    # EVK_CONV_WGRAD_SHARED (include/ever_hip.h): the launch runs beside the backward chain — wide tiles on half of the CUs
# (EVK_WGRAD_SHARED=0 / set_wgrad_shared_split(False): the side stream's launches split as if they ran alone — the same
# accumulation order as the single-stream step, which the bit-for-bit tests of the mechanism pin; +1.1 .. +2.1 % on the step when on)
_WGRAD_SHARED_ON = [os.environ.get('EVK_WGRAD_SHARED', '1') != '0']


def set_wgrad_shared_split(on):
    """runtime switch of the half-chip split of side-stream weight gradients (returns the previous setting)"""
    prev, _WGRAD_SHARED_ON[0] = _WGRAD_SHARED_ON[0], bool(on)
    return prev


_WGRAD_BATCH = [max(1, int(os.environ.get('EVK_WGRAD_BATCH', '1')))]   # (8, 16, 32 measured: 524 vs 531 tiles/s for 1, same box)


def flush_wgrad_queue(dev=None):
    """issue the queued weight gradients on the side stream, behind ONE event recorded on the backward's stream now.
    dev: that device's queue only (the backward thread of a device flushes its own); None: every device's."""
    for d in ([dev] if dev is not None else list(_WGRAD_QUEUE.keys())):
        fns = _WGRAD_QUEUE.pop(d, None)      # (atomic under the GIL: an append racing with it starts a fresh list)
        if not fns:
            continue
        if d.index is not None and d.index != torch.cuda.current_device():
            with torch.cuda.device(d):       # another device's queue (flushed from the main thread at the end of a pass)
                _issue_wgrads(d, fns)
        else:
            _issue_wgrads(d, fns)


def _issue_wgrads(d, fns):
    side = _WGRAD_SIDE[d]
    main_id = _cuda_get_stream(d.index)
    mains = _WGRAD_MAIN.setdefault(d, {})    # every stream weight gradients forked from (the backward's, the head's branch stream)
    main = mains.get(main_id[0])
    if main is None:
        main = mains[main_id[0]] = torch.cuda.current_stream(d)
    _C.call('evk_stream_fork', main.cuda_stream, side.cuda_stream)
    # torch's current stream by the raw setter (the Python context manager costs 20 us)
    _cuda_set_stream(stream_id=side.stream_id, device_index=side.device_index, device_type=side.device_type)
    try:
        for fn in fns:
            fn(side.cuda_stream)
    finally:
        _cuda_set_stream(stream_id=main_id[0], device_index=main_id[1], device_type=main_id[2])


def _wgrad_hold(*tensors):
    for t in tensors:
        if t is not None:
            _WGRAD_HOLD.append(t)
            _WGRAD_HOLD_BYTES[0] += t.numel() * t.element_size()
wgrad_stream_stats = {'side': 0, 'main': 0}     # weight gradients launched on the side stream / kept on the main stream


def set_wgrad_stream(on):
    """runtime switch of the weight-gradient side stream (bench.py measures the kernels alone with it off)"""
    prev, _WGRAD_STREAM[0] = _WGRAD_STREAM[0], bool(on)
    return prev


_USED_PARAMS = []


def _note_param_use(*params):
    """Forward bookkeeping of the side-stream rule "one use per pass": every entry point of this package that consumes a
    parameter as a convolution weight / bias counts it.  A leaf that feeds SEVERAL nodes gets its gradients summed in the
    autograd engine's input buffer as they arrive — an add on the main stream that no hook sees — so only single-use
    parameters may take their gradient from the side stream.  (Not visible here: a use of the same parameter by a torch op
    outside this package, e.g. an explicit L2 term in the loss; set EVK_WGRAD_STREAM=0 for such models.)"""
    if not _WGRAD_STREAM[0]:
        return
    if _SIDE_SELFTEST[0] is None:
        # (no test of torch.is_grad_enabled() here: most callers are autograd.Function.forward bodies, which always run with
        # grad mode off — they only come here when an input needs a gradient; the probe enables grad mode for itself.  With
        # that test in place the probe only ever ran from the one plain-Python caller, bn_relu_dot, and models without a
        # commuted decoder classifier silently trained single-stream: ChangeStar -5 %, FreeNet -10 % on one box)
        p0 = next((p for p in params if p is not None and p.is_cuda and p.requires_grad), None)
        if p0 is not None and not torch.cuda.is_current_stream_capturing():
            _SIDE_SELFTEST[0] = _side_stream_selftest(p0.device)
            if not _SIDE_SELFTEST[0]:
                _WGRAD_STREAM[0] = False
                return
    for p in params:
        if p is not None and p.requires_grad and p.is_leaf:
            n = p.__dict__.get('_evk_uses', 0)
            if n == 0:
                _USED_PARAMS.append(p)
            p._evk_uses = n + 1


def wgrad_stream_enabled():
    return bool(_WGRAD_STREAM[0])


def _leaf_ok(t):
    """a leaf whose gradient arrives for the first time in this accumulation and that nobody but this module hooks"""
    if not t.is_leaf or t.grad is not None or t.__dict__.get('_evk_uses', 0) != 1 or torch.is_grad_enabled():
        return False         # (grad mode inside a backward pass = create_graph: AccumulateGrad copies instead of storing)
    if t._backward_hooks:
        return False
    return not getattr(t, '_post_accumulate_grad_hooks', None) or getattr(t, '_evk_flat_ddp', False)


_SIDE_SELFTEST = [None]          # None: not run yet; True / False: what the engine of this torch build does
_SIDE_TESTED_TORCH = ('2.10',)   # builds the side stream's assumptions about the autograd engine were developed against


def _side_stream_selftest(dev):
    """The side stream leans on engine behaviour that is not a public contract: the id of the running
    graph task, final callbacks queued from inside a backward node, AccumulateGrad STORING a first gradient as it is (same
    storage, no read), the raw current-stream setter.  Checked once per process on a four-element problem before the first
    weight gradient goes to the side stream; on any other answer the side stream is switched off, loudly, and training goes on
    single-stream (bit-identical results, ~5 % slower)."""
    import warnings
    try:
        if _cuda_get_stream is None or _cuda_set_stream is None or not hasattr(torch._C, '_current_graph_task_id'):
            raise RuntimeError('torch._C._cuda_{get,set}Stream / _current_graph_task_id missing')
        seen = {}

        class _Probe(Function):
            @staticmethod
            def forward(ctx, w):
                return w * 2.0

            @staticmethod
            def backward(ctx, g):
                seen['gid'] = torch._C._current_graph_task_id()
                torch.autograd.Variable._execution_engine.queue_callback(lambda: seen.__setitem__('cb', True))
                out = g * 2.0
                seen['ptr'] = out.untyped_storage().data_ptr()
                return out
        cur = _cuda_get_stream(dev.index)
        with torch.enable_grad():        # (the caller is usually inside a Function.forward: grad mode is off there)
            w = torch.ones(4, device=dev, requires_grad=True)
            _Probe.apply(w).sum().backward()
        if seen.get('gid', -1) < 0:
            raise RuntimeError('no graph task id inside a backward node')
        if not seen.get('cb'):
            raise RuntimeError('a final callback queued inside a backward node did not run')
        if w.grad is None or w.grad.untyped_storage().data_ptr() != seen['ptr']:
            raise RuntimeError('AccumulateGrad copied a first gradient instead of storing it')
        if _cuda_get_stream(dev.index)[0] != cur[0]:
            raise RuntimeError('the current stream changed across a backward pass')
        ok = True
    except Exception as e:       # noqa: BLE001 (anything unexpected = do not trust the mechanism)
        warnings.warn(f'ever_amd: weight-gradient side stream disabled — this torch build ({torch.__version__}) does not behave '
                      f'as the mechanism needs ({e}); training continues single-stream (set EVK_WGRAD_STREAM=0 to silence)')
        ok = False
    if ok and not torch.__version__.startswith(_SIDE_TESTED_TORCH):
        warnings.warn(f'ever_amd: the weight-gradient side stream was developed against torch {_SIDE_TESTED_TORCH[0]}.x; this is '
                      f'{torch.__version__} — its engine self-test passed, the end-of-pass ownership check stays on')
    return ok


def _wgrad_side_stream(dev, weight, bias=None):
    """the side stream for this weight gradient, or None (see the rules above)"""
    if not _WGRAD_STREAM[0] or dev.type != 'cuda' or weight is None:
        return None
    if not _SIDE_SELFTEST[0]:        # (the probe runs from the forward, _note_param_use; never passed = no side stream)
        return None
    leaves = (weight,) if bias is None else (weight, bias)
    flat_ddp = all(getattr(t, '_evk_flat_ddp', False) for t in leaves)
    gid = torch._C._current_graph_task_id()
    if gid < 0:                      # not inside a backward pass (a direct call): stay on the main stream
        return None
    if _WGRAD_PASS['gid'] != gid:
        # first weight gradient of this backward pass
        if _WGRAD_PASS['gid'] is not None:
            # the previous pass died with an exception before its callback ran: join what it left pending.  (Its forward's
            # use counts are still there, so this pass's weights read "used twice" and stay on the main stream; the callback
            # below clears them.)
            wait_wgrad_stream()
        _WGRAD_OWNED.clear()         # (records of a pass that died: nothing stored its gradients)
        torch.autograd.Variable._execution_engine.queue_callback(_wgrad_pass_done)
        _WGRAD_PASS['gid'] = gid
    if not all(_leaf_ok(t) for t in leaves) or (_dist_initialized() and not flat_ddp):
        # a second use of a shared weight in this pass, an accumulation onto an existing .grad, somebody's hook — or another
        # reducer (torch DDP hooks the gradient ACCUMULATORS, invisible on the tensor, and copies gradients into its buckets
        # on the main stream as they arrive): this one runs on the main stream, behind whatever is pending
        wait_wgrad_stream()
        wgrad_stream_stats['main'] += 1
        return None
    wgrad_stream_stats['side'] += 1
    if _WGRAD_HOLD_CAP[0] is None:
        _WGRAD_HOLD_CAP[0] = torch.cuda.mem_get_info(dev)[1] // 4
    if _WGRAD_HOLD_BYTES[0] > _WGRAD_HOLD_CAP[0]:
        wait_wgrad_stream()
    s = _WGRAD_SIDE.get(dev)
    if s is None:
        s = _pick_side_stream(dev, avoid=_HEAD_SIDE.get(dev) or None)
        if s is not None:
            _WGRAD_SIDE[dev] = s
    if not s:                   # no stream of this process overlaps with the backward's stream: stay on it
        wgrad_stream_stats['side'] -= 1
        wgrad_stream_stats['main'] += 1
        return None
    _WGRAD_PASS['pending'] = True
    return s


_SIDE_CANDIDATES = []
# priority of the side stream (torch: lower = more urgent, clamped to the device's range; the backward's stream is torch's default
# stream, priority 0): EVK_WGRAD_PRIO
_WGRAD_PRIO = int(os.environ.get('EVK_WGRAD_PRIO', '0'))


def _pick_side_stream(dev, avoid=None):
    """A stream whose kernels really run beside those of the current stream.  HIP multiplexes streams onto a few hardware
    queues, and which stream objects share one depends on how many streams the process made before (RCCL, a communication
    stream, torch's pools): with FlatGradDDP in the process the first stream made here sat on the backward's own queue —
    every weight gradient serialised behind it, the step SLOWER than without a side stream.  So candidates are measured
    (evk_streams_overlap: two 150 us spin kernels, forked and joined by events, take 150 us or 300) and the first that
    overlaps is kept; the rejected ones stay allocated so that the next candidate lands on another queue.  False when none
    of eight overlaps; None under a stream capture, where nothing can be measured (the caller asks again later)."""
    if torch.cuda.is_current_stream_capturing():
        return None
    main = _stream()
    took = ctypes.c_float(0.0)
    for _ in range(8):
        cand = torch.cuda.Stream(dev, priority=_WGRAD_PRIO)
        rc = _C.load().evk_streams_overlap(main, cand.cuda_stream, 150, ctypes.byref(took))
        if rc < 0:
            _C.check(rc, 'evk_streams_overlap')
        if rc == 1 and avoid:    # (the other side stream of this process: the two must not share a hardware queue either)
            rc = _C.load().evk_streams_overlap(avoid.cuda_stream, cand.cuda_stream, 150, ctypes.byref(took))
            if rc < 0:
                _C.check(rc, 'evk_streams_overlap')
        if rc == 1:
            return cand
        _SIDE_CANDIDATES.append(cand)
    import warnings
    warnings.warn('ever_amd: no HIP stream of this process runs beside the backward stream (all share its hardware queue); '
                  'weight gradients stay on the backward stream')
    return False


def check_side_stream_gradient(leaf):
    """FlatGradDDP, before it replaces .grad by its bucket view: the gradient AccumulateGrad stored for `leaf` must be the
    tensor the side stream wrote (see _wgrad_pass_done, which cannot look any more once the view is in place)"""
    rec = _WGRAD_OWNED.get(id(leaf))
    g = leaf.grad
    if rec is None or g is None:
        return
    if rec[0] is leaf and rec[1] != g.untyped_storage().data_ptr():
        raise RuntimeError(
            'ever_amd: a convolution parameter of shape %s received a second gradient in this backward pass from outside '
            'the HIP convolutions while its weight gradient ran on the side stream; set EVK_WGRAD_STREAM=0 for this model'
            % (tuple(leaf.shape),))


def _wgrad_pass_done():
    _WGRAD_PASS['gid'] = None
    for p in _USED_PARAMS:
        p._evk_uses = 0
    del _USED_PARAMS[:]
    wait_wgrad_stream()
    # AccumulateGrad must have STORED each side-stream gradient as it was.  If .grad lives elsewhere the engine summed it with
    # a gradient from a consumer this package did not see (or copied it) — on the main stream, possibly before the weight
    # gradient had run: loud instead of wrong.  (FlatGradDDP has replaced .grad by its bucket views by now; a parameter
    # without .grad was differentiated by torch.autograd.grad, whose result nothing here can check.)
    owned, bad = list(_WGRAD_OWNED.values()), None
    _WGRAD_OWNED.clear()
    for leaf, addr in owned:
        g = leaf.grad
        if g is not None and not getattr(leaf, '_evk_flat_ddp', False) and g.untyped_storage().data_ptr() != addr:
            bad = leaf
            break
    if bad is not None:
        raise RuntimeError(
            'ever_amd: a convolution parameter of shape %s received a second gradient in this backward pass from outside the '
            'HIP convolutions (e.g. a regulariser built from the weights) while its weight gradient ran on the side stream; '
            'the sum may have read it too early.  Set EVK_WGRAD_STREAM=0 (or functional.set_wgrad_stream(False)) for this '
            'model.' % (tuple(bad.shape),))


def wait_wgrad_stream():
    """the current stream — and the stream the weight gradients forked from, whose pool the held tensors go back to —
    waits for every weight gradient launched on the side stream"""
    flush_wgrad_queue()
    if _WGRAD_PASS['pending']:
        for dev, s in _WGRAD_SIDE.items():
            if s is False:
                continue
            cur = torch.cuda.current_stream(dev)
            cur.wait_stream(s)
            for main in _WGRAD_MAIN.get(dev, {}).values():
                if main != cur:
                    main.wait_stream(s)
            _WGRAD_MAIN.pop(dev, None)       # (origins of THIS pass only: a capture stream must not be waited on later)
        _WGRAD_PASS['pending'] = False
        del _WGRAD_HOLD[:]
        _WGRAD_HOLD_BYTES[0] = 0


def wgrad_side_stream_of(dev):
    """FlatGradDDP: the stream its bucket pack has to follow (None when no weight gradient is pending there)"""
    flush_wgrad_queue()
    return (_WGRAD_SIDE.get(dev) or None) if _WGRAD_PASS['pending'] else None


# The head's pyramid levels on two streams.  Behind the FPN the four levels are independent until the
# decoder's mean (reference fs_relation.py:56-73 relation per level, fpn.py:183-189 decoder branch per level): level 0 (the
# 128^2 map of a 512^2 tile, three quarters of the head's work) stays on the caller's stream, levels 1.. run on a branch
# stream forked behind the FPN and joined in front of the mean.  Their kernels fill a quarter to a half of the chip
# (32^2 / 16^2 maps: 32..128 workgroups) and sit beside the 128^2 kernels instead of in front of them.  Autograd replays
# each node on the stream its forward ran on, so the backward of those levels runs on the branch stream as well, its
# tensors synchronised by the engine at the edges that cross streams (same values either way: no kernel changes its
# launch plan with the stream it is on; tests/test_head_branch_gpu.py pins bit-identity of the two orders).
# MEASURED, OFF BY DEFAULT (EVK_HEAD_BRANCH=1 turns it on; profiles/r06_experiments/ab_head_branch*.txt, head_branch_timeline.txt):
# 556.3 -> 551.1 tiles/s with levels 1..3 aside, 567.1 -> 564.8 / 566.6 with levels 2..3 / level 3 only (EVK_HEAD_BRANCH_FROM).
# Under rocprofv3 the forward has two kernels in flight for 0.86 of its 10.9 ms and is 0.07 ms shorter: the kernels that run
# side by side take 0.73 ms longer than alone (the 64^2 level fills the chip by itself, the 128^2 kernels are bound by the
# matrix pipe's power or by HBM, so sharing is a zero-sum split), and the backward is 0.65 ms LONGER (the engine's
# cross-stream events leave 0.7 ms more of it with no kernel in flight, and the weight-gradient stream waits for whichever
# of the two streams forked last).  The step is bound by resource-time, not by the dependency chain.
_HEAD_BRANCH = [os.environ.get('EVK_HEAD_BRANCH', '0') == '1']
_HEAD_FROM = int(os.environ.get('EVK_HEAD_BRANCH_FROM', '1'))     # first pyramid level that goes to the branch stream
_HEAD_SIDE = {}                  # device -> stream, or False when no stream of the process runs beside the caller's


def set_head_branch(on):
    """runtime switch of the head's branch stream (returns the previous setting)"""
    prev, _HEAD_BRANCH[0] = _HEAD_BRANCH[0], bool(on)
    return prev


class HeadBranches:
    """One fork / join of the branch stream: `with br.level(i):` around the work of pyramid level i, `br.join()` in front of
    the first consumer of all levels.  level(0) and every level of a session without a stream are no-ops."""

    def __init__(self, dev, side):
        self.dev, self.side = dev, side
        self.main_id = None          # (stream_id, device_index, device_type) of the caller's stream while a level runs aside
        self.main_raw = None
        self.forked = False

    class _Level:
        def __init__(self, br, aside):
            self.br, self.aside = br, aside

        def __enter__(self):
            br = self.br
            if not self.aside:
                return br
            if not br.forked:
                br.main_raw = _stream()
                _C.call('evk_stream_fork', br.main_raw, br.side.cuda_stream)
                weight_planes.alias_stream(br.side.cuda_stream, br.main_raw)
                br.forked = True
            br.main_id = _cuda_get_stream(br.dev.index)
            s = br.side
            _cuda_set_stream(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)
            return br

        def __exit__(self, *exc):
            br = self.br
            if self.aside and br.main_id is not None:
                _cuda_set_stream(stream_id=br.main_id[0], device_index=br.main_id[1], device_type=br.main_id[2])
                br.main_id = None
            return False

    def level(self, i):
        return HeadBranches._Level(self, bool(self.side) and i >= _HEAD_FROM)

    def join(self):
        if self.forked:
            _C.call('evk_stream_fork', self.side.cuda_stream, _stream())
            self.forked = False


def head_branches(t):
    """a HeadBranches session for the head that consumes the CUDA tensor t, or None (switched off, no raw stream setters in
    this torch build, observers installed, no second hardware queue)"""
    if not _HEAD_BRANCH[0] or not t.is_cuda or _cuda_get_stream is None or _cuda_set_stream is None:
        return None
    if observers_active():
        return None
    dev = t.device
    s = _HEAD_SIDE.get(dev)
    if s is None:
        if torch.cuda.is_current_stream_capturing():
            return None          # (under a stream capture nothing can be measured: plain order this time)
        # the weight-gradient stream is picked HERE, from the caller's stream, when it has not been yet: the first weight
        # gradient of a backward pass would otherwise pick it from inside a node that runs on the branch stream
        if _WGRAD_STREAM[0] and _WGRAD_SIDE.get(dev) is None:
            w = _pick_side_stream(dev)
            if w is not None:
                _WGRAD_SIDE[dev] = w
        s = _pick_side_stream(dev, avoid=_WGRAD_SIDE.get(dev) or None)
        if s is None:
            return None
        _HEAD_SIDE[dev] = s
    return HeadBranches(dev, s) if s else None
