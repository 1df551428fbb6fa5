"""Autograd-aware host wrappers over the C-ABI kernels (include/ever_hip.h).

Every function here takes / returns torch CUDA tensors whose *logical* shape is the reference's
NCHW and whose *memory* is dense NHWC (channels_last); convolution weights are logical OIHW with
OHWI memory.  There is no CPU implementation: a CPU tensor raises `HipPathError`.

Reference call sites replaced (ever/module/...): see the per-function docstrings.
"""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from . import timing, weight_planes
from .workspace import workspace

__all__ = [
    'HipPathError', 'empty_nhwc', 'as_nhwc', 'is_nhwc', 'image_to_nhwc', 'conv2d', 'conv2d_fork', 'conv_transpose2d', 'batch_norm_act', 'relu',
    'max_pool3x3s2', 'upsample_nearest2x_add', 'upsample_bilinear', 'global_avg_pool', 'fs_relation',
    'mean4', 'add', 'bce_with_logits', 'dice_loss_with_logits', 'cross_entropy', 'soft_cross_entropy',
]


class HipPathError(RuntimeError):
    """Raised when the HIP path is asked to run on something it cannot (CPU tensor, wrong dtype)."""


# Arithmetic of the convolution GEMMs.  Both are fp32 in, fp32 out, fp32 accumulate:
#   'bf16x3' (default) each fp32 operand is split exactly into three bf16 terms and the product is rebuilt from
#            six bf16 MFMA partial products (error per product < 2^-24: below one fp32 rounding), on the
#            v_mfma_f32_32x32x16_bf16 pipe;
#   'f32'    v_mfma_f32_32x32x2_f32, an exact fmaf chain (the parity yardstick for the split kernels).
#   'bf16'   plain bf16 operands (rounded once), ONE bf16 MFMA product per operand pair, fp32 accumulate, fp32 tensors:
#            the counterpart of the reference's `--mixed_precision bf16` (core/launcher.py:40-80).  Opt-in only.
#   'f16x2'  each fp32 operand is divided by a per-tensor power of two and split into two fp16 terms; three partial
#            products on the fp16 matrix pipe, fp32 accumulate, scales multiplied back (csrc/x3_common.hpp).  As accurate
#            against fp64 as 'bf16x3' on every layer shape (tools/check_f16x2.py) at half the matrix work.
_CONV_MATH = os.environ.get('EVK_CONV_MATH', 'f16x2')
_MATH_MODES = ('f16x2', 'bf16x3', 'f32', 'bf16')


def _planes_math():
    """True when the convolutions run on the bf16 matrix pipe from weight planes (exact split or plain bf16)."""
    return _CONV_MATH in ('f16x2', 'bf16x3', 'bf16')


def _f16x2():
    return _CONV_MATH == 'f16x2'


absmax_stats = {'hits': 0, 'standalone': 0, 'fused': 0, 'packed': 0}   # where the operand scales came from (tools / tests)
_FUSED_AMAX = os.environ.get('EVK_FUSED_ABSMAX', '1') != '0'


def _note_amax(t, bits):
    """Record that `bits` holds max|t| (written by the kernel that produced t)."""
    try:
        t._evk_amax = (t._version, t.data_ptr(), bits)
        absmax_stats['fused'] += 1
    except (AttributeError, RuntimeError):
        pass


def _inherit_amax(out, src):
    """`out` is bounded element-wise by max|src| (a convex combination, a selection, a product with a factor in [0, 1]):
    src's word is a valid — at most slightly loose — scale source for out, and saves its read pass.  The f16x2 operand
    keeps its 22 bits for every element within 2^-17 of the bound (csrc/x3_common.hpp)."""
    hit = getattr(src, '_evk_amax', None)
    if _FUSED_AMAX and hit is not None and hit[0] == src._version and hit[1] == src.data_ptr():
        _note_amax(out, hit[2])
    return out


_AMAX_WORDS = [0]


def _amax_buf(dev):
    """An activation scale buffer (64 slots of partial max|x| bit images, include/ever_hip.h: evk_absmax)."""
    if not _AMAX_WORDS[0]:
        _AMAX_WORDS[0] = int(_C.load().evk_absmax_words())
    return torch.empty((_AMAX_WORDS[0],), device=dev, dtype=torch.int32)


def absmax_value(bits):
    """The bit image of max|x| held by an activation scale buffer (tests, tools): the maximum of its slots."""
    n = bits.numel() // 64
    return int(bits.view(64, n)[:, 0].max().item())


_ZERO_POOL = {}


def _amax_zeroed(dev):
    """A scale buffer whose slots are zero (for producers that only RAISE slots: the convolution epilogues), or None.
    Buffers are cut from a pool zeroed 256 at a time: one fill launch per 256 convolution outputs."""
    if not (_FUSED_AMAX and _f16x2()):
        return None
    if not _AMAX_WORDS[0]:
        _AMAX_WORDS[0] = int(_C.load().evk_absmax_words())
    nw = _AMAX_WORDS[0]
    key = (dev, _stream())          # the fill launch and the kernels that raise the slots must share a stream's order
    pool = _ZERO_POOL.get(key)
    if pool is None or pool[1] >= pool[0].shape[0]:
        pool = _ZERO_POOL[key] = [torch.zeros((256, nw), device=dev, dtype=torch.int32), 0]
    buf = pool[0][pool[1]]
    pool[1] += 1
    return buf


def _amax_out(dev):
    """A scale buffer for a producer kernel to leave max|output| in, or None when no consumer will want it."""
    return _amax_buf(dev) if (_FUSED_AMAX and _f16x2()) else None


def absmax_bits(t, st):
    """Activation scale buffer of t (the f16x2 operand scale derives from the maximum of its slots inside the kernels).  Cached on the tensor object: the forward's scale of x serves the weight gradient, the scale of dy serves
    data and weight gradient, a block input serves both convolutions that read it."""
    hit = getattr(t, '_evk_amax', None)
    if hit is not None and hit[0] == t._version and hit[1] == t.data_ptr():
        absmax_stats['hits'] += 1
        return hit[2]
    absmax_stats['standalone'] += 1
    bits = _amax_buf(t.device)
    sp = timing.span('absmax', 0.0, 4.0 * t.numel())
    _C.call('evk_absmax', t.data_ptr(), t.numel(), bits.data_ptr(), weight_planes.absmax_workspace(t.device, st).data_ptr(), st)
    if sp is not None:
        sp.stop()
    try:
        t._evk_amax = (t._version, t.data_ptr(), bits)
    except (AttributeError, RuntimeError):
        pass
    return bits


def _weight_planes(weight, w_dense, w_ptr, d, for_dgrad, st, dev):
    """(planes pointer, weight absmax pointer or None, keep-alive) for the current plane arithmetic: from the cache of
    registered weights, else split into the shared workspace on this call (a transient re-laid-out copy)."""
    h2 = _f16x2()
    hit = weight_planes.planes_for(weight, w_dense, d, for_dgrad, st, f16x2=h2)
    if hit is not None:
        return (hit[0], hit[1], None) if h2 else (hit, None, None)
    planes = workspace(dev, _C.load().evk_conv2d_split_weight_bytes(ctypes.byref(d), for_dgrad))
    if h2:
        wb = _amax_buf(dev)          # (slot 0 = the maximum: a valid single word for the kernels' w_absmax)
        nel = d.Cout * d.kh * d.kw * d.Cin
        _C.call('evk_absmax', w_ptr, nel, wb.data_ptr(), weight_planes.absmax_workspace(dev, st).data_ptr(), st)
        _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), w_ptr, for_dgrad, planes.data_ptr(), wb.data_ptr(), st)
        return planes.data_ptr(), wb.data_ptr(), wb
    _C.call('evk_conv2d_split_weight', ctypes.byref(d), w_ptr, for_dgrad, planes.data_ptr(), st)
    return planes.data_ptr(), None, None


def _entry(x3_name):
    """C-ABI entry point of the current plane arithmetic: evk_*_x3 or its plain-bf16 twin evk_*_bf16."""
    return x3_name if _CONV_MATH != 'bf16' else x3_name.replace('_x3', '_bf16')


def set_conv_math(mode):
    global _CONV_MATH
    if mode not in _MATH_MODES:
        raise ValueError(f"conv math must be one of {_MATH_MODES}, got {mode!r}")
    prev, _CONV_MATH = _CONV_MATH, mode
    return prev


def get_conv_math():
    return _CONV_MATH


# BatchNorm statistics from the producing convolution's epilogue (EVK_BN_EPILOGUE=0: BatchNorm's own statistics pass)
_BN_EPILOGUE = os.environ.get('EVK_BN_EPILOGUE', '1') != '0'
# f16x2: activations that only convolutions read are stored already split ("packed", include/ever_hip.h:
# evk_pack_f16x2) by the BatchNorm pass that writes them (EVK_PACKED=0: fp32 everywhere, split while staging)
_PACKED = os.environ.get('EVK_PACKED', '1') != '0'


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


_WGRAD_BATCH = max(1, int(os.environ.get('EVK_WGRAD_BATCH', '1')))   # (8, 16, 32 measured: 524 vs 531 tiles/s for 1, same box)


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


def _dist_initialized():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


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


def _collectives_world():
    import torch.distributed as dist
    return dist.get_world_size() if _dist_initialized() else 1


def _mark_packed(t, bits):
    """`t` holds packed words of t / s (s from `bits`), not fp32: only the f16x2 convolution kernels may read it."""
    _note_amax(t, bits)
    t._evk_packed = (t._version, t.data_ptr())
    absmax_stats['packed'] += 1
    return t


_WGRAD_PACK_MARGIN = float(os.environ.get('EVK_WGRAD_PACK_MARGIN', '0.0'))
_WGRAD_PACK_GAIN = float(os.environ.get('EVK_WGRAD_PACK_GAIN', '1.0'))


def _wgrad_pack_pays(flops, x_elems, dy_elems):
    """A stand-alone evk_pack_f16x2 pass over the weight gradient's fp32 operand(s) first?  Measured (3x3x256 @128^2
    x16): both operands packed 1292 -> 944 us, i.e. ~27 % of a kernel that runs at ~260 TFLOP/s; a pass moves 8 bytes
    per element at ~5 TB/s.  Pays for the 3x3 convolutions of the FPN / decoder, not for 1x1 ones."""
    if x_elems + dy_elems == 0:
        return False
    if os.environ.get('EVK_WGRAD_PACK', '1') == '0':
        return False
    t_kernel = flops / 2.6e14
    t_pack = 8.0 * (x_elems + dy_elems) / 5.0e12 + 4e-6 * ((x_elems > 0) + (dy_elems > 0))
    # the im2col operand is two thirds of the staging work (256 of the 384 rows of a 128 x 256 tile)
    gain = _WGRAD_PACK_GAIN * ((0.18 if x_elems else 0.0) + (0.09 if dy_elems else 0.0))
    return gain * t_kernel - t_pack > _WGRAD_PACK_MARGIN * t_kernel


_WGRAD_TR = os.environ.get('EVK_WGRAD_TR', '1') != '0'
_WGRAD_TR_MIN_HW = int(os.environ.get('EVK_WGRAD_TR_MIN_HW', str(128 * 128)))


def _wgrad_planar_pays(d, need_db=False):
    """The planar-operand weight gradient (csrc/conv_wgrad_tr.hip: DMA + transposing LDS reads, nine-tap halo form) behind
    a stand-alone planar pack of both operands?  Measured (tools/ab_wgrad_tr.py, 3x3x256 x16): @128^2 966 -> 764 us, @64^2
    209 -> 200, @32^2 65 -> 72: it pays where the pixel reduction is long — 3x3 / stride 1 / padding 1 on maps of at least
    EVK_WGRAD_TR_MIN_HW pixels (FarSeg-R50 at 512^2: the FPN and decoder convolutions on the 128^2 maps; at 1024^2 also the
    256^2 ones), wide enough for its 128 x (9 x 64) tile."""
    return (_WGRAD_TR and _f16x2() and not need_db and d.kh == 3 and d.kw == 3 and d.stride_h == 1 and d.stride_w == 1
            and d.pad_h == 1 and d.pad_w == 1 and d.dil_h == 1 and d.dil_w == 1 and d.W % 32 == 0 and d.Cin % 64 == 0
            and d.Cout % 64 == 0 and d.Cout >= 128 and d.H * d.W >= _WGRAD_TR_MIN_HW
            and d.N * d.H * d.W * max(d.Cin, d.Cout) * 4 < 2 ** 31)


def _is_packed(t):
    hit = getattr(t, '_evk_packed', None)
    return hit is not None and hit[0] == t._version and hit[1] == t.data_ptr()


# ReLU bits (include/ever_hip.h: evk_bn_fwd_train_parts_bits): the BatchNorm + add + ReLU that ends a residual block keeps one
# bit per output element and its backward reads those instead of the output tensor; with `lazy_res` the identity branch's
# gradient is not written either — the unmasked incoming gradient travels on with the bits (`_evk_relu_bits` on a view of
# it) and is masked by its consumer: the block input's data-gradient launch while it adds, or the shortcut's BatchNorm.
_RELU_BITS = os.environ.get('EVK_RELU_BITS', '1') != '0'
_LAZY_RES = os.environ.get('EVK_LAZY_RES', '1') != '0'
relu_bits_stats = {'forward': 0, 'lazy': 0, 'masked_dgrad': 0, 'masked_bn': 0, 'materialized': 0}   # tests / tools


def _lazy_bits(t):
    """The ReLU bits an unmasked gradient travels with, or None."""
    hit = getattr(t, '_evk_relu_bits', None)
    if hit is not None and hit[0] == t._version and hit[1] == t.data_ptr():
        return hit[2]
    return None


def materialize_lazy(t):
    """`t` itself unless it is an unmasked gradient travelling with ReLU bits: then the masked tensor (one pass)."""
    bits = _lazy_bits(t) if t is not None else None
    if bits is None:
        return t
    out = torch.empty_like(t)
    _C.call('evk_relu_bits_apply', t.data_ptr(), bits.data_ptr(), out.data_ptr(), t.numel(), _stream())
    relu_bits_stats['materialized'] += 1
    return _inherit_amax(out, t)


_top_saved_hooks = getattr(torch._C._autograd, '_top_saved_tensors_default_hooks', None)


def observers_active():
    """True when something other than this package's own kernels may get to see an activation between its producer and
    its consumer: saved-tensor hooks (non-reentrant checkpointing, save_on_cpu: the unpack hook returns a NEW tensor
    object, which would not carry the `_evk_packed` mark) or global module forward hooks.  Packed tensors are raw words
    marked only by a Python attribute, so nothing is stored packed while an observer is installed; per-module
    hooks are checked by the layers (module/layers.py)."""
    if _top_saved_hooks is not None and _top_saved_hooks(True) is not None:
        return True
    from torch.nn.modules import module as _m
    return bool(_m._global_forward_hooks or _m._global_forward_pre_hooks or _m._global_forward_hooks_always_called)


def unpacked(t):
    """`t` as fp32 values: itself unless it holds packed f16x2 words, then (h + l) * s by evk_unpack_f16x2.  For readers
    outside the f16x2 convolution kernels (folded inference convolutions, debugging)."""
    if not _is_packed(t):
        return t
    bits = t._evk_amax[2]
    out = torch.empty_like(t)
    _C.call('evk_unpack_f16x2', t.data_ptr(), t.numel(), bits.data_ptr(), out.data_ptr(), _stream())
    _note_amax(out, bits)
    return out

_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """Raw handle of the current HIP stream of the current device.  ~250 calls per training step: the raw getter skips
    the Stream-object construction of torch.cuda.current_stream() (2 ms of host time per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _timed_call(family, nbytes, name, *args):
    """_C.call bracketed by HIP events when bench.py's KernelTimer is active (HBM-bound families:
    `nbytes` = algorithmic bytes of the call)."""
    sp = timing.span(family, 0.0, nbytes)
    _C.call(name, *args)
    if sp is not None:
        sp.stop()


def _require_cuda(t, what):
    if not t.is_cuda:
        raise HipPathError(
            f'{what}: ever_amd kernels run on MI355X only (got a {t.device} tensor). '
            f'There is no CPU fallback; move the model and data to cuda.')
    if t.dtype != torch.float32:
        raise HipPathError(f'{what}: fp32 tensors required, got {t.dtype}')


def empty_nhwc(n, c, h, w, device, dtype=torch.float32):
    """Logical [n,c,h,w] tensor over dense NHWC memory."""
    return torch.empty((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


def is_nhwc(t):
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def as_nhwc(t, what='tensor'):
    """Return `t` if its memory is already dense NHWC, else transpose it with the HIP kernel."""
    _require_cuda(t, what)
    if is_nhwc(t):
        return t
    return _ToNHWC.apply(t)


class _ToNHWC(Function):
    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        xc = x.contiguous()  # dense NCHW source
        out = empty_nhwc(n, c, h, w, x.device)
        _C.call('evk_nchw_to_nhwc', xc.data_ptr(), out.data_ptr(), n, c, h, w, c, _stream())
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g


def image_to_nhwc(x, cpad):
    """Model-boundary transpose: NCHW image -> NHWC with channels zero-padded to `cpad` (no grad)."""
    _require_cuda(x, 'image_to_nhwc')
    n, c, h, w = x.shape
    out = empty_nhwc(n, cpad, h, w, x.device)
    if is_nhwc(x):
        _C.call('evk_pad_channels', x.data_ptr(), out.data_ptr(), n * h * w, c, cpad, _stream())
    else:
        xc = x.contiguous()
        _C.call('evk_nchw_to_nhwc', xc.data_ptr(), out.data_ptr(), n, c, h, w, cpad, _stream())
    return out


# ------------------------------------------------------------------------------------ convolution
def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _conv_desc(n, h, w, cin, cout, kh, kw, stride, padding, dilation):
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    ho = (h + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (w + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    return _C.ConvDesc(n, h, w, cin, ho, wo, cout, kh, kw, sh, sw, ph, pw, dh, dw)


def _pad4(c):
    return (c + 3) // 4 * 4


def _pad_last(t2d_ptr, rows, c, cp, device):
    out = torch.empty((rows, cp), device=device, dtype=torch.float32)
    _C.call('evk_pad_channels', t2d_ptr, out.data_ptr(), rows, c, cp, _stream())
    return out


def _weight_ohwi(weight):
    """Weight memory as dense [O][kh][kw][I]; transposes with the HIP kernel if it is OIHW-dense."""
    if is_nhwc(weight):
        return weight
    o, i, kh, kw = weight.shape
    wc = weight.contiguous()
    out = empty_nhwc(o, i, kh, kw, weight.device)
    _C.call('evk_nchw_to_nhwc', wc.data_ptr(), out.data_ptr(), o, i, kh, kw, i, _stream())
    return out


class _ConvState:
    """What one convolution's backward needs.  The descriptor and flags stay on the autograd ctx; the tensors
    (xk, y, weight, w_ohwi) travel through ctx.save_for_backward (`_stash` / `_unstash`), so that the output saved on
    its own node forms no reference cycle, in-place writes to a saved tensor are caught by the version check, and
    saved-tensor hooks (activation checkpointing, offloading) see them."""
    __slots__ = ('desc', 'relu', 'cin', 'has_bias', 'flops', 'abytes', 'w_stride', 'xk', 'w_ohwi', 'y', 'weight',
                 'w_alias', 'scope', 'bn_parts', 'bias_leaf')


def _stash(states):
    """Tensors of the given conv states, flattened for save_for_backward; the states keep only metadata."""
    out = []
    for cs in states:
        cs.w_alias = cs.w_ohwi is None or cs.w_ohwi.data_ptr() == cs.weight.data_ptr()
        out += [cs.xk, cs.y, cs.weight, None if cs.w_alias else cs.w_ohwi]
        cs.xk = cs.y = cs.weight = cs.w_ohwi = None
    return out


def _unstash(states, saved):
    for i, cs in enumerate(states):
        cs.xk, cs.y, cs.weight, w = saved[4 * i:4 * i + 4]
        cs.w_ohwi = cs.weight.detach() if cs.w_alias else w


def _drop(states):
    """The backward is done with the tensors `_unstash` put back on the states: let go of them.  cs.y is the node's own
    output (y -> grad_fn -> ctx -> cs -> y), and a node kept alive by such a cycle keeps its whole upstream graph —
    every other state's xk / y — allocated until the cyclic collector runs: 2.8 GB per step on FarSeg-R50, and a
    caching allocator that has to grow (hipMalloc inside the step) whenever the collector is late."""
    for cs in states:
        cs.xk = cs.y = cs.weight = cs.w_ohwi = cs.bias_leaf = None


def _conv_forward(x, weight, bias, stride, padding, dilation, relu, want_stats=False):
    """evk_conv2d_fwd on NHWC x / OHWI weight -> (y, _ConvState).  want_stats: also the BatchNorm partial statistics of
    y from the epilogue (cs.bn_parts = (records tensor, count) or None when this shape's kernel cannot)."""
    n, cin, h, w = x.shape
    cout, cin_w, kh, kw = weight.shape
    if cin_w != cin:
        raise ValueError(f'conv2d: input has {cin} channels but weight expects {cin_w} (groups != 1 unsupported)')
    dev = x.device
    st = _stream()
    w_ohwi = _weight_ohwi(weight.detach())
    cin_p = _pad4(cin)
    if cin_p != cin:
        xk = _pad_last(x.data_ptr(), n * h * w, cin, cin_p, dev)
        wk = _pad_last(w_ohwi.data_ptr(), cout * kh * kw, cin, cin_p, dev)
        x_ptr, w_ptr = xk.data_ptr(), wk.data_ptr()
    else:
        xk = x
        x_ptr, w_ptr = x.data_ptr(), w_ohwi.data_ptr()
    d = _conv_desc(n, h, w, cin_p, cout, kh, kw, stride, padding, dilation)
    x_pk = _is_packed(x)        # written packed by the BatchNorm pass before this convolution (EVK_BN_PACK_Y)
    if x_pk and not (_f16x2() and cin_p == cin and cin % 8 == 0 and n * d.Ho * d.Wo > 32):
        raise HipPathError('conv2d: a packed activation reached a convolution that cannot read it '
                           f'(math {_CONV_MATH}, Cin {cin}, {n * d.Ho * d.Wo} output rows)')
    y = empty_nhwc(n, cout, d.Ho, d.Wo, dev)
    cs = _ConvState()
    cs.scope = timing.current_scope()
    cs.flops = 2.0 * n * d.Ho * d.Wo * cout * cin * kh * kw  # algorithmic (un-padded) FLOPs
    # algorithmic bytes: input + output + weights, each touched once
    cs.abytes = 4.0 * (n * h * w * cin + n * d.Ho * d.Wo * cout + cout * cin * kh * kw)
    # a handful of GEMM rows (the scene-embedding 1x1 convolutions on 1x1 maps, M = batch): the fp32 path has a
    # dedicated weight-streaming kernel for M <= 32; an MFMA tile would run K = 2048 serially on two workgroups
    small_m = n * d.Ho * d.Wo <= 32
    bn_parts = None
    if _planes_math() and cin_p == cin and cin % 8 == 0 and not small_m:
        # weights -> planes (two fp16 of w / s, or three bf16), then the split-MFMA kernel.  The planes of every registered weight are
        # refreshed by one launch per weight update (weight_planes); a weight the cache cannot follow (a transient
        # re-laid-out copy) is split into the shared workspace on every call.
        pl_ptr, wabs_ptr, _keep = _weight_planes(weight, w_ohwi, w_ptr, d, 0, st, dev)
        xbits = absmax_bits(x, st) if wabs_ptr is not None else None
        sp = timing.span('conv_igemm', cs.flops, cs.abytes)
        stats = want_stats and _BN_EPILOGUE and not relu and cout % 4 == 0
        parts, cap, nparts = None, 0, ctypes.c_int32(0)
        if stats:
            cap = int(_C.load().evk_conv2d_stats_max_parts(ctypes.byref(d)))
            parts = torch.empty((cap * 3 * cout,), device=dev, dtype=torch.float32)
        if wabs_ptr is not None:
            # an output that no BatchNorm will normalise is (mostly) another convolution's operand: its scale from here
            ybits = None if stats else _amax_zeroed(dev)
            _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), x_ptr, xbits.data_ptr(), pl_ptr, wabs_ptr, _ptr(bias), None,
                    y.data_ptr(), (1 if relu else 0) | (2 if x_pk else 0), _ptr(parts), cap, ctypes.byref(nparts),
                    _ptr(ybits), st)
            if ybits is not None:
                _note_amax(y, ybits)
        elif _CONV_MATH == 'bf16':
            _C.call('evk_conv2d_fwd_bf16', ctypes.byref(d), x_ptr, pl_ptr, _ptr(bias), y.data_ptr(), 1 if relu else 0,
                    _ptr(parts), cap, ctypes.byref(nparts), st)
        elif stats:
            _C.call('evk_conv2d_fwd_x3_stats', ctypes.byref(d), x_ptr, pl_ptr, _ptr(bias), y.data_ptr(), 0,
                    parts.data_ptr(), cap, ctypes.byref(nparts), st)
        else:
            _C.call('evk_conv2d_fwd_x3', ctypes.byref(d), x_ptr, pl_ptr, _ptr(bias), y.data_ptr(),
                    1 if relu else 0, st)
        if nparts.value > 0:
            # third field: this convolution's backward takes its dy packed (the BatchNorm that consumes the records is
            # the ONLY reader of y — conv2d(bn_stats=True)'s contract — so its dx has no other reader either)
            # (the planar weight gradient packs both operands itself: its BatchNorm's dx stays fp32)
            bn_parts = (parts, int(nparts.value),
                        _PACKED and wabs_ptr is not None and bias is None and cout % 8 == 0 and not observers_active()
                        and not _wgrad_planar_pays(d))
    else:
        sp = timing.span('conv_igemm_f32', cs.flops, cs.abytes)
        _C.call('evk_conv2d_fwd', ctypes.byref(d), x_ptr, w_ptr, _ptr(bias), y.data_ptr(), 1 if relu else 0, st)
    if sp is not None:
        sp.stop()
    cs.desc, cs.relu, cs.cin, cs.has_bias = d, relu, cin, bias is not None
    cs.bias_leaf = bias      # (the parameter itself, not saved for backward: the weight-gradient side stream's rules look at it)
    cs.w_stride = tuple(weight.stride())
    # xk (channel-padded copy when Cin % 4 != 0) is what wgrad reads
    cs.xk, cs.w_ohwi, cs.y, cs.weight = xk, w_ohwi, (y if relu else None), weight
    cs.bn_parts = bn_parts
    return y, cs


def _sparse_dgrad(cs):
    """True when the data gradient of this convolution reaches only some pixels of x (kernel smaller than the stride:
    the 1x1 / stride-2 shortcut of a residual block) and runs on the plane kernels, which accumulate in place."""
    d = cs.desc
    return (_planes_math() and (d.kh < d.stride_h or d.kw < d.stride_w) and d.Cin % 8 == 0 and d.Cout % 8 == 0
            and cs.cin == d.Cin)


def _conv_backward(cs, dy, need_dx, need_dw, need_db, accum=None, inplace=False, accum_bits=None):
    """(dx, dw, db) of one convolution.  `accum` (a tensor of x's shape or None) is added to dx inside
    the data-gradient epilogue: dx = conv_transpose(dy, w) + accum; with `inplace` (plane kernels only) dx IS accum."""
    d, xk, w_ohwi = cs.desc, cs.xk, cs.w_ohwi
    dev = dy.device
    st = _stream()
    n, cin_p, cout, kh, kw = d.N, d.Cin, d.Cout, d.kh, d.kw
    cin = cs.cin
    dy = materialize_lazy(dy)       # (an unmasked gradient with ReLU bits is only understood as `accum`)
    if accum_bits is not None and not (need_dx and _f16x2() and d.stride_h == 1 and d.stride_w == 1 and not inplace
                                       and cin_p == cin and cin % 8 == 0 and cout % 8 == 0):
        accum, accum_bits = materialize_lazy(accum), None
    dy = as_nhwc(dy, 'conv2d.backward')
    dy_pk = _is_packed(dy)      # written packed by the BatchNorm backward that follows this convolution
    if dy_pk and (cs.relu or need_db or not _f16x2() or d.Cout % 8 or cs.cin != d.Cin):
        raise HipPathError('conv2d.backward: a packed output gradient reached a convolution that cannot read it')
    if cs.relu:
        g = torch.empty_like(dy)
        _C.call('evk_relu_bwd', dy.data_ptr(), cs.y.data_ptr(), g.data_ptr(), dy.numel(), st)
        dy = g
    rows_o = n * d.Ho * d.Wo
    x3 = _planes_math()
    # narrow heads (classifier Cout = 1..7): pad dy / weight rows — to 8 output channels under the split arithmetic, so
    # that the data gradient stays on the split-MFMA kernels (its reduction is over taps x Cout), else to 4
    narrow8 = x3 and cin_p == cin and cout % 8 != 0 and cout < 8 and os.environ.get('EVK_NARROW_X3', '1') != '0'
    cout_p = 8 if narrow8 else _pad4(cout)
    if cout_p != cout:
        dyk = _pad_last(dy.data_ptr(), rows_o, cout, cout_p, dev)
        dy_ptr = dyk.data_ptr()
    else:
        dyk = dy
        dy_ptr = dy.data_ptr()
    dk = _C.ConvDesc(d.N, d.H, d.W, cin_p, d.Ho, d.Wo, cout_p, kh, kw, d.stride_h, d.stride_w, d.pad_h, d.pad_w,
                     d.dil_h, d.dil_w)
    dx = dw = db = None
    taps = kh * kw

    def _dgrad():
        nonlocal dx, accum, accum_bits
        if need_dx and x3 and cin_p == cin and (narrow8 or (cout_p == cout and cout % 8 == 0)):
            if narrow8:     # zero rows appended to the (tiny) weight: a transient copy, split on every call
                w_src = torch.zeros((cout_p, taps, cin), device=dev, dtype=torch.float32)
                w_src[:cout].copy_(w_ohwi.permute(0, 2, 3, 1).reshape(cout, taps, cin))
            else:
                w_src = w_ohwi
            pl_ptr, wabs_ptr, _keep = _weight_planes(cs.weight, w_src, w_src.data_ptr(), dk, 1, st, dev)
            acc_ptr = None
            if accum is not None:
                accum = as_nhwc(accum, 'conv2d.backward.accum')
                acc_ptr = accum.data_ptr()
            dx = accum if (inplace and accum is not None) else empty_nhwc(n, cin, d.H, d.W, dev)
            dybits = absmax_bits(dyk, st) if wabs_ptr is not None else None
            sp = timing.span('conv_igemm', cs.flops, cs.abytes, cs.scope)
            if wabs_ptr is not None:
                # in place: the slots the main branch's launch raised stay (an upper bound is all a scale needs)
                hit = getattr(dx, '_evk_amax', None) if inplace else None
                dxbits = hit[2] if hit is not None else _amax_zeroed(dev)
                if accum_bits is not None and acc_ptr is not None:
                    relu_bits_stats['masked_dgrad'] += 1
                    _C.call('evk_conv2d_dgrad_f16x2_masked', ctypes.byref(dk), dy_ptr, dybits.data_ptr(), pl_ptr, wabs_ptr,
                            acc_ptr, accum_bits.data_ptr(), dx.data_ptr(), _ptr(dxbits), 4 if dy_pk else 0, st)
                else:
                    _C.call('evk_conv2d_dgrad_f16x2_ex', ctypes.byref(dk), dy_ptr, dybits.data_ptr(), pl_ptr, wabs_ptr, acc_ptr,
                            dx.data_ptr(), _ptr(dxbits), 4 if dy_pk else 0, st)
                if dxbits is not None:
                    _note_amax(dx, dxbits)
            else:
                _C.call(_entry('evk_conv2d_dgrad_x3'), ctypes.byref(dk), dy_ptr, pl_ptr, acc_ptr, dx.data_ptr(), st)
            if sp is not None:
                sp.stop()
        elif need_dx:
            # weights as [cout_p][taps][cin_p]
            if cin_p != cin or cout_p != cout:
                wfull = torch.zeros((cout_p, taps, cin_p), device=dev, dtype=torch.float32)
                tmp = (_pad_last(w_ohwi.data_ptr(), cout * taps, cin, cin_p, dev) if cin_p != cin
                       else w_ohwi.permute(0, 2, 3, 1))  # OHWI memory order
                wfull[:cout].copy_(tmp.reshape(cout, taps, cin_p))
                w_src = wfull
            else:
                w_src = w_ohwi
            wt = torch.empty((cin_p, taps, cout_p), device=dev, dtype=torch.float32)
            _C.call('evk_conv2d_pack_dgrad_weight', ctypes.byref(dk), w_src.data_ptr(), wt.data_ptr(), st)
            acc_ptr = None
            if accum is not None:
                if cin_p != cin:
                    raise HipPathError('conv2d.backward: accum with channel-padded inputs is not supported')
                accum = as_nhwc(accum, 'conv2d.backward.accum')
                acc_ptr = accum.data_ptr()
            dxk = empty_nhwc(n, cin_p, d.H, d.W, dev)
            sp = timing.span('conv_igemm_f32', cs.flops, cs.abytes, cs.scope)
            _C.call('evk_conv2d_dgrad', ctypes.byref(dk), dy_ptr, wt.data_ptr(), acc_ptr, dxk.data_ptr(), st)
            if sp is not None:
                sp.stop()
            if cin_p != cin:
                dx = empty_nhwc(n, cin, d.H, d.W, dev)
                _C.call('evk_unpad_channels', dxk.data_ptr(), dx.data_ptr(), n * d.H * d.W, cin_p, cin, st)
            else:
                dx = dxk

    def _wgrad():
        nonlocal dw, db
        if need_dw or need_db:
            lib = _C.load()
            ws_bytes = (lib.evk_conv2d_wgrad_x3_workspace_bytes if x3 else lib.evk_conv2d_wgrad_workspace_bytes)(
                ctypes.byref(dk))
            dwk = torch.empty((cout_p, taps, cin_p), device=dev, dtype=torch.float32)
            dbk = torch.empty((cout_p,), device=dev, dtype=torch.float32) if need_db else None
            # (the gradient comes in OHWI memory order: a parameter laid out otherwise gets a deep copy from AccumulateGrad —
            # a read of dw on the backward's stream — so its weight gradient stays there)
            wstr = cs.w_stride
            contract = tuple(wstr) == (taps * cin, 1, kw * cin, cin) or (taps == 1 and wstr[0] == cin and wstr[1] == 1)
            side = _wgrad_side_stream(dev, cs.weight, cs.bias_leaf if need_db else None) if contract else None

            # with BATCHED side-stream launches (graph capture: EVK_WGRAD_BATCH = 32) the closure would look at host-side
            # tensor state — the operand-scale caches, the packed flag — up to 32 layers after this layer's backward ran; what it
            # needs is resolved here, when the launch is queued.  (A batch of one runs the closure right away: nothing to resolve,
            # and a missing scale is then computed on the side stream, off the backward's chain.)
            pre = None
            if side is not None and _WGRAD_BATCH > 1 and x3 and _f16x2():
                pre = (absmax_bits(xk, st), absmax_bits(dyk, st), _is_packed(xk))

            # the half-chip split-K plan is a property of the CONFIGURATION (side stream enabled + split on), not of the stream
            # this launch ends up on (the probe that picks the side stream is a timing measurement, and a launch that
            # fell back to the backward's stream used to take the other plan — other bits for the same model and switches)
            shared = _WGRAD_SHARED if (_WGRAD_STREAM[0] and _WGRAD_SHARED_ON[0]) else 0

            def launch(st):
                """the weight-gradient launches of this layer on stream `st` (torch's current stream when this runs)"""
                ws = workspace(dev, ws_bytes)
                h2 = x3 and _f16x2()
                dy_pk_ = dy_pk
                sp = timing.span('conv_wgrad' if x3 else 'conv_wgrad_f32', cs.flops, cs.abytes, cs.scope)
                if h2:
                    if pre is not None:
                        xbits, dybits, x_pk = pre
                    else:
                        xbits, dybits, x_pk = absmax_bits(xk, st), absmax_bits(dyk, st), _is_packed(xk)
                    if side is not None:     # (slices of a pooled buffer of the main stream: keep the pool block until this has run)
                        _wgrad_hold(xbits, dybits)
                    xw_ptr, dyw_ptr, _tmp = xk.data_ptr(), dy_ptr, None
                    planar = 0
                    if cout_p == cout and cin_p == cin and not x_pk and not dy_pk_ and _wgrad_planar_pays(dk, need_db):
                        xq, dq = torch.empty_like(xk), torch.empty_like(dyk)
                        _C.call('evk_pack_planar_f16x2', xk.data_ptr(), xk.numel(), xbits.data_ptr(), xq.data_ptr(), st)
                        _C.call('evk_pack_planar_f16x2', dy_ptr, dyk.numel(), dybits.data_ptr(), dq.data_ptr(), st)
                        xw_ptr, dyw_ptr, _tmp, planar = xq.data_ptr(), dq.data_ptr(), [xq, dq], 8 | 16
                    elif _PACKED and not need_db and cout_p == cout and _wgrad_pack_pays(cs.flops, 0 if x_pk else xk.numel(),
                                                                                     0 if dy_pk_ else dyk.numel()):
                        # the kernel's bound is the split of its operands while staging (each element is staged by many
                        # workgroups): where the matrix work per byte is high, one streaming pass that stores them split first
                        _tmp = []
                        if not x_pk:
                            xp = torch.empty_like(xk)
                            _C.call('evk_pack_f16x2', xk.data_ptr(), xk.numel(), xbits.data_ptr(), xp.data_ptr(), st)
                            xw_ptr, x_pk = xp.data_ptr(), True
                            _tmp.append(xp)
                        if not dy_pk_:
                            dp = torch.empty_like(dyk)
                            _C.call('evk_pack_f16x2', dy_ptr, dyk.numel(), dybits.data_ptr(), dp.data_ptr(), st)
                            dyw_ptr, dy_pk_ = dp.data_ptr(), True
                            _tmp.append(dp)
                    _C.call('evk_conv2d_wgrad_f16x2_ex', ctypes.byref(dk), xw_ptr, xbits.data_ptr(), dyw_ptr, dybits.data_ptr(),
                            dwk.data_ptr(), _ptr(dbk), ws.data_ptr(), ws_bytes,
                            (planar if planar else ((2 if x_pk else 0) | (4 if dy_pk_ else 0))) | shared, st)
                else:
                    _C.call(_entry('evk_conv2d_wgrad_x3') if x3 else 'evk_conv2d_wgrad', ctypes.byref(dk), xk.data_ptr(), dy_ptr,
                            dwk.data_ptr(), _ptr(dbk), ws.data_ptr(), ws_bytes, st)
                if sp is not None:
                    sp.stop()
                if dw2 is not None:
                    _C.call('evk_unpad_channels', dwk.data_ptr(), dw2.data_ptr(), cout_p * taps, cin_p, cin, st)

            # the tensors autograd gets are fixed now; the launches that fill them may come later (side stream, in batches)
            dw2 = torch.empty((cout_p * taps, cin), device=dev, dtype=torch.float32) if (need_dw and cin_p != cin) else None
            if need_dw:
                dwv = dw2.reshape(cout_p, taps, cin) if dw2 is not None else dwk
                # logical OIHW view over OHWI memory (matches a channels_last parameter)
                dw = dwv[:cout].reshape(cout, kh, kw, cin).permute(0, 3, 1, 2)
                wstr = cs.w_stride
                if kh * kw == 1 and dw.stride() != wstr and wstr[0] == cin and wstr[1] == 1:
                    # 1x1 kernels: the size-1 dims make the stride tuple ambiguous; present exactly the
                    # parameter's strides so AccumulateGrad / DDP bucket views alias instead of copying
                    dw = dw.as_strided(dw.shape, wstr)
            if need_db:
                db = dbk[:cout]
            if side is None:
                launch(st)
            else:
                # the weight gradient beside the rest of the backward (see _WGRAD_STREAM above).  Its launches are QUEUED and
                # issued in batches: one fork event, one switch of torch's current stream and back per batch instead of per
                # layer (host time), and a captured step has a handful of edges between its two branches instead of 2 x 53
                _wgrad_hold(xk, dyk, dwk, dbk, dw2)
                _WGRAD_QUEUE.setdefault(dev, []).append(launch)
                if need_dw:              # what AccumulateGrad has to store as it is (checked at the end of the pass)
                    _WGRAD_OWNED[id(cs.weight)] = (cs.weight, dw.untyped_storage().data_ptr())
                if need_db:
                    _WGRAD_OWNED[id(cs.bias_leaf)] = (cs.bias_leaf, db.untyped_storage().data_ptr())
                if len(_WGRAD_QUEUE.get(dev, ())) >= _WGRAD_BATCH:
                    flush_wgrad_queue(dev)

    # EVK_WGRAD_EARLY=1 (A/B): the weight gradient is forked BEFORE the data gradient is enqueued, so that it may start
    # beside it instead of behind it; the operand scale of dy is fixed first (both read it, from different streams)
    # EVK_WGRAD_EARLY=2 (round 5): only the layers whose weight gradient is the nine-tap planar kernel (3x3x256 on the 128^2
    # maps: one 686 us workgroup per CU at 244 registers x 2 waves per SIMD — NOTHING co-resides with it).  Forked behind its
    # data gradient it runs beside the short HBM-bound kernels that follow on the backward's stream and holds every one of them
    # off the chip (profiles/r05_stream_timeline.txt: a 5 us finalisation kernel "running" 180 us); forked in front, it time-slices
    # with its own layer's 780 us halo data gradient — two long matrix-bound kernels, where nothing short waits.
    early = _WGRAD_EARLY_MODE == 1 or (_WGRAD_EARLY_MODE == 2 and need_dw and cout_p == cout and cin_p == cin and not dy_pk
                                       and _wgrad_planar_pays(dk, need_db))
    if early and need_dx and (need_dw or need_db) and _f16x2() and x3:
        absmax_bits(dyk, st)
        _wgrad()
        _dgrad()
    else:
        _dgrad()
        _wgrad()
    return dx, dw, db


class _Conv2dFn(Function):
    """nn.Conv2d forward/backward on the MFMA implicit-GEMM kernels.

    Replaces aten::convolution(+_backward) at reference ever/module/_resnets.py:21-29,149,
    fpn.py:23-37,72-73,165,179 and fs_relation.py:23-53.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, relu, want_stats=False, slot=None):
        y, cs = _conv_forward(x, weight, bias, stride, padding, dilation, relu, want_stats)
        if any(ctx.needs_input_grad):
            _note_param_use(weight, bias)
        ctx.cs = cs
        ctx.slot = slot
        ctx.save_for_backward(*_stash([cs]))
        _BN_HANDOFF[0] = cs.bn_parts      # picked up by conv2d() right after apply (same thread, no autograd in between)
        cs.bn_parts = None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        cs = ctx.cs
        _unstash([cs], ctx.saved_tensors)
        acc = None
        if ctx.slot is not None:     # a later consumer's gradient of x, parked by _SlotOutFn: summed in the epilogue below
            ctx.slot.consumed = True
            acc, ctx.slot.grad = ctx.slot.grad, None
        try:
            if acc is not None and not ctx.needs_input_grad[0]:
                acc = None
            dx, dw, db = _conv_backward(cs, dy, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                        cs.has_bias and ctx.needs_input_grad[2], accum=acc)
        finally:
            _drop([cs])
        return dx, dw, db, None, None, None, None, None, None


_BN_HANDOFF = [None, None]   # statistics records of the convolution(s) that just ran: [main, shortcut]


def _attach_parts(y, parts):
    """BatchNorm partial statistics ride on the tensor object to the BatchNorm that consumes it next."""
    if parts is not None:
        y._evk_bn_parts = parts
    return y


class _GroupDenseFn(Function):
    """grouped weight [Cout, Cin/g, kh, kw] -> block-diagonal dense [Cout, Cin, kh, kw] (include/ever_hip.h:
    evk_group_weight_expand); backward gathers the diagonal blocks of the dense gradient"""

    @staticmethod
    def forward(ctx, weight, groups):
        w = _weight_ohwi(weight.detach())
        cout, cpg, kh, kw = weight.shape
        dense = empty_nhwc(cout, cpg * groups, kh, kw, weight.device)
        _C.call('evk_group_weight_expand', w.data_ptr(), dense.data_ptr(), cout, kh * kw, cpg * groups, groups, _stream())
        ctx.groups, ctx.shape = groups, tuple(weight.shape)
        return dense

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = _weight_ohwi(g)
        cout, cpg, kh, kw = ctx.shape
        dw = empty_nhwc(cout, cpg, kh, kw, g.device)
        _C.call('evk_group_weight_gather', g.data_ptr(), dw.data_ptr(), cout, kh * kw, cpg * ctx.groups, ctx.groups, _stream())
        return dw, None


def grouped_dense_weight(weight, groups):
    """The dense weight a grouped convolution (reference _resnets.py:21-24, ResNeXt) runs with: exact zeros outside the
    groups, so every dense kernel computes the grouped convolution; differentiable w.r.t. `weight`."""
    if groups == 1:
        return weight
    _require_cuda(weight, 'grouped convolution weight')
    dense = _GroupDenseFn.apply(weight, int(groups))
    # a fresh tensor every call: not a weight the plane cache should register (a new slot, planes allocation and job table
    # per step — ADVICE r3); the convolution splits it into the shared workspace instead (weight_planes.planes_for)
    dense._evk_transient = True
    return dense


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, relu=False, bn_stats=False, grad_slot=None):
    """bn_stats=True: the caller applies a training-mode BatchNorm to the result next; where the kernel can, the
    epilogue leaves that BatchNorm's partial statistics on the returned tensor (`_evk_bn_parts`) and
    batch_norm_act() skips its own statistics pass over it.
    grad_slot: a GradSlot this convolution CLAIMS — x has a second, LATER consumer that was handed slot_output(x, slot); its
    gradient is parked there (backward runs it first) and added inside this convolution's data-gradient epilogue."""
    _require_cuda(x, 'conv2d')
    x = as_nhwc(x, 'conv2d')
    _BN_HANDOFF[0] = None
    if grad_slot is not None:
        if x.shape[1] % 8 or not _planes_math() or not (torch.is_grad_enabled() and x.requires_grad) or grad_slot.claimed:
            grad_slot = None        # (no accumulate epilogue on this path, or nothing to differentiate)
        else:
            grad_slot.claimed = True
    y = _Conv2dFn.apply(x, weight, bias, _pair(stride), _pair(padding), _pair(dilation), bool(relu), bool(bn_stats), grad_slot)
    parts, _BN_HANDOFF[0] = _BN_HANDOFF[0], None
    return _attach_parts(y, parts)


class GradSlot:
    """A gradient handed from a LATER consumer of a tensor to an EARLIER one, so that the sum of the two input
    gradients happens inside a data-gradient kernel's epilogue instead of autograd's own add pass.

    ResNetEncoder's stage output c_i feeds the next stage's first block (a `conv2d_fork` node) AND, later in the
    forward, the head (FPN lateral).  Backward runs the head first: its gradient w.r.t. c_i is parked here
    (`_SlotOutFn`), and the fork node — which runs afterwards — feeds it to its first data-gradient launch as `accum`.
    Safety: a slot is only handed to the head if a fork node CLAIMED it during the forward, and parking a gradient in
    a slot whose claimant has already run raises instead of dropping the gradient."""
    __slots__ = ('grad', 'claimed', 'consumed')

    def __init__(self):
        self.grad, self.claimed, self.consumed = None, False, False


class _SlotOutFn(Function):
    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        return x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        slot = ctx.slot
        if slot.consumed or slot.grad is not None:
            raise RuntimeError('gradient slot: the claiming convolution ran its backward before this gradient arrived '
                               '(or backward ran twice on one graph); disable with EVK_GRAD_SLOTS=0')
        slot.grad = as_nhwc(g, 'grad slot')
        return None, None


def slot_output(x, slot):
    """The view of `x` to hand to the later consumer: its gradient goes to `slot` instead of to autograd's sum."""
    out = _SlotOutFn.apply(x, slot)
    hit = getattr(x, '_evk_amax', None)          # the same values under another tensor object: keep the operand scale
    if hit is not None and hit[0] == x._version and out.data_ptr() == x.data_ptr():
        _note_amax(out, hit[2])
    return out


def grad_slots_enabled():
    return os.environ.get('EVK_GRAD_SLOTS', '1') != '0'


class _ConvForkFn(Function):
    """Two consumers of one tensor in ONE autograd node: x feeds conv_main AND a second branch — a residual block's
    shortcut (identity, or the down-sampling 1x1 conv; reference _resnets.py:52-69, 92-112, 188-192) or a sibling
    convolution (FS-Relation's content / re-encode pair on each pyramid level, fs_relation.py:41-52, 60-63).  Seeing
    both consumers lets the backward fold the sum of the two input gradients into the epilogue of the last
    data-gradient kernel (dx = dgrad(dy_main) + d_other) instead of a separate add pass over x."""

    @staticmethod
    def forward(ctx, x, w_main, w_short, b_main, b_short, cfg_main, cfg_short, slot=None, want_stats=(False, False)):
        y, cs = _conv_forward(x, w_main, b_main, *cfg_main, False, want_stats[0])
        if any(ctx.needs_input_grad):
            _note_param_use(w_main, b_main, w_short, b_short)
        ctx.cs_main = cs
        ctx.slot = slot
        _BN_HANDOFF[0], _BN_HANDOFF[1] = cs.bn_parts, None
        cs.bn_parts = None
        if w_short is None:
            ctx.cs_short = None
            ctx.save_for_backward(*_stash([cs]))
            return y, x.view_as(x)
        ys, css = _conv_forward(x, w_short, b_short, *cfg_short, False, want_stats[1])
        ctx.cs_short = css
        _BN_HANDOFF[1] = css.bn_parts
        css.bn_parts = None
        ctx.save_for_backward(*_stash([cs, css]))
        return y, ys

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, dshort):
        states = [ctx.cs_main] if ctx.cs_short is None else [ctx.cs_main, ctx.cs_short]
        _unstash(states, ctx.saved_tensors)
        try:
            return _ConvForkFn._backward(ctx, dy, dshort)
        finally:
            _drop(states)

    @staticmethod
    def _backward(ctx, dy, dshort):
        need_dx = ctx.needs_input_grad[0]
        dws = dbs = None
        acc = None
        slot_g = None
        if ctx.slot is not None:   # a later consumer's gradient of x, parked by _SlotOutFn (the head ran first)
            ctx.slot.consumed = True
            slot_g, ctx.slot.grad = ctx.slot.grad, None
            if not need_dx:
                slot_g = None
        acc_bits = None
        if ctx.cs_short is None:
            acc = dshort  # gradient of the identity shortcut (possibly unmasked, with the block's ReLU bits)
            acc_bits = _lazy_bits(acc) if acc is not None else None
        elif dshort is not None and dy is not None and need_dx and _sparse_dgrad(ctx.cs_short):
            # strided 1x1 shortcut: the main branch's (dense) data gradient first, the shortcut's one-pixel-in-four
            # contribution accumulated into it in place — no zero fill / copy of the whole tensor for the other three
            dx, dw, db = _conv_backward(ctx.cs_main, dy, True, ctx.needs_input_grad[1],
                                        ctx.cs_main.has_bias and ctx.needs_input_grad[3], accum=slot_g)
            dx, dws, dbs = _conv_backward(ctx.cs_short, dshort, True, ctx.needs_input_grad[2],
                                          ctx.cs_short.has_bias and ctx.needs_input_grad[4], accum=dx, inplace=True)
            return dx, dw, dws, db, dbs, None, None, None, None
        elif dshort is not None:
            acc, dws, dbs = _conv_backward(ctx.cs_short, dshort, need_dx, ctx.needs_input_grad[2],
                                           ctx.cs_short.has_bias and ctx.needs_input_grad[4], accum=slot_g)
            slot_g = None
        if slot_g is not None:     # the shortcut convolution did not run: plain sum
            acc, acc_bits = (slot_g if acc is None else add(materialize_lazy(acc), slot_g)), None
        if dy is None:   # only the second branch reached the loss
            return materialize_lazy(acc), None, dws, None, dbs, None, None, None, None
        dx, dw, db = _conv_backward(ctx.cs_main, dy, need_dx, ctx.needs_input_grad[1],
                                    ctx.cs_main.has_bias and ctx.needs_input_grad[3],
                                    accum=acc if need_dx else None, accum_bits=acc_bits if need_dx else None)
        return dx, dw, dws, db, dbs, None, None, None, None


def conv2d_fork(x, conv_main, conv_short=None, bn_stats=(False, False)):
    """(conv_main(x), conv_short(x) or x) with a fused input-gradient sum.  bn_stats: see conv2d()."""
    _require_cuda(x, 'conv2d_fork')
    x = as_nhwc(x, 'conv2d_fork')
    if x.shape[1] % 4:
        y = conv2d(x, conv_main.weight, conv_main.bias, conv_main.stride, conv_main.padding, conv_main.dilation)
        s = x if conv_short is None else conv2d(x, conv_short.weight, conv_short.bias, conv_short.stride,
                                                conv_short.padding, conv_short.dilation)
        return y, s
    cfg_m = (_pair(conv_main.stride), _pair(conv_main.padding), _pair(conv_main.dilation))
    _BN_HANDOFF[0] = _BN_HANDOFF[1] = None
    if conv_short is None:
        y, s = _ConvForkFn.apply(x, conv_main.weight, None, conv_main.bias, None, cfg_m, None, None,
                                 (bool(bn_stats[0]), False))
        parts, _BN_HANDOFF[0] = _BN_HANDOFF[0], None
        return _attach_parts(y, parts), s
    cfg_s = (_pair(conv_short.stride), _pair(conv_short.padding), _pair(conv_short.dilation))
    slot = getattr(x, '_evk_grad_slot', None)
    if slot is not None and not slot.claimed and torch.is_grad_enabled() and x.requires_grad:
        slot.claimed = True   # this node will add the parked gradient inside its shortcut data-gradient launch
    else:
        slot = None
    y, s = _ConvForkFn.apply(x, conv_main.weight, conv_short.weight, conv_main.bias, conv_short.bias, cfg_m, cfg_s, slot,
                             (bool(bn_stats[0]), bool(bn_stats[1])))
    pm, ps = _BN_HANDOFF
    _BN_HANDOFF[0] = _BN_HANDOFF[1] = None
    return _attach_parts(y, pm), _attach_parts(s, ps)


# ------------------------------------------------------------------------------------ transposed convolution
class _ConvTranspose2dFn(Function):
    """nn.ConvTranspose2d = the adjoint of the convolution C that reads the same weight memory as OHWI (include/ever_hip.h,
    evk_conv_transpose2d_*): forward on the residue-class data-gradient kernel, input gradient on the forward kernel,
    weight gradient on the weight-gradient kernel with the operand roles swapped.  No reference call site;
    parity is against torch.nn.ConvTranspose2d."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, output_padding, dilation):
        n, cin_t, h, w = x.shape
        if weight.shape[0] != cin_t:
            raise ValueError(f'conv_transpose2d: input has {cin_t} channels but weight expects {weight.shape[0]}')
        cout_t, kh, kw = weight.shape[1], weight.shape[2], weight.shape[3]
        if cin_t % 4 or cout_t % 4:
            raise HipPathError('conv_transpose2d: channel counts must be multiples of 4 on the HIP path')
        ho = (h - 1) * stride[0] - 2 * padding[0] + dilation[0] * (kh - 1) + output_padding[0] + 1
        wo = (w - 1) * stride[1] - 2 * padding[1] + dilation[1] * (kw - 1) + output_padding[1] + 1
        # descriptor of C: its input is this operator's output
        d = _C.ConvDesc(n, ho, wo, cout_t, h, w, cin_t, kh, kw, stride[0], stride[1], padding[0], padding[1],
                        dilation[0], dilation[1])
        chk_h = (ho + 2 * padding[0] - dilation[0] * (kh - 1) - 1) // stride[0] + 1
        chk_w = (wo + 2 * padding[1] - dilation[1] * (kw - 1) - 1) // stride[1] + 1
        if (chk_h, chk_w) != (h, w) or ho <= 0 or wo <= 0:
            raise ValueError('conv_transpose2d: output_padding must be smaller than stride or dilation')
        dev, st = x.device, _stream()
        w_ohwi = _weight_ohwi(weight.detach())      # [Cin_t][kh][kw][Cout_t]
        y = empty_nhwc(n, cout_t, ho, wo, dev)
        x3 = _planes_math() and cin_t % 8 == 0
        flops = 2.0 * n * h * w * cin_t * cout_t * kh * kw
        sp = timing.span('conv_igemm' if x3 else 'conv_igemm_f32', flops, 4.0 * (x.numel() + y.numel() + weight.numel()))
        if x3:
            pl_ptr = weight_planes.planes_for(weight, w_ohwi, d, 1, st)
            if pl_ptr is None:
                planes = workspace(dev, _C.load().evk_conv2d_split_weight_bytes(ctypes.byref(d), 1))
                _C.call('evk_conv2d_split_weight', ctypes.byref(d), w_ohwi.data_ptr(), 1, planes.data_ptr(), st)
                pl_ptr = planes.data_ptr()
            _C.call('evk_conv_transpose2d_fwd_x3', ctypes.byref(d), x.data_ptr(), pl_ptr, _ptr(bias), y.data_ptr(), st)
        else:
            wt = torch.empty((cout_t, kh * kw, cin_t), device=dev, dtype=torch.float32)
            _C.call('evk_conv2d_pack_dgrad_weight', ctypes.byref(d), w_ohwi.data_ptr(), wt.data_ptr(), st)
            _C.call('evk_conv_transpose2d_fwd', ctypes.byref(d), x.data_ptr(), wt.data_ptr(), _ptr(bias), y.data_ptr(), st)
        if sp is not None:
            sp.stop()
        ctx.desc, ctx.flops, ctx.has_bias, ctx.w_stride = d, flops, bias is not None, tuple(weight.stride())
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        d = ctx.desc
        dev, st = x.device, _stream()
        gy = as_nhwc(gy, 'conv_transpose2d.backward')
        w_ohwi = _weight_ohwi(weight.detach())
        cin_t, cout_t, kh, kw = weight.shape
        x3m = _planes_math()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            x3 = x3m and cout_t % 8 == 0
            sp = timing.span('conv_igemm' if x3 else 'conv_igemm_f32', ctx.flops, 4.0 * (x.numel() + gy.numel()))
            if x3:
                pl_ptr = weight_planes.planes_for(weight, w_ohwi, d, 0, st)
                if pl_ptr is None:
                    planes = workspace(dev, _C.load().evk_conv2d_split_weight_bytes(ctypes.byref(d), 0))
                    _C.call('evk_conv2d_split_weight', ctypes.byref(d), w_ohwi.data_ptr(), 0, planes.data_ptr(), st)
                    pl_ptr = planes.data_ptr()
                _C.call('evk_conv_transpose2d_dgrad_x3', ctypes.byref(d), gy.data_ptr(), pl_ptr, dx.data_ptr(), st)
            else:
                _C.call('evk_conv_transpose2d_dgrad', ctypes.byref(d), gy.data_ptr(), w_ohwi.data_ptr(), dx.data_ptr(), st)
            if sp is not None:
                sp.stop()
        need_dw, need_db = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if need_dw or need_db:
            ws_bytes = _C.load().evk_conv_transpose2d_wgrad_workspace_bytes(ctypes.byref(d), 1 if x3m else 0)
            ws = workspace(dev, ws_bytes)
            dwk = torch.empty((cin_t, kh * kw, cout_t), device=dev, dtype=torch.float32) if need_dw else None
            db = torch.empty((cout_t,), device=dev, dtype=torch.float32) if need_db else None
            sp = timing.span('conv_wgrad' if x3m else 'conv_wgrad_f32', ctx.flops, 4.0 * (x.numel() + gy.numel()))
            _C.call('evk_conv_transpose2d_wgrad_x3' if x3m else 'evk_conv_transpose2d_wgrad', ctypes.byref(d), x.data_ptr(),
                    gy.data_ptr(), _ptr(dwk), _ptr(db), ws.data_ptr(), ws_bytes, st)
            if sp is not None:
                sp.stop()
            if need_dw:
                dw = dwk.reshape(cin_t, kh, kw, cout_t).permute(0, 3, 1, 2)   # logical [Cin_t, Cout_t, kh, kw]
                if kh * kw == 1 and dw.stride() != ctx.w_stride and ctx.w_stride[1] == 1:
                    dw = dw.as_strided(dw.shape, ctx.w_stride)
        return dx, dw, db, None, None, None, None


def conv_transpose2d(x, weight, bias=None, stride=1, padding=0, output_padding=0, dilation=1):
    """F.conv_transpose2d (groups = 1).  weight: [Cin, Cout, kh, kw] (channels_last memory preferred)."""
    _require_cuda(x, 'conv_transpose2d')
    x = as_nhwc(x, 'conv_transpose2d')
    return _ConvTranspose2dFn.apply(x, weight, bias, _pair(stride), _pair(padding), _pair(output_padding), _pair(dilation))


# ------------------------------------------------------------------------------------ ResNet stem (space-to-depth)
class _StemConvFn(Function):
    """conv 7x7 / stride 2 / padding 3 on a 3- or 4-band image (reference _resnets.py:149) as a space-to-depth 4x4
    convolution with 16 input channels on the split-MFMA kernels (csrc/stem_s2d.hip): the image is re-laid once, the
    weight on every call (12 KB), the products and their sum are the ones of the 7x7 form.  No gradient to the image."""

    @staticmethod
    def forward(ctx, x, weight, want_stats=False):
        n, c, h, w = x.shape
        cout = weight.shape[0]
        dev, st = x.device, _stream()
        nchw = not is_nhwc(x)
        if nchw and not x.is_contiguous():
            x = x.contiguous()
        xs = torch.empty((n, h // 2 + 3, w // 2 + 3, 16), device=dev, dtype=torch.float32)
        _C.call('evk_stem_s2d', x.data_ptr(), xs.data_ptr(), n, c, h, w, 1 if nchw else 0, st)
        w7 = _weight_ohwi(weight.detach())                       # [Cout][7][7][C]
        w4 = torch.empty((cout, 4, 4, 16), device=dev, dtype=torch.float32)
        _C.call('evk_stem_s2d_weight', w7.data_ptr(), w4.data_ptr(), cout, c, st)
        d = _C.ConvDesc(n, h // 2 + 3, w // 2 + 3, 16, h // 2, w // 2, cout, 4, 4, 1, 1, 0, 0, 1, 1)
        pl_ptr, wabs_ptr, _keep = _weight_planes(weight, w4, w4.data_ptr(), d, 0, st, dev)   # (w4 is transient: split here)
        xbits = absmax_bits(xs, st) if wabs_ptr is not None else None
        y = empty_nhwc(n, cout, h // 2, w // 2, dev)
        flops = 2.0 * n * (h // 2) * (w // 2) * cout * c * 49     # algorithmic: the 7x7 taps, not the 4x4x16 padding
        nbytes = 4.0 * (x.numel() + y.numel() + weight.numel())
        sp = timing.span('conv_igemm', flops, nbytes)
        if wabs_ptr is not None:
            # BatchNorm statistics of the stem's output from the epilogue as well (the largest map of the network)
            parts, cap, nparts = None, 0, ctypes.c_int32(0)
            if want_stats and _BN_EPILOGUE and cout % 4 == 0:
                cap = int(_C.load().evk_conv2d_stats_max_parts(ctypes.byref(d)))
                parts = torch.empty((cap * 3 * cout,), device=dev, dtype=torch.float32)
            _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), xs.data_ptr(), xbits.data_ptr(), pl_ptr, wabs_ptr, None, None,
                    y.data_ptr(), 0, _ptr(parts), cap, ctypes.byref(nparts), None, st)
            _BN_HANDOFF[0] = (parts, int(nparts.value)) if nparts.value > 0 else None
        elif _CONV_MATH == 'bf16':
            _C.call('evk_conv2d_fwd_bf16', ctypes.byref(d), xs.data_ptr(), pl_ptr, None, y.data_ptr(), 0, None, 0,
                    ctypes.byref(ctypes.c_int32(0)), st)
        else:
            _C.call('evk_conv2d_fwd_x3', ctypes.byref(d), xs.data_ptr(), pl_ptr, None, y.data_ptr(), 0, st)
        if sp is not None:
            sp.stop()
        ctx.desc, ctx.flops, ctx.nbytes, ctx.cin, ctx.scope = d, flops, nbytes, c, timing.current_scope()
        ctx.w_stride = tuple(weight.stride())
        ctx.save_for_backward(xs)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (xs,) = ctx.saved_tensors
        d, c = ctx.desc, ctx.cin
        dev, st = xs.device, _stream()
        dy = as_nhwc(dy, 'stem.backward')
        lib = _C.load()
        ws_bytes = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d))
        ws = workspace(dev, ws_bytes)
        dw4 = torch.empty((d.Cout, 4, 4, 16), device=dev, dtype=torch.float32)
        h2 = _f16x2()
        if h2:
            xbits, dybits = absmax_bits(xs, st), absmax_bits(dy, st)
        sp = timing.span('conv_wgrad', ctx.flops, ctx.nbytes, ctx.scope)
        if h2:
            _C.call('evk_conv2d_wgrad_f16x2', ctypes.byref(d), xs.data_ptr(), xbits.data_ptr(), dy.data_ptr(), dybits.data_ptr(),
                    dw4.data_ptr(), None, ws.data_ptr(), ws_bytes, st)
        else:
            _C.call(_entry('evk_conv2d_wgrad_x3'), ctypes.byref(d), xs.data_ptr(), dy.data_ptr(), dw4.data_ptr(), None,
                    ws.data_ptr(), ws_bytes, st)
        if sp is not None:
            sp.stop()
        dw7 = torch.empty((d.Cout, 7, 7, c), device=dev, dtype=torch.float32)
        _C.call('evk_stem_s2d_weight_bwd', dw4.data_ptr(), dw7.data_ptr(), d.Cout, c, st)
        return None, dw7.permute(0, 3, 1, 2), None               # logical OIHW over OHWI memory, as the parameter


def stem_conv_applicable(x, conv):
    """True when `conv` is the 7x7 / stride-2 / padding-3 stem on a <= 4-band image that needs no gradient, under the
    split arithmetic: the cases the space-to-depth form covers."""
    return (_planes_math() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not x.requires_grad
            and tuple(conv.kernel_size) == (7, 7) and tuple(conv.stride) == (2, 2) and tuple(conv.padding) == (3, 3)
            and tuple(conv.dilation) == (1, 1) and conv.bias is None and conv.groups == 1 and x.shape[1] <= 4
            and conv.out_channels % 8 == 0 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
            and os.environ.get('EVK_STEM_S2D', '1') != '0')


def stem_conv7x7s2(x, weight, bn_stats=False):
    _require_cuda(x, 'stem_conv7x7s2')
    _BN_HANDOFF[0] = None
    y = _StemConvFn.apply(x, weight, bool(bn_stats))
    parts, _BN_HANDOFF[0] = _BN_HANDOFF[0], None
    return _attach_parts(y, parts)


# ------------------------------------------------------------------------------------ batch norm
class _BatchNormActFn(Function):
    """BatchNorm2d (+ residual add) (+ ReLU) in one pass.

    Replaces aten::batch_norm, `out += identity`, relu_ at reference ever/module/_resnets.py:95-112,
    fs_relation.py:39-53, fpn.py:163-167.
    """

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, training, momentum, eps, relu, parts=None,
                pack_out=False, lazy_res=False):
        n, c, h, w = x.shape
        rows = n * h * w
        dev = x.device
        st = _stream()
        lib = _C.load()
        ws_bytes = lib.evk_bn_workspace_bytes(rows, c)
        ws = workspace(dev, ws_bytes)
        y = empty_nhwc(n, c, h, w, dev)
        save_mean = torch.empty((c,), device=dev, dtype=torch.float32)
        save_invstd = torch.empty((c,), device=dev, dtype=torch.float32)
        flags = 1 if relu else 0
        # pack_out: y is one convolution's operand and nothing else — written packed, its scale bounded from the
        # statistics records before the apply pass (EVK_BN_PACK_Y); rows: that convolution must be on the plane kernels
        pack = bool(pack_out and _PACKED and _f16x2() and training and parts is not None and residual is None
                    and c % 8 == 0 and rows >= 256 and not observers_active())
        abits = _amax_zeroed(dev) if pack else _amax_out(dev)
        pack = pack and abits is not None
        if pack:
            flags |= 4
        # algorithmic bytes (fp32): statistics read + apply read/write (+ residual read)
        nb = 4.0 * x.numel() * ((3 if training else 2) + (1 if residual is not None else 0))
        rbits = None
        if training and parts is not None and residual is not None and relu and _RELU_BITS:
            # the end of a residual block: the ReLU bits go out beside y and the backward reads them instead of y
            rbits = torch.empty((lib.evk_relu_bits_bytes(x.numel()) // 4,), device=dev, dtype=torch.int32)
            relu_bits_stats['forward'] += 1
            _timed_call('bn', nb - 4.0 * x.numel(), 'evk_bn_fwd_train_parts_bits', x.data_ptr(), _ptr(residual), _ptr(weight),
                        _ptr(bias), _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), y.data_ptr(),
                        save_mean.data_ptr(), save_invstd.data_ptr(), rows, c, flags, parts[0].data_ptr(), parts[1],
                        ws.data_ptr(), ws_bytes, _ptr(abits), rbits.data_ptr(), st)
        elif training and parts is not None:
            # statistics came with x from the convolution's epilogue: merge the records, apply (2|x| of traffic)
            _timed_call('bn', nb - 4.0 * x.numel(), 'evk_bn_fwd_train_parts', x.data_ptr(), _ptr(residual), _ptr(weight),
                        _ptr(bias), _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), y.data_ptr(),
                        save_mean.data_ptr(), save_invstd.data_ptr(), rows, c, flags, parts[0].data_ptr(), parts[1],
                        ws.data_ptr(), ws_bytes, _ptr(abits), st)
        elif training:
            _timed_call('bn', nb, 'evk_bn_fwd_train', x.data_ptr(), _ptr(residual), _ptr(weight), _ptr(bias), _ptr(running_mean),
                    _ptr(running_var), float(momentum), float(eps), y.data_ptr(), save_mean.data_ptr(),
                    save_invstd.data_ptr(), rows, c, flags, ws.data_ptr(), ws_bytes, _ptr(abits), st)
        else:
            _timed_call('bn', nb, 'evk_bn_fwd_eval', x.data_ptr(), _ptr(residual), _ptr(weight), _ptr(bias), running_mean.data_ptr(),
                    running_var.data_ptr(), float(eps), y.data_ptr(), save_mean.data_ptr(), save_invstd.data_ptr(),
                    rows, c, flags, ws.data_ptr(), ws_bytes, _ptr(abits), st)
        global _AMAX_HANDOFF
        _AMAX_HANDOFF = (abits, pack)
        ctx.training = training
        ctx.pack_dx = bool(parts is not None and len(parts) > 2 and parts[2])
        ctx.relu = relu
        ctx.has_res = residual is not None
        # the ReLU mask is recomputed from x in backward unless a residual was added (then y — or its bits — is needed)
        ctx.lazy_res = bool(lazy_res and rbits is not None and _LAZY_RES and not observers_active())
        ctx.save_for_backward(x, y if (relu and residual is not None and rbits is None) else None, weight, bias, save_mean,
                              save_invstd, rbits)
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var) if t is not None])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y, weight, bias, save_mean, save_invstd, rbits = ctx.saved_tensors
        n, c, h, w = x.shape
        rows = n * h * w
        dev = x.device
        st = _stream()
        in_bits = _lazy_bits(dy)
        if in_bits is not None and (ctx.relu or rbits is not None):
            dy, in_bits = materialize_lazy(dy), None     # an own mask AND an incoming one: apply the incoming one first
        dy = as_nhwc(dy, 'batch_norm.backward')
        lib = _C.load()
        ws_bytes = lib.evk_bn_workspace_bytes(rows, c)
        ws = workspace(dev, ws_bytes)
        dx = torch.empty_like(x)
        need_res = ctx.has_res and ctx.needs_input_grad[1]
        lazy = need_res and ctx.lazy_res and rbits is not None and in_bits is None
        dres = torch.empty_like(x) if (need_res and not lazy) else None
        has_affine = weight is not None
        dgamma = torch.empty((c,), device=dev, dtype=torch.float32) if has_affine else None
        dbeta = torch.empty((c,), device=dev, dtype=torch.float32) if has_affine else None
        # reduce pass reads dy, x (+y mask); apply pass reads g, x and writes dx (+ the residual gradient write)
        nb = 4.0 * x.numel() * (5 + (1 if y is not None else 0) + (1 if dres is not None else 0))
        # dx is the producing convolution's dy (data and weight gradient operand) — and, when that convolution said so
        # in forward, nothing else: written packed under a scale bounded before the apply pass (EVK_BN_PACK_DX)
        pack = ctx.pack_dx and _f16x2()
        abits = _amax_zeroed(dev) if pack else _amax_out(dev)
        pack = pack and abits is not None
        mask_bits = rbits if rbits is not None else in_bits
        if in_bits is not None:
            relu_bits_stats['masked_bn'] += 1
        _timed_call('bn', nb, 'evk_bn_bwd_bits', dy.data_ptr(), x.data_ptr(), _ptr(y), _ptr(weight), _ptr(bias),
                    save_mean.data_ptr(), save_invstd.data_ptr(), dx.data_ptr(), _ptr(dres), _ptr(dgamma), _ptr(dbeta), rows, c,
                    (1 if ctx.relu else 0) | (2 if pack else 0),
                    1 if ctx.training else 0, ws.data_ptr(), ws_bytes, _ptr(abits), _ptr(mask_bits), st)
        if lazy:
            # the identity branch's gradient = dy where the block's output was positive: handed on unmasked with the bits
            relu_bits_stats['lazy'] += 1
            dres = dy.view_as(dy)
            dres._evk_relu_bits = (dres._version, dres.data_ptr(), rbits)
            _inherit_amax(dres, dy)
        if pack:
            _mark_packed(dx, abits)
        elif abits is not None:
            _note_amax(dx, abits)
        if ctx.has_res and not need_res:
            dres = None
        return (dx, dres, dgamma if ctx.needs_input_grad[2] else None, dbeta if ctx.needs_input_grad[3] else None,
                None, None, None, None, None, None, None, None, None)


def batch_norm_act(x, weight, bias, running_mean, running_var, training, momentum, eps, residual=None, relu=False,
                   pack_out=False, lazy_res=False):
    """lazy_res: the caller guarantees that the gradient of `residual` reaches only readers of this package that take an
    unmasked gradient with ReLU bits (a `conv2d_fork` identity output, or the BatchNorm of a shortcut convolution): the
    backward then hands the incoming gradient on as it is instead of writing a masked copy (EVK_LAZY_RES).
    pack_out: the caller guarantees that ONE convolution of this package (forward + weight gradient) is the only
    reader of the result; under the f16x2 arithmetic it is then stored packed (include/ever_hip.h: EVK_BN_PACK_Y)."""
    _require_cuda(x, 'batch_norm')
    x = as_nhwc(x, 'batch_norm')
    if x.shape[1] % 4 != 0:
        raise HipPathError(f'batch_norm: channel count {x.shape[1]} must be a multiple of 4')
    if residual is not None:
        residual = as_nhwc(residual, 'batch_norm.residual')
    use_batch_stats = training or running_mean is None
    parts = getattr(x, '_evk_bn_parts', None) if use_batch_stats else None
    if parts is not None:
        del x._evk_bn_parts
    global _AMAX_HANDOFF
    _AMAX_HANDOFF = None
    if use_batch_stats and running_mean is not None:
        weight_planes.note_running_stats_changed()
    y = _BatchNormActFn.apply(x, residual, weight, bias, running_mean, running_var, bool(use_batch_stats),
                              0.0 if momentum is None else momentum, eps, bool(relu), parts, bool(pack_out), bool(lazy_res))
    if _AMAX_HANDOFF is not None:       # the pass left max|y| (or its bound) there: y is the next convolution's operand
        abits, packed = _AMAX_HANDOFF
        if packed:
            _mark_packed(y, abits)
        elif abits is not None:
            _note_amax(y, abits)
        _AMAX_HANDOFF = None
    return y


_AMAX_HANDOFF = None


class _BnReluPoolFn(Function):
    """The stem's BatchNorm (batch statistics from the convolution epilogue's records) + ReLU + MaxPool2d(3, 2, 1) as
    one pass each way (csrc/bn.hip: bn_relu_pool_fwd_kernel): the normalised full-resolution map and its gradient are
    never written.  Replaces bn1 / relu / maxpool of reference ever/module/_resnets.py:150-153."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, parts):
        n, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        dev, st = x.device, _stream()
        ws_bytes = _C.load().evk_bn_workspace_bytes(n * h * w, c)
        ws = workspace(dev, ws_bytes)
        y = empty_nhwc(n, c, ho, wo, dev)
        code = torch.empty((n, ho, wo, c), device=dev, dtype=torch.uint8)
        save_mean = torch.empty((c,), device=dev, dtype=torch.float32)
        save_invstd = torch.empty((c,), device=dev, dtype=torch.float32)
        abits = _amax_out(dev)
        # algorithmic bytes: read x, write the pooled map and the codes
        nb = 4.0 * x.numel() + 5.0 * y.numel()
        _timed_call('bn', nb, 'evk_bn_relu_pool_fwd_train_parts', x.data_ptr(), _ptr(weight), _ptr(bias), _ptr(running_mean),
                    _ptr(running_var), float(momentum), float(eps), y.data_ptr(), code.data_ptr(), save_mean.data_ptr(),
                    save_invstd.data_ptr(), n, h, w, c, parts[0].data_ptr(), parts[1], ws.data_ptr(), ws_bytes, _ptr(abits), st)
        global _AMAX_HANDOFF
        _AMAX_HANDOFF = (abits, False)
        ctx.save_for_backward(x, weight, bias, save_mean, save_invstd, code)
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var) if t is not None])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dp):
        x, weight, bias, save_mean, save_invstd, code = ctx.saved_tensors
        n, c, h, w = x.shape
        dev, st = x.device, _stream()
        dp = as_nhwc(dp, 'bn_relu_pool.backward')
        ws_bytes = _C.load().evk_bn_workspace_bytes(n * h * w, c)
        ws = workspace(dev, ws_bytes)
        dx = torch.empty_like(x)
        has_affine = weight is not None
        dgamma = torch.empty((c,), device=dev, dtype=torch.float32) if has_affine else None
        dbeta = torch.empty((c,), device=dev, dtype=torch.float32) if has_affine else None
        abits = _amax_out(dev)
        # both passes read x, the pooled gradient and the codes; the apply pass writes dx
        nb = 4.0 * x.numel() * 3 + 2 * 5.0 * dp.numel()
        _timed_call('bn', nb, 'evk_bn_relu_pool_bwd', dp.data_ptr(), code.data_ptr(), x.data_ptr(), _ptr(weight), _ptr(bias),
                    save_mean.data_ptr(), save_invstd.data_ptr(), dx.data_ptr(), _ptr(dgamma), _ptr(dbeta), n, h, w, c, 1,
                    ws.data_ptr(), ws_bytes, _ptr(abits), st)
        if abits is not None:
            _note_amax(dx, abits)
        return (dx, dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None,
                None, None, None, None, None)


_STEM_POOL = os.environ.get('EVK_STEM_POOL', '1') != '0'


def batch_norm_relu_max_pool(x, weight, bias, running_mean, running_var, momentum, eps):
    """max_pool3x3s2(relu(batch_norm(x))) in training mode.  One fused pass each way when x carries the statistics
    records of the convolution that produced it (conv2d / stem_conv7x7s2 with bn_stats=True); the two separate passes
    otherwise (EVK_STEM_POOL=0 forces them)."""
    _require_cuda(x, 'batch_norm_relu_max_pool')
    x = as_nhwc(x, 'batch_norm_relu_max_pool')
    parts = getattr(x, '_evk_bn_parts', None)
    n, c, h, w = x.shape
    if parts is None or not _STEM_POOL or c % 4 or n * h * w * (c // 4) >= 2 ** 31:
        return max_pool3x3s2(batch_norm_act(x, weight, bias, running_mean, running_var, True, momentum, eps, relu=True))
    del x._evk_bn_parts
    global _AMAX_HANDOFF
    _AMAX_HANDOFF = None
    if running_mean is not None:
        weight_planes.note_running_stats_changed()
    y = _BnReluPoolFn.apply(x, weight, bias, running_mean, running_var, 0.0 if momentum is None else momentum, eps, parts)
    if _AMAX_HANDOFF is not None:
        if _AMAX_HANDOFF[0] is not None:
            _note_amax(y, _AMAX_HANDOFF[0])
        _AMAX_HANDOFF = None
    return y


# ------------------------------------------------------------------------------------ pointwise
class _ReluFn(Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        _C.call('evk_relu_fwd', x.data_ptr(), y.data_ptr(), x.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = as_nhwc(dy, 'relu.backward') if dy.dim() == 4 else dy.contiguous()
        dx = torch.empty_like(y)
        _C.call('evk_relu_bwd', dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), _stream())
        return dx


def relu(x):
    """nn.ReLU (reference fs_relation.py:25)."""
    _require_cuda(x, 'relu')
    if x.dim() == 4:
        x = as_nhwc(x, 'relu')
    return _ReluFn.apply(x)


class _AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        _C.call('evk_add', a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream())
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g, g


def add(a, b):
    _require_cuda(a, 'add')
    a, b = as_nhwc(a, 'add'), as_nhwc(b, 'add')
    if a.shape != b.shape:
        raise ValueError(f'add: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}')
    return _AddFn.apply(a, b)


class _MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = empty_nhwc(n, c, ho, wo, x.device)
        code = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8)
        _C.call('evk_maxpool3x3s2_fwd', x.data_ptr(), y.data_ptr(), code.data_ptr(), n, h, w, c, _stream())
        ctx.save_for_backward(code)
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (code,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        dy = as_nhwc(dy, 'max_pool.backward')
        dx = empty_nhwc(n, c, h, w, dy.device)
        _C.call('evk_maxpool3x3s2_bwd', dy.data_ptr(), code.data_ptr(), dx.data_ptr(), n, h, w, c, _stream())
        return dx


def max_pool3x3s2(x):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (reference _resnets.py:153)."""
    _require_cuda(x, 'max_pool')
    x = as_nhwc(x, 'max_pool')
    if x.shape[1] % 4:
        raise HipPathError('max_pool: channels must be a multiple of 4')
    return _inherit_amax(_MaxPoolFn.apply(x), x)      # a selection of x's elements


class _Nearest2xAddFn(Function):
    @staticmethod
    def forward(ctx, top, lateral):
        n, c, h, w = lateral.shape
        out = empty_nhwc(n, c, h, w, lateral.device)
        bits = _amax_zeroed(lateral.device)      # the sum is the FPN output convolution's operand
        _C.call('evk_upsample_nearest2x_add_fwd', top.data_ptr(), lateral.data_ptr(), out.data_ptr(), n, h, w, c,
                _ptr(bits), _stream())
        if bits is not None:
            _note_amax(out, bits)
        ctx.shape = (n, c, h, w)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        g = as_nhwc(g, 'nearest2x.backward')
        dtop = None
        if ctx.needs_input_grad[0]:
            dtop = empty_nhwc(n, c, h // 2, w // 2, g.device)
            _C.call('evk_upsample_nearest2x_bwd', g.data_ptr(), dtop.data_ptr(), n, h, w, c, _stream())
        return dtop, (g if ctx.needs_input_grad[1] else None)


def upsample_nearest2x_add(top, lateral):
    """`inner_lateral + F.interpolate(last_inner, scale_factor=2, mode="nearest")` (reference fpn.py:100-105)."""
    _require_cuda(top, 'upsample_nearest2x_add')
    top, lateral = as_nhwc(top, 'fpn.top'), as_nhwc(lateral, 'fpn.lateral')
    n, c, h, w = lateral.shape
    if top.shape != (n, c, h // 2, w // 2) or h % 2 or w % 2:
        # same failure mode as the reference's `inner_lateral + inner_top_down` (fpn.py:105)
        raise RuntimeError(f'The size of tensor a ({tuple(lateral.shape)}) must match the size of tensor b '
                           f'(nearest x2 of {tuple(top.shape)})')
    return _Nearest2xAddFn.apply(top, lateral)


class _Subsample2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, x.device)
        _C.call('evk_subsample2_fwd', x.data_ptr(), y.data_ptr(), n, h, w, c, _stream())
        _inherit_amax(y, x)                       # a selection of x's elements: its scale bounds them
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        n, c, h, w = ctx.shape
        g = as_nhwc(g, 'subsample2.backward')
        dx = empty_nhwc(n, c, h, w, g.device)
        _C.call('evk_subsample2_bwd', g.data_ptr(), dx.data_ptr(), n, h, w, c, _stream())
        return dx


def max_pool1x1s2(x):
    """F.max_pool2d(x, 1, 2, 0) (reference fpn.py:118-120, LastLevelMaxPool): every second pixel of every second row."""
    _require_cuda(x, 'max_pool1x1s2')
    x = as_nhwc(x, 'max_pool1x1s2')
    if x.shape[1] % 4:
        raise HipPathError('max_pool1x1s2: channels must be a multiple of 4')
    return _Subsample2Fn.apply(x)


class _BilinearFn(Function):
    @staticmethod
    def forward(ctx, x, ho, wo):
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, ho, wo, x.device)
        _timed_call('resample_loss', 4.0 * (x.numel() + y.numel()), 'evk_upsample_bilinear_fwd', x.data_ptr(),
                    y.data_ptr(), n, h, w, ho, wo, c, _stream())
        ctx.dims = (n, c, h, w, ho, wo)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, c, h, w, ho, wo = ctx.dims
        dy = as_nhwc(dy, 'bilinear.backward')
        dx = empty_nhwc(n, c, h, w, dy.device)
        _timed_call('resample_loss', 4.0 * (dy.numel() + dx.numel()), 'evk_upsample_bilinear_bwd', dy.data_ptr(),
                    dx.data_ptr(), n, h, w, ho, wo, c, _stream())
        return dx, None, None


def upsample_bilinear(x, scale_factor):
    """nn.UpsamplingBilinear2d(scale_factor) == bilinear with align_corners=True (reference fpn.py:168,180)."""
    _require_cuda(x, 'upsample_bilinear')
    x = as_nhwc(x, 'upsample_bilinear')
    sh, sw = (scale_factor, scale_factor) if not isinstance(scale_factor, (tuple, list)) else scale_factor
    ho, wo = int(x.shape[2] * sh), int(x.shape[3] * sw)  # floor(in * scale), as aten
    return _inherit_amax(_BilinearFn.apply(x, ho, wo), x)   # convex combinations of x's elements


class _GapFn(Function):
    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, 1, 1, x.device)
        _C.call('evk_gap_fwd', x.data_ptr(), y.data_ptr(), n, h * w, c, _stream())
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        dy = dy.reshape(n, c).contiguous()
        dx = empty_nhwc(n, c, h, w, dy.device)
        _C.call('evk_gap_bwd', dy.data_ptr(), dx.data_ptr(), n, h * w, c, _stream())
        return dx


def global_avg_pool(x):
    """F.adaptive_avg_pool2d(x, 1) (reference fs_relation.py:177)."""
    _require_cuda(x, 'global_avg_pool')
    x = as_nhwc(x, 'global_avg_pool')
    return _GapFn.apply(x)


class _RelationFn(Function):
    @staticmethod
    def forward(ctx, scene, content, feat):
        n, c, h, w = content.shape
        out = empty_nhwc(n, c, h, w, content.device)
        r = torch.empty((n, h * w), device=content.device, dtype=torch.float32)
        _C.call('evk_relation_fwd', scene.data_ptr(), content.data_ptr(), feat.data_ptr(), out.data_ptr(),
                r.data_ptr(), n, h * w, c, _stream())
        ctx.save_for_backward(scene, content, feat, r)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        scene, content, feat, r = ctx.saved_tensors
        n, c, h, w = content.shape
        dev = content.device
        dout = as_nhwc(dout, 'fs_relation.backward')
        lib = _C.load()
        ws_bytes = lib.evk_relation_workspace_bytes(n, h * w, c)
        ws = workspace(dev, ws_bytes)
        dscene = empty_nhwc(n, c, 1, 1, dev)
        dcontent = torch.empty_like(content)
        dfeat = torch.empty_like(feat)
        _C.call('evk_relation_bwd', dout.data_ptr(), scene.data_ptr(), content.data_ptr(), feat.data_ptr(),
                r.data_ptr(), dscene.data_ptr(), dcontent.data_ptr(), dfeat.data_ptr(), n, h * w, c, ws.data_ptr(),
                ws_bytes, _stream())
        return dscene, dcontent, dfeat


def fs_relation(scene, content, feat):
    """`sigmoid((scene * content).sum(dim=1, keepdim=True)) * feat` (reference fs_relation.py:61-71)."""
    _require_cuda(content, 'fs_relation')
    content, feat = as_nhwc(content, 'fs_relation.content'), as_nhwc(feat, 'fs_relation.feat')
    n, c, h, w = content.shape
    scene = as_nhwc(scene.reshape(n, c, 1, 1), 'fs_relation.scene')
    return _inherit_amax(_RelationFn.apply(scene, content, feat), feat)   # sigmoid(.) * feat


class _RelationBnFn(Function):
    """FS-Relation on the two convolution outputs directly: BatchNorm (batch statistics from the convolutions' epilogue
    records) + ReLU of both branches happen inside the relation kernels (include/ever_hip.h: evk_relation_bn_*), their
    backward sums come out of the relation backward.  Replaces content_encoder[1:], feature_reencoder[1:] and the relation
    of reference fs_relation.py:39-53,61-71 as one node."""

    @staticmethod
    def forward(ctx, scene, zc, zf, wc, bc, wf, bf, rmc, rvc, rmf, rvf, cfg):
        (parts_c, parts_f, mom_c, eps_c, mom_f, eps_f) = cfg
        n, c, h, w = zc.shape
        rows, dev, st = n * h * w, zc.device, _stream()
        stats = torch.empty((2, 4, c), device=dev, dtype=torch.float32)   # per BatchNorm: mean, invstd, scale, shift
        for k, (parts, g, b, rm, rv, mom, eps) in enumerate(((parts_c, wc, bc, rmc, rvc, mom_c, eps_c),
                                                               (parts_f, wf, bf, rmf, rvf, mom_f, eps_f))):
            _C.call('evk_bn_finalize_parts', parts[0].data_ptr(), parts[1], c, rows, _ptr(g), _ptr(b), _ptr(rm), _ptr(rv),
                    float(mom), float(eps), stats[k, 0].data_ptr(), stats[k, 1].data_ptr(), stats[k, 2].data_ptr(), st)
        out = empty_nhwc(n, c, h, w, dev)
        r = torch.empty((n, h * w), device=dev, dtype=torch.float32)
        abits = _amax_zeroed(dev)
        _C.call('evk_relation_bn_fwd', scene.data_ptr(), zc.data_ptr(), stats[0, 2].data_ptr(), zf.data_ptr(),
                stats[1, 2].data_ptr(), out.data_ptr(), r.data_ptr(), n, h * w, c, _ptr(abits), st)
        global _AMAX_HANDOFF
        _AMAX_HANDOFF = (abits, False)
        ctx.pack = (bool(len(parts_c) > 2 and parts_c[2]), bool(len(parts_f) > 2 and parts_f[2]))
        ctx.save_for_backward(scene, zc, zf, wc, wf, stats, r)
        ctx.mark_non_differentiable(*[t for t in (rmc, rvc, rmf, rvf) if t is not None])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        scene, zc, zf, wc, wf, stats, r = ctx.saved_tensors
        n, c, h, w = zc.shape
        rows, dev, st = n * h * w, zc.device, _stream()
        dout = as_nhwc(dout, 'fs_relation.backward')
        lib = _C.load()
        nb = int(lib.evk_relation_bn_parts(n, h * w))
        ws_bytes = lib.evk_relation_bn_workspace_bytes(n, h * w, c)
        # (own buffer, not the shared workspace: the BatchNorm backward launches below read it while they use that one)
        ws = torch.empty((ws_bytes // 4,), device=dev, dtype=torch.float32)
        dscene = empty_nhwc(n, c, 1, 1, dev)
        gc, gf = torch.empty_like(zc), torch.empty_like(zf)
        _C.call('evk_relation_bn_bwd', dout.data_ptr(), scene.data_ptr(), zc.data_ptr(), stats[0, 2].data_ptr(),
                stats[0, 0].data_ptr(), zf.data_ptr(), stats[1, 2].data_ptr(), stats[1, 0].data_ptr(), r.data_ptr(),
                dscene.data_ptr(), gc.data_ptr(), gf.data_ptr(), n, h * w, c, ws.data_ptr(), ws_bytes, st)
        coef = workspace(dev, 16 * c * 4)
        grads = []
        for k, (g, z, gamma) in enumerate(((gc, zc, wc), (gf, zf, wf))):
            sums = ws[nb * c * (1 + 4 * k):]
            maxima = ws[nb * c * (3 + 4 * k):]
            pack = ctx.pack[k] and _f16x2()
            abits = _amax_zeroed(dev) if pack else _amax_out(dev)
            pack = pack and abits is not None
            dz = torch.empty_like(z)
            dgamma = torch.empty((c,), device=dev, dtype=torch.float32) if gamma is not None else None
            dbeta = torch.empty((c,), device=dev, dtype=torch.float32) if gamma is not None else None
            # algorithmic bytes: read g, z, write dz
            _timed_call('bn', 12.0 * z.numel(), 'evk_bn_bwd_from_partials', g.data_ptr(), z.data_ptr(), _ptr(gamma),
                        stats[k, 0].data_ptr(), stats[k, 1].data_ptr(), sums.data_ptr(), maxima.data_ptr(), nb, dz.data_ptr(),
                        _ptr(dgamma), _ptr(dbeta), rows, c, 2 if pack else 0, 1, coef.data_ptr(), 16 * c * 4, _ptr(abits), st)
            if pack:
                _mark_packed(dz, abits)
            elif abits is not None:
                _note_amax(dz, abits)
            grads.append((dz, dgamma, dbeta))
        (dzc, dgc, dbc), (dzf, dgf, dbf) = grads
        return dscene, dzc, dzf, dgc, dbc, dgf, dbf, None, None, None, None, None


def fs_relation_bn(scene, zc, zf, bn_c, bn_f):
    """`sigmoid(<scene, relu(bn_c(zc))>) * relu(bn_f(zf))` with both training-mode BatchNorms inside the relation kernels
    (see _RelationBnFn); zc, zf carry their convolutions' statistics records (`_evk_bn_parts`), or None is returned and the
    caller runs the layers one by one."""
    pc, pf = getattr(zc, '_evk_bn_parts', None), getattr(zf, '_evk_bn_parts', None)
    if pc is None or pf is None or pc[1] <= 0 or pf[1] <= 0 or zc.shape != zf.shape or zc.shape[1] % 4 or zc.shape[1] > 448:
        return None
    del zc._evk_bn_parts, zf._evk_bn_parts
    n, c, h, w = zc.shape
    scene = as_nhwc(scene.reshape(n, c, 1, 1), 'fs_relation.scene')
    weight_planes.note_running_stats_changed()
    global _AMAX_HANDOFF
    _AMAX_HANDOFF = None

    def stat(bn, name):
        return getattr(bn, name) if bn.track_running_stats else None
    out = _RelationBnFn.apply(scene, zc, zf, bn_c.weight, bn_c.bias, bn_f.weight, bn_f.bias, stat(bn_c, 'running_mean'),
                              stat(bn_c, 'running_var'), stat(bn_f, 'running_mean'), stat(bn_f, 'running_var'),
                              (pc, pf, bn_c.momentum, bn_c.eps, bn_f.momentum, bn_f.eps))
    if _AMAX_HANDOFF is not None:
        abits, _ = _AMAX_HANDOFF
        if abits is not None:
            _note_amax(out, abits)
        _AMAX_HANDOFF = None
    return out


class _BnReluDotFn(Function):
    """out = conv1x1(relu(bn(z))) for a narrow classifier (K <= 16), BatchNorm with batch statistics from the producing
    convolution's epilogue records, as one consumer of z (include/ever_hip.h: evk_bn_relu_dot_*): the normalised map is
    never written, its K-fold outer-product gradient never formed.  Replaces blocks[i][-1][1:3] + classifier[0] of reference
    fpn.py:163-170,179-193 in the commuted decoder (module/fpn.py)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, w, bias, rm, rv, cfg):
        parts, mom, eps = cfg
        n, c, h, wd = z.shape
        k = w.shape[0]
        rows, dev, st = n * h * wd, z.device, _stream()
        stats = torch.empty((4, c), device=dev, dtype=torch.float32)      # mean, invstd, scale, shift
        _C.call('evk_bn_finalize_parts', parts[0].data_ptr(), parts[1], c, rows, _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv),
                float(mom), float(eps), stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), st)
        w2 = _weight_ohwi(w.detach()).reshape(k, c)
        out = empty_nhwc(n, k, h, wd, dev)
        # algorithmic bytes: read z (the K-channel result is noise beside it)
        _timed_call('bn', 4.0 * z.numel(), 'evk_bn_relu_dot_fwd', z.data_ptr(), stats[2].data_ptr(), w2.data_ptr(), _ptr(bias),
                    out.data_ptr(), rows, c, k, st)
        ctx.pack = bool(len(parts) > 2 and parts[2])
        ctx.has_bias = bias is not None
        ctx.save_for_backward(z, gamma, w, stats)
        ctx.mark_non_differentiable(*[t for t in (rm, rv) if t is not None])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dl):
        z, gamma, w, stats = ctx.saved_tensors
        n, c, h, wd = z.shape
        k = w.shape[0]
        rows, dev, st = n * h * wd, z.device, _stream()
        dl = as_nhwc(dl, 'bn_relu_dot.backward')
        lib = _C.load()
        ws_bytes = lib.evk_bn_relu_dot_workspace_bytes(rows, c, k)
        ws = workspace(dev, ws_bytes)
        pack = ctx.pack and _f16x2()
        abits = _amax_zeroed(dev) if pack else _amax_out(dev)
        pack = pack and abits is not None
        dz = torch.empty_like(z)
        dgamma = torch.empty((c,), device=dev, dtype=torch.float32) if gamma is not None else None
        dbeta = torch.empty((c,), device=dev, dtype=torch.float32) if gamma is not None else None
        # (OHWI memory = [k][c] for a 1x1 kernel, presented with exactly the parameter's strides: the size-1 dims make the
        # stride tuple ambiguous, and a gradient in "another layout" costs AccumulateGrad / the bucket pack a copy)
        dw = torch.empty_strided(w.shape, w.stride(), device=dev, dtype=torch.float32) if (
            w.stride(0) == c and w.stride(1) == 1) else torch.empty_like(w, memory_format=torch.channels_last)
        dbias = torch.empty((k,), device=dev, dtype=torch.float32) if ctx.has_bias else None
        w2 = _weight_ohwi(w.detach()).reshape(k, c)
        # algorithmic bytes: read z twice, write dz
        _timed_call('bn', 12.0 * z.numel(), 'evk_bn_relu_dot_bwd', dl.data_ptr(), z.data_ptr(), stats[2].data_ptr(), _ptr(gamma),
                    stats[0].data_ptr(), stats[1].data_ptr(), w2.data_ptr(), dz.data_ptr(), _ptr(dgamma), _ptr(dbeta),
                    dw.data_ptr(), _ptr(dbias), rows, c, k, 2 if pack else 0, ws.data_ptr(), ws_bytes, _ptr(abits), st)
        if pack:
            _mark_packed(dz, abits)
        elif abits is not None:
            _note_amax(dz, abits)
        return dz, dgamma, dbeta, dw, dbias, None, None, None


def bn_relu_dot(z, bn, conv):
    """`conv(relu(bn(z)))` for a training-mode BatchNorm2d whose statistics records ride on z (`_evk_bn_parts`) and a 1x1
    convolution with at most 16 outputs, as one pass each way (see _BnReluDotFn); None when that form does not apply."""
    parts = getattr(z, '_evk_bn_parts', None)
    k, c = conv.weight.shape[0], z.shape[1]
    if (parts is None or parts[1] <= 0 or k > 16 or c % 4 or c > 1024 or tuple(conv.weight.shape[2:]) != (1, 1)
            or conv.weight.shape[1] != c or (c > 256 and k > 4) or 16 * (4 + k) * c > 65536):
        return None      # (the kernel's register / LDS budget: csrc/bn.hip evk_bn_relu_dot_bwd)
    del z._evk_bn_parts
    weight_planes.note_running_stats_changed()
    if torch.is_grad_enabled():
        _note_param_use(conv.weight, conv.bias)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BnReluDotFn.apply(z, bn.weight, bn.bias, conv.weight, conv.bias, rm, rv, (parts, bn.momentum, bn.eps))


class _Mean4Fn(Function):
    @staticmethod
    def forward(ctx, a, b, c, d):
        out = torch.empty_like(a)
        bits = _amax_zeroed(a.device)            # the mean is the classifier convolution's operand
        _C.call('evk_mean4_fwd', a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), out.data_ptr(), a.numel(),
                _ptr(bits), _stream())
        if bits is not None:
            _note_amax(out, bits)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = as_nhwc(g, 'mean4.backward')
        q = torch.empty_like(g)
        _C.call('evk_scale', g.data_ptr(), 0.25, q.data_ptr(), g.numel(), _stream())
        return q, q, q, q


def mean4(a, b, c, d):
    """`sum(inner_feat_list) / len(inner_feat_list)` for the 4 decoder branches (reference fpn.py:189)."""
    ts = [as_nhwc(t, 'mean4') for t in (a, b, c, d)]
    for t in ts[1:]:
        if t.shape != ts[0].shape:
            raise ValueError('mean4: shape mismatch')
    return _Mean4Fn.apply(*ts)


# ------------------------------------------------------------------------------------ losses
def _stats_buf(k, device):
    n = _C.load().evk_loss_stats_doubles(k)
    return torch.empty((n,), device=device, dtype=torch.float64)


def _labels(y_true, npix, what):
    if y_true.dtype != torch.int64:
        y_true = y_true.long()
    y_true = y_true.contiguous()
    if y_true.numel() != npix:
        raise ValueError(f'{what}: labels have {y_true.numel()} elements, logits have {npix} pixels')
    return y_true


class _BceFn(Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index, eps, pos_weight, reduction):
        npix = logits.numel()
        stats = _stats_buf(2, logits.device)
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        _timed_call('resample_loss', 12.0 * npix, 'evk_bce_fwd_ex', logits.data_ptr(), labels.data_ptr(), npix, ignore_index, eps,
                    pos_weight, reduction, loss.data_ptr(), stats.data_ptr(), _stream())
        ctx.save_for_backward(logits, labels, stats)
        ctx.ignore_index = ignore_index
        ctx.eps = eps
        ctx.pw, ctx.red = pos_weight, reduction
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, labels, stats = ctx.saved_tensors
        g = g.contiguous().float()
        d = torch.empty_like(logits)
        _timed_call('resample_loss', 16.0 * logits.numel(), 'evk_bce_bwd_ex', logits.data_ptr(), labels.data_ptr(), logits.numel(), ctx.ignore_index, ctx.eps,
                    ctx.pw, ctx.red, stats.data_ptr(), g.data_ptr(), d.data_ptr(), 0, _stream())
        return d, None, None, None, None, None


def bce_with_logits(y_pred, y_true, ignore_index=255, label_smoothing=0.0, pos_weight=None, reduction='mean'):
    """reference ever/module/loss.py:229-235 (reduction 'mean' | 'sum', optional pos_weight: a number or a one-element
    tensor — the heads on this path have one logit channel); label_smoothing > 0 gives
    label_smoothing_binary_cross_entropy (loss.py:222-226)."""
    if reduction not in ('mean', 'sum', 'none'):
        raise ValueError(f"binary_cross_entropy_with_logits: reduction '{reduction}'")
    if pos_weight is None:
        pw = 1.0
    elif isinstance(pos_weight, torch.Tensor):
        if pos_weight.numel() != 1:
            raise NotImplementedError('binary_cross_entropy_with_logits: pos_weight must have one element (one logit channel)')
        pw = float(pos_weight.reshape(()).item())
    else:
        pw = float(pos_weight)
    _require_cuda(y_pred, 'binary_cross_entropy_with_logits')
    if y_pred.dim() == 4:
        if y_pred.shape[1] != 1:
            raise ValueError('binary_cross_entropy_with_logits: logits must have one channel')
        y_pred = as_nhwc(y_pred, 'bce')
    else:
        y_pred = y_pred.contiguous()
    labels = _labels(y_true, y_pred.numel(), 'bce')
    if reduction == 'none':
        # One value per NON-ignored pixel, in pixel order (the reference compacts logits and targets with masked_select first,
        # loss.py:10-17,229-235): a data-dependent shape, so the compaction is a boolean index (one host round trip, as in the
        # reference).  Per pixel BCE(z, t) = t * CE([0, z], 1) + (1 - t) * CE([0, z], 0): the two-class per-pixel cross entropy
        # kernel on the logit pair (0, z), which is the same softplus arithmetic.
        from . import functional_next as HN
        z = y_pred.reshape(-1, 1, 1, 1)
        z2 = as_nhwc(torch.cat([torch.zeros_like(z), z], dim=1), 'bce.none')
        yl = labels.reshape(-1, 1, 1)
        valid = yl != ignore_index
        if not label_smoothing and pw == 1.0:
            per = HN.cross_entropy_per_pixel(z2, yl, ignore_index)
        else:
            l1 = HN.cross_entropy_per_pixel(z2, torch.where(valid, torch.ones_like(yl), yl), ignore_index)
            l0 = HN.cross_entropy_per_pixel(z2, torch.where(valid, torch.zeros_like(yl), yl), ignore_index)
            t = yl.to(torch.float32)
            if label_smoothing:
                t = torch.where(yl == 0, t + label_smoothing, t - label_smoothing)
            per = pw * t * l1 + (1.0 - t) * l0
        return per.reshape(-1)[valid.reshape(-1)]
    return _BceFn.apply(y_pred, labels, int(ignore_index), float(label_smoothing), pw, 0 if reduction == 'mean' else 1)


_rank_sum_hook = None  # tests: callable(tensor, what) -> world size, summing `tensor` in place over virtual ranks


def _sum_over_ranks(t, what):
    """In-place SUM of a device tensor over the data-parallel ranks (RCCL all-reduce, asynchronous to the host);
    returns the world size.  `_rank_sum_hook` replaces the collective in single-process tests."""
    if _rank_sum_hook is not None:
        return _rank_sum_hook(t, what)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
        return dist.get_world_size()
    return 1


class _DiceFn(Function):
    @staticmethod
    def forward(ctx, logits, labels, smooth, ignore_index, ignore_channel, sync):
        n, c, h, w = logits.shape
        npix = n * h * w
        stats = _stats_buf(2 * c, logits.device)
        st = _stream()
        _timed_call('resample_loss', (4.0 * c + 8.0) * npix, 'evk_dice_stats', logits.data_ptr(), labels.data_ptr(), npix, c, ignore_index, stats.data_ptr(), st)
        # loss.py:46-48: inter / z summed over the ranks before the ratio.  The collective is enqueued behind the
        # statistics kernel and the finishing kernel behind it: the host never waits for it.
        world = _sum_over_ranks(stats[:2 * c], 'dice_stats') if sync else 1
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        _C.call('evk_dice_finish', stats.data_ptr(), c, float(smooth), ignore_channel, loss.data_ptr(), st)
        ctx.save_for_backward(logits, labels, stats)
        ctx.cfg = (float(smooth), ignore_index, ignore_channel, world)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, labels, stats = ctx.saved_tensors
        smooth, ignore_index, ignore_channel, world = ctx.cfg
        n, c, h, w = logits.shape
        g = g.contiguous().float()
        if world > 1:
            # backward of torch.distributed.nn.all_reduce(SUM) is an all_reduce(SUM) of the upstream grads
            g = g.clone()
            _sum_over_ranks(g, 'dice_grad')
        d = torch.empty_like(logits)
        _timed_call('resample_loss', (8.0 * c + 8.0) * n * h * w, 'evk_dice_bwd', logits.data_ptr(), labels.data_ptr(), n * h * w, c, ignore_index, stats.data_ptr(),
                smooth, ignore_channel, g.data_ptr(), d.data_ptr(), 0, _stream())
        return d, None, None, None, None, None


def dice_loss_with_logits(y_pred, y_true, smooth_value=1.0, ignore_index=255, ignore_channel=-1,
                          sync_statistics=True):
    """reference ever/module/loss.py:40-75."""
    _require_cuda(y_pred, 'dice_loss_with_logits')
    if y_pred.dim() != 4 or y_true.dim() != 3:
        raise AssertionError('dice_loss_with_logits expects y_pred [N,C,H,W] and y_true [N,H,W]')
    y_pred = as_nhwc(y_pred, 'dice')
    labels = _labels(y_true, y_pred.numel() // y_pred.shape[1], 'dice')
    return _DiceFn.apply(y_pred, labels, smooth_value, int(ignore_index), int(ignore_channel), bool(sync_statistics))


class _CeFn(Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index, eps):
        n, c, h, w = logits.shape
        stats = _stats_buf(3, logits.device)
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        _C.call('evk_ce_fwd', logits.data_ptr(), labels.data_ptr(), n * h * w, c, ignore_index, eps, loss.data_ptr(),
                stats.data_ptr(), _stream())
        ctx.save_for_backward(logits, labels, stats)
        ctx.cfg = (ignore_index, eps)
        count = stats[1:2].to(torch.float32).reshape(())     # valid pixels (for reduction='sum'); not differentiable
        ctx.mark_non_differentiable(count)
        return loss, count

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _gcount=None):
        logits, labels, stats = ctx.saved_tensors
        ignore_index, eps = ctx.cfg
        n, c, h, w = logits.shape
        g = g.contiguous().float()
        d = torch.empty_like(logits)
        _C.call('evk_ce_bwd', logits.data_ptr(), labels.data_ptr(), n * h * w, c, ignore_index, eps, stats.data_ptr(),
                g.data_ptr(), d.data_ptr(), 0, _stream())
        return d, None, None, None


class _SoftCeFn(Function):
    @staticmethod
    def forward(ctx, logits, target):
        n, c, h, w = logits.shape
        stats = _stats_buf(1, logits.device)
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        _C.call('evk_soft_ce_fwd', logits.data_ptr(), target.data_ptr(), n * h * w, c, loss.data_ptr(),
                stats.data_ptr(), _stream())
        ctx.save_for_backward(logits, target)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        n, c, h, w = logits.shape
        g = g.contiguous().float()
        d = torch.empty_like(logits)
        _C.call('evk_soft_ce_bwd', logits.data_ptr(), target.data_ptr(), n * h * w, c, g.data_ptr(), d.data_ptr(),
                _stream())
        return d, None


def soft_cross_entropy(y_pred, target):
    """reference ever/module/loss.py:238-242 (target is a per-pixel distribution [N,C,H,W]; no gradient to it)."""
    _require_cuda(y_pred, 'soft_cross_entropy')
    assert y_pred.dim() == 4 and target.dim() == 4
    y_pred, target = as_nhwc(y_pred, 'soft_ce'), as_nhwc(target.detach(), 'soft_ce.target')
    return _SoftCeFn.apply(y_pred, target)


def cross_entropy(y_pred, y_true, ignore_index=255, label_smoothing=0.0, reduction='mean'):
    """F.cross_entropy(ignore_index=...) / label_smoothing_cross_entropy (reference loss.py:207-219).
    reduction 'sum' = the mean times the number of valid pixels, a device word of the same kernel (no host round trip; 0 when
    every pixel is ignored, as the reference's sums over nothing).  'none': per pixel, 0 on ignored pixels — the reference's
    own expression (a compacted 1-D term plus an uncompacted one, loss.py:213-219) is defined only where nothing is ignored,
    and equals this there; built from the per-pixel cross entropy: -sum_c q_c log p_c with q = (1 - eps) onehot + eps / C,
    the uniform part as the sum over the C constant-label cross entropies."""
    _require_cuda(y_pred, 'cross_entropy')
    if reduction not in ('mean', 'sum', 'none'):
        raise ValueError(f"cross_entropy: reduction '{reduction}'")
    squeeze = y_pred.dim() == 2          # flat [M, C] logits with [M] targets (the form the reference's 'none' is defined on)
    if squeeze:
        y_pred = y_pred.reshape(y_pred.shape[0], y_pred.shape[1], 1, 1)
    y_pred = as_nhwc(y_pred, 'cross_entropy')
    if reduction == 'none':
        from . import functional_next as HN
        n, c, h, w = y_pred.shape
        yt = y_true.to(torch.int64).reshape(n, h, w)
        out = HN.cross_entropy_per_pixel(y_pred, yt, ignore_index)
        if label_smoothing:
            valid = yt != ignore_index
            uni = None
            for k in range(c):
                t = HN.cross_entropy_per_pixel(y_pred, torch.where(valid, torch.full_like(yt, k), yt), ignore_index)
                uni = t if uni is None else uni + t
            out = out * (1.0 - label_smoothing) + uni * (label_smoothing / c)
        return out.reshape(y_true.shape)
    labels = _labels(y_true, y_pred.numel() // y_pred.shape[1], 'cross_entropy')
    mean, count = _CeFn.apply(y_pred, labels, int(ignore_index), float(label_smoothing))
    if reduction == 'mean':
        return mean
    return torch.where(count > 0, mean * count, torch.zeros_like(mean))


# ------------------------------------------------------------------ SURVEY §8 f2 / f3 rows
class _ProbStatsFn(Function):
    """(tp, sum_p, sum_y) per class over the valid pixels as a float32 [3, C] tensor; backward is the adjoint
    kernel, so any differentiable function of the statistics (tversky, dice variants) trains through it."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        n, c, h, w = logits.shape
        lib = _C.load()
        stats = torch.empty((lib.evk_prob_stats_doubles(c),), device=logits.device, dtype=torch.float64)
        _C.call('evk_prob_stats', logits.data_ptr(), labels.data_ptr(), n * h * w, c, ignore_index, stats.data_ptr(),
                _stream())
        ctx.save_for_backward(logits, labels)
        ctx.ignore_index = ignore_index
        return stats[:3 * c].reshape(3, c)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, labels = ctx.saved_tensors
        n, c, h, w = logits.shape
        g = g.float().contiguous()
        d = torch.empty_like(logits)
        _C.call('evk_prob_stats_bwd', logits.data_ptr(), labels.data_ptr(), n * h * w, c, ctx.ignore_index,
                g[0].data_ptr(), g[1].data_ptr(), d.data_ptr(), 0, _stream())
        return d, None, None


def prob_stats(y_pred, y_true, ignore_index=255):
    _require_cuda(y_pred, 'prob_stats')
    y_pred = as_nhwc(y_pred, 'prob_stats')
    labels = _labels(y_true, y_pred.numel() // y_pred.shape[1], 'prob_stats')
    return _ProbStatsFn.apply(y_pred, labels, int(ignore_index))


def _all_reduce_sum_differentiable(t):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        import torch.distributed.nn as dist_nn
        return dist_nn.all_reduce(t)
    return t


def tversky_loss_with_logits(y_pred, y_true, alpha, beta=None, gamma=1.0, smooth_value=1.0, ignore_index=255,
                             reduction='mean', sync_statistics=True):
    """reference ever/module/loss.py:78-143: statistics by the HIP kernel, the C-element ratio by autograd."""
    st = prob_stats(y_pred, y_true, ignore_index)  # float64 [3, C]
    tp, sp, sy = st[0], st[1], st[2]
    if isinstance(alpha, (list, tuple)):
        alpha = torch.as_tensor(alpha, dtype=st.dtype, device=st.device)
    if beta is None:
        beta = 1. - alpha
    fp, fn = sp - tp, sy - tp
    num, den = tp, tp + alpha * fn + beta * fp
    if sync_statistics:
        num, den = _all_reduce_sum_differentiable(num), _all_reduce_sum_differentiable(den)
    coeff = (num + smooth_value) / (den + smooth_value)
    loss = ((1. - coeff) ** gamma).float()
    if reduction == 'mean':
        return loss.mean()
    if reduction == 'none':
        return loss
    raise ValueError(f'unknown reduction: {reduction}')


class _FocalFn(Function):
    @staticmethod
    def forward(ctx, logits, target, gamma, alpha, mode, mean):
        n = logits.numel()
        stats = torch.empty((1 + 256,), device=logits.device, dtype=torch.float64)
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        _C.call('evk_focal_fwd', logits.data_ptr(), target.data_ptr(), n, gamma, alpha, mode, mean, loss.data_ptr(),
                stats.data_ptr(), _stream())
        ctx.save_for_backward(logits, target)
        ctx.cfg = (gamma, alpha, mode, mean)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        gamma, alpha, mode, mean = ctx.cfg
        g = g.contiguous().float()
        d = torch.empty_like(logits)
        _C.call('evk_focal_bwd', logits.data_ptr(), target.data_ptr(), logits.numel(), gamma, alpha, mode, mean,
                g.data_ptr(), d.data_ptr(), _stream())
        return d, None, None, None, None, None


def _focal(y_pred, y_true, gamma, alpha, mode, mean, what):
    _require_cuda(y_pred, what)
    if y_pred.shape != y_true.shape:
        raise ValueError(f'{what}: logits {tuple(y_pred.shape)} and targets {tuple(y_true.shape)} must have the same shape')
    yp = y_pred if y_pred.is_contiguous() or (y_pred.dim() == 4 and is_nhwc(y_pred)) else y_pred.contiguous()
    # element-wise: any common dense layout works as long as both operands share it
    yt = y_true.detach().float()
    if yt.stride() != yp.stride():
        yt = torch.empty_like(yp).copy_(yt)
    return _FocalFn.apply(yp, yt, float(gamma), float(alpha), int(mode), int(mean))


def focal_loss(y_pred, y_true, gamma=2.0, normalize=False):
    """reference loss.py:158-176"""
    return _focal(y_pred, y_true, gamma, -1.0, 2 if normalize else 0, 0 if normalize else 1, 'focal_loss')


def sigmoid_focal_loss(y_pred, y_true, alpha=-1, gamma=2, reduction='mean'):
    """reference loss.py:179-201 (fvcore form)"""
    if reduction not in ('mean', 'sum'):
        raise NotImplementedError("sigmoid_focal_loss: reduction must be 'mean' or 'sum' on the HIP path")
    return _focal(y_pred, y_true, gamma, alpha, 1, 1 if reduction == 'mean' else 0, 'sigmoid_focal_loss')


def confusion_matrix_update(cm, y_true, y_pred=None, logits=None):
    """cm (int64 [C, C], cuda) += counts.  Either integer predictions or NCHW logits (threshold / argmax fused)."""
    c = cm.shape[0]
    assert cm.is_cuda and cm.dtype == torch.int64 and cm.is_contiguous() and cm.shape == (c, c)
    yt = y_true.to(device=cm.device, dtype=torch.int64).contiguous()
    if logits is not None:
        _require_cuda(logits, 'confusion_matrix')
        lg = as_nhwc(logits.detach(), 'confusion_matrix')
        cl = lg.shape[1]
        if yt.numel() != lg.numel() // cl:
            raise ValueError('confusion_matrix: label / logit pixel counts differ')
        _C.call('evk_confusion_from_logits', lg.data_ptr(), yt.data_ptr(), yt.numel(), cl, c, cm.data_ptr(), _stream())
    else:
        yp = y_pred.to(device=cm.device, dtype=torch.int64).contiguous()
        if yp.numel() != yt.numel():
            raise ValueError('confusion_matrix: y_true and y_pred sizes differ')
        _C.call('evk_confusion_matrix', yt.data_ptr(), yp.data_ptr(), yt.numel(), c, cm.data_ptr(), _stream())
    return cm


# ---------------------------------------------------------------------------------------------------------------------
# The no-grad forward of every layer kind as a dispatcher-level operator (hip/oplib.py): what `torch.jit.trace`
# (reference api/infer_tool.py:70-74: export_model) and a compiler's shape pass record instead of an opaque Python call.
# Eager calls keep the direct path; the names below are what the modules (and this file) call from here on.
from . import oplib as _oplib  # noqa: E402


def _conv_out(h, k, s, p, d):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


def _l2(v):
    return [int(e) for e in _pair(v)]


_conv2d_plain = conv2d
conv2d = _oplib.traceable(
    'conv2d', '(Tensor x, Tensor weight, Tensor? bias, int[] stride, int[] padding, int[] dilation, bool relu) -> Tensor',
    _conv2d_plain, impl_fn=lambda x, w, b, s, p, d, relu: _conv2d_plain(x, w, b, tuple(s), tuple(p), tuple(d), relu=relu),
    adapt=lambda x, weight, bias=None, stride=1, padding=0, dilation=1, relu=False, bn_stats=False, grad_slot=None:
        (x, weight, bias, _l2(stride), _l2(padding), _l2(dilation), bool(relu)),
    fake=lambda x, w, b, s, p, d, relu: _oplib.nhwc_like(
        x, x.shape[0], w.shape[0], _conv_out(x.shape[2], w.shape[2], s[0], p[0], d[0]), _conv_out(x.shape[3], w.shape[3], s[1], p[1], d[1])))

_bn_act_plain = batch_norm_act
batch_norm_act = _oplib.traceable(
    'batch_norm_eval', '(Tensor x, Tensor? weight, Tensor? bias, Tensor running_mean, Tensor running_var, float eps, '
                       'Tensor? residual, bool relu) -> Tensor',
    _bn_act_plain, impl_fn=lambda x, w, b, rm, rv, eps, res, relu: _bn_act_plain(x, w, b, rm, rv, False, 0.1, eps, residual=res, relu=relu),
    adapt=lambda x, weight, bias, running_mean, running_var, training, momentum, eps, residual=None, relu=False, pack_out=False,
        lazy_res=False: (x, weight, bias, running_mean, running_var, float(eps), residual, bool(relu)),
    fake=lambda x, w, b, rm, rv, eps, res, relu: _oplib.nhwc_like(x, *x.shape),
    applies=lambda x, weight, bias, running_mean, running_var, training, *a, **k: not training and running_mean is not None)


def _same_shape_fake(x, *rest):
    return _oplib.nhwc_like(x, *x.shape) if x.dim() == 4 else torch.empty_like(x)


_relu_plain, _add_plain, _max_pool_plain, _nearest_add_plain = relu, add, max_pool3x3s2, upsample_nearest2x_add
_bilinear_plain, _gap_plain, _relation_plain, _mean4_plain, _stem_plain = upsample_bilinear, global_avg_pool, fs_relation, mean4, stem_conv7x7s2
relu = _oplib.traceable('relu', '(Tensor x) -> Tensor', _relu_plain, adapt=lambda x: (x,), fake=_same_shape_fake)
add = _oplib.traceable('add', '(Tensor a, Tensor b) -> Tensor', _add_plain, adapt=lambda a, b: (a, b), fake=_same_shape_fake)
max_pool3x3s2 = _oplib.traceable(
    'max_pool3x3s2', '(Tensor x) -> Tensor', _max_pool_plain, adapt=lambda x: (x,),
    fake=lambda x: _oplib.nhwc_like(x, x.shape[0], x.shape[1], _conv_out(x.shape[2], 3, 2, 1, 1), _conv_out(x.shape[3], 3, 2, 1, 1)))
upsample_nearest2x_add = _oplib.traceable(
    'upsample_nearest2x_add', '(Tensor top, Tensor lateral) -> Tensor', _nearest_add_plain, adapt=lambda top, lateral: (top, lateral),
    fake=lambda top, lateral: _oplib.nhwc_like(lateral, *lateral.shape))
upsample_bilinear = _oplib.traceable(
    'upsample_bilinear', '(Tensor x, float scale_h, float scale_w) -> Tensor',
    _bilinear_plain, impl_fn=lambda x, sh, sw: _bilinear_plain(x, (sh, sw)),
    adapt=lambda x, scale_factor: (x,) + tuple(float(s) for s in ((scale_factor, scale_factor) if not isinstance(
        scale_factor, (tuple, list)) else scale_factor)),
    fake=lambda x, sh, sw: _oplib.nhwc_like(x, x.shape[0], x.shape[1], int(x.shape[2] * sh), int(x.shape[3] * sw)))
global_avg_pool = _oplib.traceable('global_avg_pool', '(Tensor x) -> Tensor', _gap_plain, adapt=lambda x: (x,),
                                   fake=lambda x: _oplib.nhwc_like(x, x.shape[0], x.shape[1], 1, 1))
fs_relation = _oplib.traceable('fs_relation', '(Tensor scene, Tensor content, Tensor feat) -> Tensor', _relation_plain,
                               adapt=lambda scene, content, feat: (scene, content, feat),
                               fake=lambda scene, content, feat: _oplib.nhwc_like(feat, *feat.shape))
mean4 = _oplib.traceable('mean4', '(Tensor a, Tensor b, Tensor c, Tensor d) -> Tensor', _mean4_plain,
                         adapt=lambda a, b, c, d: (a, b, c, d), fake=lambda a, b, c, d: _oplib.nhwc_like(a, *a.shape))
stem_conv7x7s2 = _oplib.traceable(
    'stem_conv7x7s2', '(Tensor x, Tensor weight) -> Tensor', _stem_plain, impl_fn=lambda x, w: _stem_plain(x, w, False),
    adapt=lambda x, weight, bn_stats=False: (x, weight),
    fake=lambda x, w: _oplib.nhwc_like(x, x.shape[0], w.shape[0], _conv_out(x.shape[2], 7, 2, 3, 1), _conv_out(x.shape[3], 7, 2, 3, 1)))
