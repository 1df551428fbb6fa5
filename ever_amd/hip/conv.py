"""Convolution family of the host wrappers (reference call sites: _resnets.py:21-29 conv3x3 / conv1x1, fpn.py:52-53,
fs_relation.py:22-23; include/ever_hip.h: evk_conv2d_*): forward, data / weight gradients, fork nodes, gradient slots,
transposed convolution, the ResNet stem.  Part of the hip/functional.py facade."""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _C
from . import timing, weight_planes
from .workspace import workspace
from ._base import (  # noqa: F401
    HipPathError, _BN_EPILOGUE, _PACKED, _amax_zeroed, _entry, _f16x2, _is_packed, _lazy_bits, _note_amax, _planes_math,
    _ptr, _require_cuda, _stream, _weight_planes, absmax_bits, as_nhwc, empty_nhwc, get_conv_math, is_nhwc,
    materialize_lazy, observers_active, relu_bits_stats, _conv_out,
)
from .streams import (  # noqa: F401
    _WGRAD_BATCH, _WGRAD_EARLY_MODE, _WGRAD_OWNED, _WGRAD_QUEUE, _WGRAD_SHARED, _WGRAD_SHARED_ON, _WGRAD_STREAM,
    _note_param_use, _wgrad_hold, _wgrad_side_stream, flush_wgrad_queue,
)


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
                           f'(math {get_conv_math()}, Cin {cin}, {n * d.Ho * d.Wo} output rows)')
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
        elif get_conv_math() == 'bf16':
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
            if side is not None and _WGRAD_BATCH[0] > 1 and x3 and _f16x2():
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
                if len(_WGRAD_QUEUE.get(dev, ())) >= _WGRAD_BATCH[0]:
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
        from .pointwise import add      # (pointwise imports this module)
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
        elif get_conv_math() == 'bf16':
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


# ---------------------------------------------------------------------------------------------------------------------
# The no-grad forward of this family's layers as dispatcher-level operators (hip/oplib.py): what `torch.jit.trace`
# (reference api/infer_tool.py:70-74: export_model) and a compiler's shape pass record instead of an opaque Python call.
# Eager calls keep the direct path; the names below are what the modules (and this file) call from here on.
from . import oplib as _oplib  # noqa: E402


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

_stem_plain = stem_conv7x7s2
stem_conv7x7s2 = _oplib.traceable(
    'stem_conv7x7s2', '(Tensor x, Tensor weight) -> Tensor', _stem_plain, impl_fn=lambda x, w: _stem_plain(x, w, False),
    adapt=lambda x, weight, bn_stats=False: (x, weight),
    fake=lambda x, w: _oplib.nhwc_like(x, x.shape[0], w.shape[0], _conv_out(x.shape[2], 7, 2, 3, 1), _conv_out(x.shape[3], 7, 2, 3, 1)))
