"""Packed activation operands of the f16x2 arithmetic (include/ever_hip.h: evk_pack_f16x2): a producer that knows the
operand scale stores every element already split (h | l << 16); the convolution kernels then only permute bytes while
staging.  Contract checked here: under the SAME scale buffer a packed operand gives bit-identical results to the fp32
operand, on every kernel family (halo 3x3, wave-specialised and single-role implicit GEMM, strided residue classes,
both weight-gradient kernels), and unpack(pack(x)) is x to 22 bits."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _scale_buf(lib, t, aws, st):
    from ever_amd import _C
    b = torch.zeros(int(lib.evk_absmax_words()), dtype=torch.int32, device=t.device)
    _C.call('evk_absmax', t.data_ptr(), t.numel(), b.data_ptr(), aws.data_ptr(), st)
    return b


def _pack(t, bits, st):
    from ever_amd import _C
    out = torch.empty(t.shape, dtype=torch.int32, device=t.device)
    _C.call('evk_pack_f16x2', t.data_ptr(), t.numel(), bits.data_ptr(), out.data_ptr(), st)
    return out


def test_pack_round_trip_keeps_22_bits(cuda):
    from ever_amd import _C
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=cuda)
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(1 << 16, generator=g) * torch.logspace(-6, 2, 1 << 16)).to(cuda)
    bits = _scale_buf(lib, x, aws, st)
    pk = _pack(x, bits, st)
    back = torch.empty_like(x)
    _C.call('evk_unpack_f16x2', pk.data_ptr(), x.numel(), bits.data_ptr(), back.data_ptr(), st)
    torch.cuda.synchronize()
    big = x.abs().max().item()
    err = (back.double() - x.double()).abs()
    # 22 bits relative for elements within 2^-17 of the largest, an absolute 2^-38 of the largest below
    bound = torch.maximum(x.double().abs() * 2.0 ** -21, torch.full_like(err, big * 2.0 ** -37))
    assert bool((err <= bound).all()), float((err / bound).max())
    zeros = torch.zeros(1024, device=cuda)
    assert int(_pack(zeros, bits, st).abs().max()) == 0


CASES = [
    # n, cin, h, w, cout, k, stride, pad
    (2, 64, 32, 32, 64, 3, 1, 1),       # halo 3x3
    (2, 256, 32, 32, 256, 3, 1, 1),     # halo 3x3, wide
    (2, 128, 16, 16, 128, 3, 2, 1),     # strided 3x3: residue classes in the data gradient
    (4, 256, 32, 32, 512, 1, 1, 0),     # 1x1 wave-specialised
    (2, 64, 16, 16, 256, 1, 1, 0),      # 1x1 single-role
    (2, 256, 64, 64, 128, 1, 2, 0),     # strided 1x1 shortcut
    (1, 32, 24, 40, 48, 3, 1, 1),       # ragged tiles
]


@pytest.mark.parametrize('case', CASES)
def test_packed_operands_are_bit_identical(cuda, case):
    from ever_amd import _C
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    n, cin, h, w, cout, k, s, p = case
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    d = _C.ConvDesc(n, h, w, cin, ho, wo, cout, k, k, s, s, p, p, 1, 1)
    g = torch.Generator().manual_seed(cin * 7 + cout + k + s)
    x = (torch.randn(n, h, w, cin, generator=g) + 0.25).to(cuda)
    wt = (torch.randn(cout, k, k, cin, generator=g) * 0.05).to(cuda)
    dy = (torch.randn(n, ho, wo, cout, generator=g) * 1e-3).to(cuda)
    aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=cuda)
    bx, bw, bdy = (_scale_buf(lib, t, aws, st) for t in (x, wt, dy))
    xp, dyp = _pack(x, bx, st), _pack(dy, bdy, st)
    pf = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=cuda)
    pd = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 1), dtype=torch.uint8, device=cuda)
    _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 0, pf.data_ptr(), bw.data_ptr(), st)
    _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 1, pd.data_ptr(), bw.data_ptr(), st)
    zero = ctypes.c_int32(0)

    def fwd(src, flags):
        y = torch.empty(n, ho, wo, cout, device=cuda)
        _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), src.data_ptr(), bx.data_ptr(), pf.data_ptr(), bw.data_ptr(), None,
                None, y.data_ptr(), flags, None, 0, ctypes.byref(zero), None, st)
        return y

    def dgrad(src, flags):
        dx = torch.empty_like(x)
        _C.call('evk_conv2d_dgrad_f16x2_ex', ctypes.byref(d), src.data_ptr(), bdy.data_ptr(), pd.data_ptr(), bw.data_ptr(),
                None, dx.data_ptr(), None, flags, st)
        return dx

    wsb = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d))
    wsp = torch.empty(max(wsb, 16), dtype=torch.uint8, device=cuda)

    def wgrad(xs, dys, flags):
        dw = torch.empty_like(wt)
        _C.call('evk_conv2d_wgrad_f16x2_ex', ctypes.byref(d), xs.data_ptr(), bx.data_ptr(), dys.data_ptr(), bdy.data_ptr(),
                dw.data_ptr(), None, wsp.data_ptr(), wsb, flags, st)
        return dw

    X, DY = 2, 4   # EVK_CONV_X_PACKED, EVK_CONV_DY_PACKED
    y0, y1 = fwd(x, 0), fwd(xp, X)
    g0, g1 = dgrad(dy, 0), dgrad(dyp, DY)
    w0 = wgrad(x, dy, 0)
    torch.cuda.synchronize()
    assert y0.abs().max().item() > 0 and g0.abs().max().item() > 0 and w0.abs().max().item() > 0
    assert torch.equal(y0, y1), (y0 - y1).abs().max().item()
    assert torch.equal(g0, g1), (g0 - g1).abs().max().item()
    for flags, xs, dys in ((X, xp, dy), (DY, x, dyp), (X | DY, xp, dyp)):
        w1 = wgrad(xs, dys, flags)
        torch.cuda.synchronize()
        assert torch.equal(w0, w1), (flags, (w0 - w1).abs().max().item())


def test_packed_dy_refuses_a_bias_gradient(cuda):
    from ever_amd import _C
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    d = _C.ConvDesc(1, 8, 8, 32, 8, 8, 32, 1, 1, 1, 1, 0, 0, 1, 1)
    x = torch.randn(1, 8, 8, 32, device=cuda)
    aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=cuda)
    b = _scale_buf(lib, x, aws, st)
    dw, db = torch.empty(32, 1, 1, 32, device=cuda), torch.empty(32, device=cuda)
    wsb = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d))
    wsp = torch.empty(max(wsb, 16), dtype=torch.uint8, device=cuda)
    with pytest.raises(Exception):
        _C.call('evk_conv2d_wgrad_f16x2_ex', ctypes.byref(d), x.data_ptr(), b.data_ptr(), x.data_ptr(), b.data_ptr(),
                dw.data_ptr(), db.data_ptr(), wsp.data_ptr(), wsb, 4, st)


def _chain(F, x, ws, bns, pack):
    """conv -> BN -> ReLU three times (3x3, 1x1, strided 3x3), as the residual blocks chain them"""
    prev = F._PACKED
    F._PACKED = pack
    try:
        cfgs = ((1, 1), (1, 0), (2, 1))
        h = x
        for i, (w, (gm, bt), (s, p)) in enumerate(zip(ws, bns, cfgs)):
            h = F.conv2d(h, w, None, stride=s, padding=p, bn_stats=True)
            # the first two results are read by the next convolution alone: stored packed (EVK_BN_PACK_Y)
            h = F.batch_norm_act(h, gm, bt, None, None, True, 0.1, 1e-5, relu=True, pack_out=i < 2)
        loss = (h * h).sum()
        grads = torch.autograd.grad(loss, [x] + list(ws) + [t for pr in bns for t in pr])
        return loss.detach(), [g.detach().clone() for g in grads]
    finally:
        F._PACKED = prev


def test_batchnorm_passes_hand_the_convolutions_packed_operands(cuda):
    """dx of a BatchNorm that follows a convolution is written packed (EVK_BN_PACK_DX) and read packed by that
    convolution's two gradients; the output of a BatchNorm that only a convolution reads is written packed
    (EVK_BN_PACK_Y) and read packed by its forward and weight gradient: same loss, gradients equal to the fp32-operand
    run within the f16x2 tolerance, and both within it of a float64 reference."""
    from ever_amd.hip import functional as F
    import torch.nn.functional as TF
    prev = F.set_conv_math('f16x2')
    try:
        g = torch.Generator().manual_seed(21)
        x = torch.randn(4, 64, 32, 32, generator=g).to(cuda).requires_grad_()
        shapes = ((64, 64, 3), (128, 64, 1), (128, 128, 3))
        ws = [(torch.randn(o, i, k, k, generator=g) / (i * k * k) ** 0.5).to(cuda)
              .contiguous(memory_format=torch.channels_last).requires_grad_() for o, i, k in shapes]
        bns = [((torch.rand(o, generator=g) + 0.5).to(cuda).requires_grad_(),
                (torch.randn(o, generator=g) * 0.1).to(cuda).requires_grad_()) for o, _, _ in shapes]
        F.absmax_stats.update(hits=0, standalone=0, fused=0, packed=0)
        l1, g1 = _chain(F, x, ws, bns, True)
        assert F.absmax_stats['packed'] == 5, F.absmax_stats    # three dx, two y
        seen = F.absmax_stats['packed']
        l0, g0 = _chain(F, x, ws, bns, False)
        assert F.absmax_stats['packed'] == seen
        # float64 reference
        xd = x.detach().double().cpu().requires_grad_()
        wd = [w.detach().double().cpu().requires_grad_() for w in ws]
        bd = [(a.detach().double().cpu().requires_grad_(), b.detach().double().cpu().requires_grad_()) for a, b in bns]
        h = xd
        for w, (gm, bt), (s, p) in zip(wd, bd, ((1, 1), (1, 0), (2, 1))):
            h = TF.relu(TF.batch_norm(TF.conv2d(h, w, None, s, p), None, None, gm, bt, True, 0.1, 1e-5))
        lr = (h * h).sum()
        gr = torch.autograd.grad(lr, [xd] + wd + [t for pr in bd for t in pr])
        torch.cuda.synchronize()
        assert abs(l1.item() - lr.item()) <= 2e-6 * abs(lr.item())
        for a, b, r in zip(g1, g0, gr):
            scale = r.abs().max().item()
            ea = (a.double().cpu() - r).abs().max().item() / scale
            eb = (b.double().cpu() - r).abs().max().item() / scale
            assert ea < 2e-5 and eb < 2e-5, (ea, eb)
            assert ea < 4 * eb + 1e-6, (ea, eb)     # packing (a bounded, slightly loose scale) costs no accuracy
    finally:
        F.set_conv_math(prev)


@pytest.mark.parametrize('scale', [3e-2, 1.0, 1e10])   # (below ~1e-2 the BatchNorm eps dominates the variance: ill-conditioned)
def test_packed_batchnorm_chain_is_scale_invariant(cuda, scale):
    """the bounds the BatchNorm passes pack under come from statistics records / per-channel maxima: any magnitude of
    the network input must give the same relative accuracy (BatchNorm itself removes the scale after the first layer)"""
    from ever_amd.hip import functional as F
    prev = F.set_conv_math('f16x2')
    try:
        g = torch.Generator().manual_seed(33)
        x0 = torch.randn(2, 64, 32, 32, generator=g)
        shapes = ((64, 64, 3), (128, 64, 1), (128, 128, 3))
        ws = [(torch.randn(o, i, k, k, generator=g) / (i * k * k) ** 0.5).to(cuda)
              .contiguous(memory_format=torch.channels_last).requires_grad_() for o, i, k in shapes]
        bns = [((torch.rand(o, generator=g) + 0.5).to(cuda).requires_grad_(),
                (torch.randn(o, generator=g) * 0.1).to(cuda).requires_grad_()) for o, _, _ in shapes]
        x1 = (x0 * scale).to(cuda).requires_grad_()
        l1, g1 = _chain(F, x1, ws, bns, True)
        l0, g0 = _chain(F, x1, ws, bns, False)
        torch.cuda.synchronize()
        assert torch.isfinite(l1) and abs(l1.item() - l0.item()) <= 5e-6 * abs(l0.item())
        for a, b in zip(g1[1:], g0[1:]):          # weight / BatchNorm gradients (the input gradient scales with 1 / scale)
            ref = b.abs().max().item()
            assert torch.isfinite(a).all()
            assert (a - b).abs().max().item() <= 4e-5 * ref + 1e-30
    finally:
        F.set_conv_math(prev)


def test_packed_batchnorm_all_zero_input(cuda):
    """zero statistics records bound the output by |beta| alone; with beta = 0 the bound is 0 and the packed tensor exact
    zeros (scale 1)"""
    from ever_amd.hip import functional as F
    prev = F.set_conv_math('f16x2')
    try:
        x = torch.zeros(2, 64, 16, 16, device=cuda, requires_grad=True)
        w = torch.randn(64, 64, 3, 3, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
        w2 = torch.randn(64, 64, 1, 1, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
        for beta in (0.0, 0.25):
            gm = torch.ones(64, device=cuda, requires_grad=True)
            bt = torch.full((64,), beta, device=cuda, requires_grad=True)
            h = F.conv2d(x, w, None, stride=1, padding=1, bn_stats=True)
            h = F.batch_norm_act(h, gm, bt, None, None, True, 0.1, 1e-5, relu=True, pack_out=True)
            assert F._is_packed(h)
            y = F.conv2d(h, w2, None)
            torch.cuda.synchronize()
            ref = torch.nn.functional.conv2d(torch.full((2, 64, 16, 16), beta, device=cuda), w2.detach())
            assert torch.allclose(y, ref, rtol=1e-5, atol=1e-6), (y - ref).abs().max().item()
            y.sum().backward()
            torch.cuda.synchronize()
            assert torch.isfinite(w.grad).all() and torch.isfinite(w2.grad).all()
    finally:
        F.set_conv_math(prev)


def test_a_packed_tensor_is_refused_by_readers_that_cannot_take_it(cuda):
    """a packed activation must never be read as fp32: convolutions outside the f16x2 plane kernels raise"""
    from ever_amd.hip import functional as F
    prev = F.set_conv_math('f16x2')
    try:
        x = torch.randn(2, 64, 16, 16, device=cuda)
        w = torch.randn(64, 64, 3, 3, device=cuda).contiguous(memory_format=torch.channels_last)
        gm, bt = torch.ones(64, device=cuda), torch.zeros(64, device=cuda)
        with torch.no_grad():
            h = F.conv2d(x, w, None, stride=1, padding=1, bn_stats=True)
            h = F.batch_norm_act(h, gm, bt, None, None, True, 0.1, 1e-5, relu=True, pack_out=True)
            assert F._is_packed(h)
            F.set_conv_math('bf16x3')
            with pytest.raises(F.HipPathError):
                F.conv2d(h, w, None, padding=1)
    finally:
        F.set_conv_math(prev)


def test_nothing_is_stored_packed_while_an_observer_is_installed(cuda):
    """packed words are marked by a Python attribute on the tensor object only.  Saved-tensor hooks hand the
    backward NEW tensor objects (non-reentrant checkpointing, save_on_cpu), module forward hooks show intermediate
    activations to user code: while either is installed the BatchNorm passes store fp32 — same results as without, and no
    packed tensor is produced."""
    from ever_amd.hip import functional as F
    prev = F.set_conv_math('f16x2')
    try:
        g = torch.Generator().manual_seed(5)
        x = torch.randn(2, 64, 32, 32, generator=g).to(cuda).requires_grad_()
        shapes = ((64, 64, 3), (128, 64, 1), (128, 128, 3))
        ws = [(torch.randn(o, i, k, k, generator=g) / (i * k * k) ** 0.5).to(cuda)
              .contiguous(memory_format=torch.channels_last).requires_grad_() for o, i, k in shapes]
        bns = [((torch.rand(o, generator=g) + 0.5).to(cuda).requires_grad_(),
                (torch.randn(o, generator=g) * 0.1).to(cuda).requires_grad_()) for o, _, _ in shapes]
        l0, g0 = _chain(F, x, ws, bns, False)
        F.absmax_stats.update(hits=0, standalone=0, fused=0, packed=0)
        copies = []

        def pack_hook(t):
            copies.append(1)
            return t.clone()          # a new tensor object with new memory, as an offloading hook would return

        with torch.autograd.graph.saved_tensors_hooks(pack_hook, lambda t: t):
            l1, g1 = _chain(F, x, ws, bns, True)
        assert copies and F.absmax_stats['packed'] == 0, F.absmax_stats
        torch.cuda.synchronize()
        assert abs(l1.item() - l0.item()) <= 1e-6 * abs(l0.item())
        for a, b in zip(g1, g0):
            assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
        # ... and the packed path is back once the hooks are gone
        _chain(F, x, ws, bns, True)
        assert F.absmax_stats['packed'] == 5
    finally:
        F.set_conv_math(prev)


def test_a_folded_convolution_unpacks_a_packed_input(cuda):
    """a training-mode BatchNorm in front of a folded conv+BN pair (mixed train / eval sub-modules) may hand it packed
    words: module/fold.py reads them through evk_unpack_f16x2 instead of as fp32"""
    from ever_amd.hip import functional as F
    prev = F.set_conv_math('f16x2')
    try:
        x = torch.randn(2, 64, 16, 16, device=cuda)
        w = torch.randn(64, 64, 3, 3, device=cuda).contiguous(memory_format=torch.channels_last)
        gm, bt = torch.ones(64, device=cuda), torch.zeros(64, device=cuda)
        with torch.no_grad():
            h = F.conv2d(x, w, None, stride=1, padding=1, bn_stats=True)
            hp = F.batch_norm_act(h, gm, bt, None, None, True, 0.1, 1e-5, relu=True, pack_out=True)
            h = F.conv2d(x, w, None, stride=1, padding=1, bn_stats=True)
            hf = F.batch_norm_act(h, gm, bt, None, None, True, 0.1, 1e-5, relu=True, pack_out=False)
            assert F._is_packed(hp) and not F._is_packed(hf)
            up = F.unpacked(hp)
            assert not F._is_packed(up)
            torch.cuda.synchronize()
            assert (up - hf).abs().max().item() <= 2.0 ** -20 * hf.abs().max().item()
    finally:
        F.set_conv_math(prev)
