"""Per-kernel parity of the HIP path (through the C-ABI) against stock PyTorch fp32 on CPU.

Tolerances: conv / BN results are fp32 with a different summation order than oneDNN -> rtol 1e-4
(the north-star contract is 1e-3 relative); index-producing ops (maxpool) are bit-exact.
"""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-4, atol=1e-5, what=''):
    a = a.detach().cpu().contiguous().double()
    b = b.detach().cpu().contiguous().double()
    assert a.shape == b.shape, f'{what}: shape {a.shape} vs {b.shape}'
    err = (a - b).abs().max().item()
    scale = b.abs().max().item()
    assert err <= atol + rtol * scale, f'{what}: max abs err {err:.3e} vs scale {scale:.3e}'


CONV_CASES = [
    # n, cin, h, w, cout, k, stride, pad, dil, bias
    (2, 64, 16, 16, 64, 1, 1, 0, 1, False),
    (2, 64, 16, 16, 256, 1, 1, 0, 1, True),
    (2, 256, 16, 16, 128, 1, 2, 0, 1, False),   # downsample 1x1 s2 (scatter dgrad)
    (2, 64, 20, 12, 64, 3, 1, 1, 1, False),     # ragged spatial size
    (2, 128, 16, 16, 128, 3, 2, 1, 1, False),   # 3x3 s2
    (1, 64, 16, 16, 96, 3, 1, 2, 2, False),     # dilation 2 (output_stride 16)
    (2, 3, 32, 32, 64, 7, 2, 3, 1, False),      # stem, Cin=3 (channel padding path)
    (2, 4, 32, 32, 64, 7, 2, 3, 1, False),      # 4-band stem
    (2, 256, 8, 8, 1, 1, 1, 0, 1, True),        # classifier Cout=1
    (2, 128, 8, 8, 16, 3, 1, 1, 1, True),       # FarSeg-paper classifier 3x3, 16 classes
    (4, 2048, 1, 1, 256, 1, 1, 0, 1, True),     # scene MLP on 1x1 maps
    (1, 512, 9, 7, 520, 3, 1, 1, 1, False),     # Cout not a multiple of the N tile
    (3, 72, 11, 13, 40, 3, 1, 1, 1, True),      # channels % 8 == 0 only, K = 648 not a multiple of 32 (K padding)
    (2, 24, 24, 24, 264, 3, 2, 1, 1, False),    # stride-2 residue classes with a ragged N tile
    (2, 64, 16, 8, 64, 3, 1, 1, 1, False),      # Wo % 8 == 0: single-decomposition wgrad gather
    (2, 64, 128, 128, 64, 3, 1, 1, 1, True),    # LDS-halo 3x3 kernel, 64-wide N tile, 8x16 patches (256 workgroups)
    (2, 48, 128, 128, 160, 3, 1, 1, 1, False),  # LDS-halo kernel, 8x16 patches, ragged N tile, 3 channel chunks
    (4, 32, 128, 128, 128, 3, 1, 1, 1, True),   # LDS-halo kernel, 16x16 patches
    (3, 32, 72, 256, 64, 3, 1, 1, 1, False),    # LDS-halo kernel, non-square, H % 16 != 0 (8-row patches only)
    (3, 16, 96, 320, 128, 3, 1, 1, 1, True),    # LDS-halo kernel, one channel chunk, 16-row patches, W = 20 patches
    (8, 32, 77, 43, 128, 3, 1, 1, 1, True),     # LDS-halo kernel, patches hanging over both edges (H % 8 = 5, W % 16 = 11)
    (2, 96, 154, 86, 128, 3, 1, 1, 1, False),   # LDS-halo kernel, ragged patches, 128-wide tile (the FreeNet scene at stride 4)
    (4, 200, 96, 88, 96, 3, 1, 1, 1, False),    # LDS-halo kernel, 12.5 channel chunks (the FreeNet input convolution), ragged patches
    (8, 24, 64, 64, 64, 3, 1, 1, 1, True),      # LDS-halo kernel, 1.5 channel chunks
    (8, 64, 60, 104, 160, 3, 1, 1, 1, False),   # LDS-halo kernel, 16-row patches with a ragged last row (H % 16 = 12), ragged N tile
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_fwd_bwd(cuda, case, conv_math):
    from ever_amd.hip import functional as F
    n, cin, h, w, cout, k, s, p, d, bias = case
    g = torch.Generator().manual_seed(1234 + cin + cout + k)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = TF.conv2d(xr, wr, br, stride=s, padding=p, dilation=d)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xg = x.to(cuda).requires_grad_(True)
    wg = wt.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bg = b.to(cuda).requires_grad_(True) if bias else None
    yg = F.conv2d(xg, wg, bg, stride=s, padding=p, dilation=d)
    yg.backward(gy.to(cuda))
    torch.cuda.synchronize()
    _close(yg, yr, what='y')
    _close(xg.grad, xr.grad, what='dx')
    _close(wg.grad, wr.grad, what='dw')
    if bias:
        _close(bg.grad, br.grad, what='db')


def test_conv_split_matches_fp64_as_well_as_fp32_mfma(cuda):
    """The split kernels (3-term bf16, 2-term scaled fp16) must be as close to an fp64 evaluation as the exact-fp32 MFMA kernels are
    (forward, data gradient, weight gradient), on operands with a non-zero mean (no cancellation luck)."""
    from ever_amd.hip import functional as F
    g = torch.Generator().manual_seed(99)
    n, cin, h, w, cout, k = 2, 256, 24, 24, 192, 3
    x = torch.randn(n, cin, h, w, generator=g) + 0.5
    wt = (torch.randn(cout, cin, k, k, generator=g) + 0.1) / (cin * k * k) ** 0.5
    gy = torch.randn(n, cout, h, w, generator=g) + 0.25
    x64, w64 = x.double().requires_grad_(), wt.double().requires_grad_()
    y64 = TF.conv2d(x64, w64, None, padding=1)
    y64.backward(gy.double())
    errs = {}
    for mode in ('f32', 'bf16x3', 'f16x2'):
        prev = F.set_conv_math(mode)
        try:
            xg = x.to(cuda).requires_grad_()
            wg = wt.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
            yg = F.conv2d(xg, wg, None, padding=1)
            yg.backward(gy.to(cuda))
            torch.cuda.synchronize()
        finally:
            F.set_conv_math(prev)
        rel = lambda a, b: ((a.detach().cpu().double() - b).abs().max() / b.abs().max()).item()
        errs[mode] = (rel(yg, y64.detach()), rel(xg.grad, x64.grad), rel(wg.grad, w64.grad))
    print('max rel err vs fp64 (y, dx, dw):', errs)
    for split in ('bf16x3', 'f16x2'):
        for e32, e3 in zip(errs['f32'], errs[split]):
            assert e3 <= 2.0 * e32 + 2e-7, errs
            assert e3 < 5e-6, errs


def test_conv2d_relu_epilogue(cuda):
    from ever_amd.hip import functional as F
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 64, 8, 8, generator=g)
    w = torch.randn(32, 64, 1, 1, generator=g) / 8
    b = torch.randn(32, generator=g)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = TF.relu(TF.conv2d(xr, wr, br))
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg, wg, bg = x.to(cuda).requires_grad_(), w.to(cuda).requires_grad_(), b.to(cuda).requires_grad_()
    yg = F.conv2d(xg, wg, bg, relu=True)
    yg.backward(gy.to(cuda))
    _close(yg, yr, what='y')
    _close(xg.grad, xr.grad, what='dx')
    _close(wg.grad, wr.grad, what='dw')
    _close(bg.grad, br.grad, what='db')


@pytest.mark.parametrize('c,h,w,res,relu,train', [
    (64, 16, 16, False, True, True), (256, 8, 8, True, True, True), (2048, 4, 4, False, False, True),
    (64, 9, 7, True, False, True), (128, 8, 8, True, True, False), (12, 5, 5, False, True, True),
])
def test_batch_norm_act(cuda, c, h, w, res, relu, train):
    from ever_amd.hip import functional as F
    g = torch.Generator().manual_seed(c + h)
    n = 3
    x = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
    r = torch.randn(n, c, h, w, generator=g) if res else None
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g)
    rm, rv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5

    xr = x.clone().requires_grad_()
    rr = r.clone().requires_grad_() if res else None
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rm_r, rv_r = rm.clone(), rv.clone()
    yr = TF.batch_norm(xr, rm_r, rv_r, gr, br, training=train, momentum=0.1, eps=1e-5)
    if res:
        yr = yr + rr
    if relu:
        yr = TF.relu(yr)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xg = x.to(cuda).requires_grad_()
    rg = r.to(cuda).requires_grad_() if res else None
    gg, bg = gamma.to(cuda).requires_grad_(), beta.to(cuda).requires_grad_()
    rm_g, rv_g = rm.to(cuda), rv.to(cuda)
    yg = F.batch_norm_act(xg, gg, bg, rm_g, rv_g, train, 0.1, 1e-5, residual=rg, relu=relu)
    yg.backward(gy.to(cuda))
    _close(yg, yr, what='y')
    _close(xg.grad, xr.grad, what='dx', rtol=2e-4)
    _close(gg.grad, gr.grad, what='dgamma', rtol=2e-4)
    _close(bg.grad, br.grad, what='dbeta', rtol=2e-4)
    if res:
        _close(rg.grad, rr.grad, what='dres')
    _close(rm_g, rm_r, what='running_mean')
    _close(rv_g, rv_r, what='running_var')


def test_maxpool(cuda):
    from ever_amd.hip import functional as F
    g = torch.Generator().manual_seed(3)
    for (h, w) in [(16, 16), (15, 9)]:
        x = torch.randn(2, 64, h, w, generator=g)
        xr = x.clone().requires_grad_()
        yr = TF.max_pool2d(xr, 3, 2, 1)
        gy = torch.randn(yr.shape, generator=g)
        yr.backward(gy)
        xg = x.to(cuda).requires_grad_()
        yg = F.max_pool3x3s2(xg)
        yg.backward(gy.to(cuda))
        assert torch.equal(yg.cpu().contiguous(), yr.detach()), 'maxpool fwd must be bit exact'
        assert torch.equal(xg.grad.cpu().contiguous(), xr.grad), 'maxpool bwd must be bit exact'


def test_nearest2x_add(cuda):
    from ever_amd.hip import functional as F
    g = torch.Generator().manual_seed(4)
    top = torch.randn(2, 256, 4, 6, generator=g)
    lat = torch.randn(2, 256, 8, 12, generator=g)
    tr, lr = top.clone().requires_grad_(), lat.clone().requires_grad_()
    yr = lr + TF.interpolate(tr, scale_factor=2, mode='nearest')
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    tg, lg = top.to(cuda).requires_grad_(), lat.to(cuda).requires_grad_()
    yg = F.upsample_nearest2x_add(tg, lg)
    yg.backward(gy.to(cuda))
    assert torch.equal(yg.cpu().contiguous(), yr.detach())
    _close(tg.grad, tr.grad, what='dtop', rtol=1e-6)
    assert torch.equal(lg.grad.cpu().contiguous(), lr.grad)


@pytest.mark.parametrize('c,h,w,s', [(256, 8, 8, 2), (1, 16, 16, 4), (16, 8, 12, 4), (128, 1, 1, 2), (64, 5, 3, 2),
                                     (512, 7, 9, 4), (132, 6, 5, 8), (320, 3, 1, 2)])
def test_bilinear_align_corners(cuda, c, h, w, s):
    from ever_amd.hip import functional as F
    g = torch.Generator().manual_seed(5 + c)
    x = torch.randn(2, c, h, w, generator=g)
    xr = x.clone().requires_grad_()
    yr = torch.nn.UpsamplingBilinear2d(scale_factor=s)(xr)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = x.to(cuda).requires_grad_()
    yg = F.upsample_bilinear(xg, s)
    yg.backward(gy.to(cuda))
    _close(yg, yr, what='y', rtol=1e-6, atol=1e-6)
    _close(xg.grad, xr.grad, what='dx', rtol=1e-5, atol=1e-5)


def test_bilinear_kat(cuda):
    """UpsamplingBilinear2d(2) of [[0,1],[2,3]] (captured from the reference import)."""
    from ever_amd.hip import functional as F
    x = torch.tensor([[0., 1.], [2., 3.]]).reshape(1, 1, 2, 2).to(cuda)
    y = F.upsample_bilinear(x, 2).cpu().reshape(4, 4)
    t = 1. / 3
    exp = torch.tensor([[0, t, 2 * t, 1], [2 * t, 1, 1 + t, 1 + 2 * t], [1 + t, 1 + 2 * t, 2, 2 + t],
                        [2, 2 + t, 2 + 2 * t, 3]])
    _close(y, exp, rtol=1e-6, atol=1e-6)


def test_gap_relation_mean4(cuda):
    from ever_amd.hip import functional as F
    g = torch.Generator().manual_seed(6)
    n, c, h, w = 3, 256, 6, 5
    x = torch.randn(n, 2048, 4, 4, generator=g)
    xr = x.clone().requires_grad_()
    yr = TF.adaptive_avg_pool2d(xr, 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = x.to(cuda).requires_grad_()
    yg = F.global_avg_pool(xg)
    yg.backward(gy.to(cuda))
    _close(yg, yr, what='gap', rtol=1e-5)
    _close(xg.grad, xr.grad, what='gap dx', rtol=1e-6)

    scene = torch.randn(n, c, 1, 1, generator=g) * 0.2
    content = torch.randn(n, c, h, w, generator=g) * 0.3
    feat = torch.randn(n, c, h, w, generator=g)
    sr, cr, fr = scene.clone().requires_grad_(), content.clone().requires_grad_(), feat.clone().requires_grad_()
    rel = torch.sigmoid((sr * cr).sum(dim=1, keepdim=True))
    outr = rel * fr
    go = torch.randn(outr.shape, generator=g)
    outr.backward(go)
    sg, cg, fg = scene.to(cuda).requires_grad_(), content.to(cuda).requires_grad_(), feat.to(cuda).requires_grad_()
    outg = F.fs_relation(sg, cg, fg)
    outg.backward(go.to(cuda))
    _close(outg, outr, what='relation', rtol=1e-5)
    _close(sg.grad, sr.grad, what='dscene', rtol=1e-4)
    _close(cg.grad, cr.grad, what='dcontent', rtol=1e-4)
    _close(fg.grad, fr.grad, what='dfeat', rtol=1e-5)

    ts = [torch.randn(2, 64, 4, 4, generator=g) for _ in range(4)]
    trs = [t.clone().requires_grad_() for t in ts]
    mr = sum(trs) / len(trs)
    gm = torch.randn(mr.shape, generator=g)
    mr.backward(gm)
    tgs = [t.to(cuda).requires_grad_() for t in ts]
    mg = F.mean4(*tgs)
    mg.backward(gm.to(cuda))
    assert torch.equal(mg.cpu().contiguous(), mr.detach())
    for a, b in zip(tgs, trs):
        assert torch.equal(a.grad.cpu().contiguous(), b.grad)


def _ref_bce(y_pred, y_true, ignore_index=255):
    yp, yt = y_pred.reshape(-1), y_true.reshape(-1)
    valid = yt != ignore_index
    return TF.binary_cross_entropy_with_logits(yp[valid], yt[valid].float())


def _ref_dice(y_pred, y_true, smooth=1.0, ignore_index=255, ignore_channel=-1):
    c = y_pred.size(1)
    yp = y_pred.permute(0, 2, 3, 1).reshape(-1, c)
    yt = y_true.reshape(-1)
    valid = yt != ignore_index
    yp, yt = yp[valid], yt[valid]
    w = torch.ones(c, dtype=torch.bool)
    if c == 1:
        prob, tgt = yp.sigmoid(), yt.reshape(-1, 1).float()
    else:
        prob, tgt = yp.log_softmax(dim=1).exp(), TF.one_hot(yt.long(), c).float()
        if ignore_channel != -1:
            w[ignore_channel] = False
    prob, tgt = prob[:, w], tgt[:, w]
    inter = (prob * tgt).sum(0)
    z = prob.sum(0) + tgt.sum(0) + smooth
    return 1. - ((2 * inter + smooth) / z).mean()


@pytest.mark.parametrize('c', [1, 3, 16])
def test_losses(cuda, c):
    from ever_amd.hip import functional as F
    g = torch.Generator().manual_seed(11 + c)
    n, h, w = 2, 24, 20
    logits = torch.randn(n, c, h, w, generator=g) * 2
    labels = torch.randint(0, max(c, 2), (n, h, w), generator=g)
    labels[:, :4, :4] = 255
    if c == 1:
        lr = logits.clone().requires_grad_()
        ref = _ref_bce(lr, labels)
        ref.backward()
        lg = logits.to(cuda).requires_grad_()
        out = F.bce_with_logits(lg, labels.to(cuda))
        out.backward()
        _close(out, ref, what='bce', rtol=1e-5)
        _close(lg.grad, lr.grad, what='dbce', rtol=1e-4, atol=1e-9)
    else:
        lr = logits.clone().requires_grad_()
        ref = TF.cross_entropy(lr, labels, ignore_index=255)
        ref.backward()
        lg = logits.to(cuda).requires_grad_()
        out = F.cross_entropy(lg, labels.to(cuda), ignore_index=255)
        out.backward()
        _close(out, ref, what='ce', rtol=1e-5)
        _close(lg.grad, lr.grad, what='dce', rtol=1e-4, atol=1e-9)
    lr = logits.clone().requires_grad_()
    ref = _ref_dice(lr, labels)
    (ref * 0.5).backward()
    lg = logits.to(cuda).requires_grad_()
    out = F.dice_loss_with_logits(lg, labels.to(cuda))
    (out * 0.5).backward()
    _close(out, ref, what='dice', rtol=1e-5)
    _close(lg.grad, lr.grad, what='ddice', rtol=1e-4, atol=1e-10)


def test_loss_kats(cuda):
    """Known answers captured from the imported reference."""
    from ever_amd.hip import functional as F
    logits = torch.tensor([[.5, -1.], [2., 0.]]).reshape(1, 1, 2, 2).to(cuda)
    labels = torch.tensor([[1, 0], [1, 255]]).reshape(1, 2, 2).to(cuda)
    assert abs(F.bce_with_logits(logits, labels).item() - 0.3047555983) < 1e-6
    assert abs(F.dice_loss_with_logits(logits, labels).item() - 0.1604470611) < 1e-6
    l3 = torch.tensor([[[1., 0.], [0., 2.]], [[0., 1.], [0., 0.]], [[-1., 0.], [3., 0.]]]).reshape(1, 3, 2, 2).to(cuda)
    y3 = torch.tensor([[0, 1], [2, 255]]).reshape(1, 2, 2).to(cuda)
    assert abs(F.dice_loss_with_logits(l3, y3).item() - 0.1912899017) < 1e-6
    assert abs(F.cross_entropy(l3, y3).item() - 0.3513244689) < 1e-6
    assert abs(F.cross_entropy(l3, y3, label_smoothing=0.1).item() - 0.4735466838) < 1e-6


def test_soft_ce_and_label_smoothing_bce(cuda):
    """reference loss.py:222-226 (LS-BCE) and :238-242 (soft CE) incl. the KATs captured from the reference."""
    import json
    import os
    from ever_amd.module import loss as L
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'op_kats.json')) as f:
        k = json.load(f)
    logits = torch.tensor([[.5, -1.], [2., 0.]]).reshape(1, 1, 2, 2).to(cuda)
    labels = torch.tensor([[1, 0], [1, 255]]).reshape(1, 2, 2).to(cuda)
    assert abs(L.label_smoothing_binary_cross_entropy(logits, labels).item() - k['ls_bce']) < 1e-6
    l3 = torch.tensor([[[1., 0.], [0., 2.]], [[0., 1.], [0., 0.]], [[-1., 0.], [3., 0.]]]).reshape(1, 3, 2, 2)
    tgt = torch.softmax(l3.flip(1), dim=1)
    assert abs(L.soft_cross_entropy(l3.to(cuda), tgt.to(cuda)).item() - 1.9859406948) < 1e-6
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 5, 9, 7, generator=g)
    t = torch.softmax(torch.randn(2, 5, 9, 7, generator=g), dim=1)
    xr = x.clone().requires_grad_()
    ref = -(t * TF.log_softmax(xr, dim=1)).mean(dim=(0, 2, 3)).sum()
    (ref * 0.7).backward()
    xg = x.to(cuda).requires_grad_()
    out = L.soft_cross_entropy(xg, t.to(cuda))
    (out * 0.7).backward()
    _close(out, ref, what='soft ce', rtol=1e-5)
    _close(xg.grad, xr.grad, what='d soft ce', rtol=1e-4, atol=1e-9)
    yb = torch.randint(0, 2, (2, 9, 7), generator=g)
    yb[:, :2, :2] = 255
    xb = torch.randn(2, 1, 9, 7, generator=g)
    xr = xb.clone().requires_grad_()
    v = yb.reshape(-1) != 255
    tt = yb.reshape(-1)[v].float()
    tt = torch.where(tt == 0, tt + 0.1, tt - 0.1)
    ref = TF.binary_cross_entropy_with_logits(xr.reshape(-1)[v], tt)
    ref.backward()
    xg = xb.to(cuda).requires_grad_()
    out = L.label_smoothing_binary_cross_entropy(xg, yb.to(cuda))
    out.backward()
    _close(out, ref, what='ls bce', rtol=1e-5)
    _close(xg.grad, xr.grad, what='d ls bce', rtol=1e-4, atol=1e-9)


def test_all_ignored_is_nan(cuda):
    from ever_amd.hip import functional as F
    logits = torch.randn(1, 1, 4, 4).to(cuda)
    labels = torch.full((1, 4, 4), 255).to(cuda)
    assert torch.isnan(F.bce_with_logits(logits, labels)).item()  # mean of an empty selection (loss.py:10-17)


def test_cpu_tensor_fails_loudly():
    from ever_amd.hip import functional as F
    with pytest.raises(F.HipPathError):
        F.conv2d(torch.randn(1, 4, 4, 4), torch.randn(4, 4, 1, 1))


@pytest.mark.parametrize('c,h,w,nchw', [(3, 64, 48, True), (4, 32, 32, True), (3, 34, 62, False), (1, 16, 16, True)])
def test_stem_space_to_depth_conv_matches_torch(cuda, c, h, w, nchw):
    """The 7x7 / stride-2 / padding-3 stem as a space-to-depth 4x4 convolution on the split-MFMA kernels
    (csrc/stem_s2d.hip) vs torch conv2d on the CPU (fp64): forward and weight gradient, 1..4 bands, NCHW and NHWC
    images, sizes that are not multiples of the tiles."""
    from ever_amd.hip import functional as HF
    import ever_amd as er
    if HF.get_conv_math() == 'f32':
        pytest.skip('the space-to-depth stem belongs to the split arithmetics (EVK_CONV_MATH=f32 keeps the fp32 kernel)')
    torch.manual_seed(c * 100 + h)
    conv = er.module.Conv2d(c, 64, 7, 2, 3, bias=False).to(cuda)
    x = torch.randn(2, c, h, w)
    ref = torch.nn.functional.conv2d(x.double(), conv.weight.detach().cpu().double(), None, 2, 3)
    g = torch.randn_like(ref)
    wr = conv.weight.detach().cpu().double().requires_grad_()
    torch.nn.functional.conv2d(x.double(), wr, None, 2, 3).backward(g)
    xg = x.to(cuda)
    if not nchw:
        xg = xg.contiguous(memory_format=torch.channels_last)
    assert HF.stem_conv_applicable(xg, conv)
    y = HF.stem_conv7x7s2(xg, conv.weight)
    y.backward(g.float().to(cuda))
    assert tuple(y.shape) == tuple(ref.shape)
    a, b = y.detach().cpu().double(), ref
    assert float((a - b).abs().max() / b.abs().max()) < 1e-5
    ga, gb = conv.weight.grad.cpu().double(), wr.grad
    assert float((ga - gb).abs().max() / gb.abs().max()) < 1e-5
    assert c == 1 or conv.weight.grad.stride() == conv.weight.stride()


BN_EPI_CASES = [
    # n, h, w, cin, cout, k, stride, pad, bias     (kernel families: x3 single-role, wave-specialised 128/256 wide, LDS halo)
    (2, 32, 32, 64, 64, 1, 1, 0, False), (16, 64, 64, 128, 512, 1, 1, 0, False), (16, 32, 32, 256, 128, 1, 1, 0, True),
    (4, 64, 64, 64, 64, 3, 1, 1, False), (16, 32, 32, 128, 128, 3, 1, 1, False), (16, 64, 64, 256, 256, 3, 1, 1, False),
    (8, 64, 64, 128, 256, 3, 2, 1, False), (3, 17, 13, 32, 64, 1, 1, 0, False), (16, 128, 128, 64, 256, 1, 1, 0, False),
    # the R18 / 64x64 plumbing configuration: a handful of row tiles, one row tile, strided, fewer rows than a tile
    (2, 16, 16, 64, 64, 3, 1, 1, False), (2, 8, 8, 128, 128, 3, 1, 1, False), (2, 16, 16, 64, 128, 3, 2, 1, False),
    (2, 16, 16, 64, 128, 1, 2, 0, False), (2, 6, 6, 64, 64, 3, 1, 1, False),
    # LDS-halo kernel with patches hanging over the edges of the map: the dropped rows must not enter the records
    (16, 77, 43, 64, 64, 3, 1, 1, False), (2, 154, 86, 32, 128, 3, 1, 1, False),
]


@pytest.mark.parametrize('case', BN_EPI_CASES)
def test_batchnorm_statistics_from_the_conv_epilogue(cuda, case):
    """conv -> training-mode BatchNorm with the statistics taken from the convolution's epilogue records (count, mean,
    M2 per row-part, merged with Chan's formula) against the same pair with BatchNorm's own statistics pass
    (EVK_BN_EPILOGUE=0 path): output, saved statistics, running statistics and all gradients; and against torch fp64."""
    import ever_amd as er
    from ever_amd.hip import functional as HF
    n, h, w, cin, cout, k, s, p, bias = case
    torch.manual_seed(cin + cout + k)
    conv = er.module.Conv2d(cin, cout, k, s, p, bias=bias).to(cuda)
    bn_a, bn_b = er.module.BatchNorm2d(cout).to(cuda), er.module.BatchNorm2d(cout).to(cuda)
    with torch.no_grad():
        bn_a.weight.uniform_(0.5, 1.5)
        bn_a.bias.uniform_(-0.3, 0.3)
        bn_b.load_state_dict(bn_a.state_dict())
    x = torch.randn(n, cin, h, w, device=cuda) + 0.7          # non-zero mean: exercises the pivot shift
    g = None
    outs = []
    for fused, bn in ((True, bn_a), (False, bn_b)):
        conv.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_()
        y = conv(xg, bn_stats=fused)
        assert (getattr(y, '_evk_bn_parts', None) is not None) == fused or not fused, 'no epilogue statistics produced'
        if fused:
            assert getattr(y, '_evk_bn_parts', None) is not None, f'{case}: this shape produced no epilogue statistics'
        z = bn(y, relu=True)
        if g is None:
            g = torch.randn_like(z)
        z.backward(g)
        outs.append((z.detach(), xg.grad, conv.weight.grad.clone(), bn.weight.grad, bn.bias.grad,
                     bn.running_mean.clone(), bn.running_var.clone()))
    names = ['out', 'dx', 'dw', 'dgamma', 'dbeta', 'running_mean', 'running_var']
    # the two runs round their statistics differently: an output within an ulp of zero may fall on either side of the ReLU (one
    # of 3.4 M elements of the 77 x 43 case, depending on the reduce passes' block count), and the gradients then differ by that
    # element's share — counted, bounded, and the strict bound kept wherever no mask bit differs
    flips = int(((outs[0][0] > 0) != (outs[1][0] > 0)).sum())
    assert flips <= 3, flips
    for nm, a, b in zip(names, outs[0], outs[1]):
        err = float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))
        assert err < (2e-2 if flips and nm in ('dx', 'dw', 'dgamma', 'dbeta') else 2e-5), (nm, err, flips)
    # and against torch (fp64 on the CPU)
    ref_c = torch.nn.Conv2d(cin, cout, k, s, p, bias=bias).double()
    ref_c.load_state_dict({kk: v.cpu().double() for kk, v in conv.state_dict().items()})
    ref_b = torch.nn.BatchNorm2d(cout).double()
    ref_b.weight.data, ref_b.bias.data = bn_b.weight.detach().cpu().double(), bn_b.bias.detach().cpu().double()
    zr = torch.relu(ref_b(ref_c(x.cpu().double())))
    err = float((outs[0][0].cpu().double() - zr).abs().max() / zr.abs().max())
    assert err < 1e-4, err
    assert float((outs[0][5].cpu().double() - ref_b.running_mean).abs().max()) < 1e-5
    assert float((outs[0][6].cpu().double() - ref_b.running_var).abs().max()) < 1e-5 * float(ref_b.running_var.abs().max()) + 1e-6


@pytest.mark.parametrize('shape', [(2, 64, 32, 48), (1, 64, 30, 34), (3, 32, 16, 16)])
def test_stem_batchnorm_relu_maxpool_fused_equals_separate_passes(cuda, shape):
    """BatchNorm + ReLU + MaxPool2d(3, 2, 1) of the stem as one pass each way (csrc/bn.hip) against the separate passes
    and against torch in float64: pooled map, the gradient of the convolution output, dgamma, dbeta."""
    from ever_amd.hip import functional as F
    import torch.nn.functional as TF
    n, c, h, w = shape
    g = torch.Generator().manual_seed(h * w + c)
    img = torch.randn(n, 3, 2 * h, 2 * w, generator=g).to(cuda)
    wt = (torch.randn(c, 3, 7, 7, generator=g) * 0.1).to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
    gamma = (torch.rand(c, generator=g) + 0.5).to(cuda).requires_grad_()
    beta = (torch.randn(c, generator=g) * 0.3).to(cuda).requires_grad_()
    dp = torch.randn(n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, generator=g).to(cuda)

    def run(fused):
        prev = F._STEM_POOL
        F._STEM_POOL = fused
        try:
            rm, rv = torch.zeros(c, device=cuda), torch.ones(c, device=cuda)
            y = F.stem_conv7x7s2(img, wt, bn_stats=True)
            # (no records from the epilogue at 32 channels: batch_norm_relu_max_pool then runs the separate passes itself)
            assert c < 64 or getattr(y, '_evk_bn_parts', None) is not None
            out = F.batch_norm_relu_max_pool(y, gamma, beta, rm, rv, 0.1, 1e-5)
            grads = torch.autograd.grad(out, [wt, gamma, beta], dp)
            return out.detach(), [t.detach().clone() for t in grads], rm, rv
        finally:
            F._STEM_POOL = prev

    o1, g1, rm1, rv1 = run(True)
    o0, g0, rm0, rv0 = run(False)
    torch.cuda.synchronize()
    assert torch.equal(o1, o0)
    assert torch.equal(rm1, rm0) and torch.equal(rv1, rv0)
    for a, b in zip(g1, g0):
        assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item() + 1e-12
    # float64 reference
    imgd = img.double().cpu()
    wd, gd, bd = (t.detach().double().cpu().requires_grad_() for t in (wt, gamma, beta))
    yd = TF.conv2d(imgd, wd, None, 2, 3)
    od = TF.max_pool2d(TF.relu(TF.batch_norm(yd, None, None, gd, bd, True, 0.1, 1e-5)), 3, 2, 1)
    gr = torch.autograd.grad(od, [wd, gd, bd], dp.double().cpu())
    assert (o1.double().cpu() - od).abs().max().item() < 2e-5 * od.abs().max().item()
    for a, r in zip(g1, gr):
        assert (a.double().cpu() - r).abs().max().item() < 5e-5 * r.abs().max().item() + 1e-9
