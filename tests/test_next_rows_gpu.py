"""SURVEY §8 f2 / f3 device pieces through the C-ABI against the reference's outputs
(tests/golden/next_kats.{json,npz}): tversky / focal losses with gradients, GPU confusion matrix."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import portable

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
with open(os.path.join(GOLD, 'next_kats.json')) as f:
    KATS = json.load(f)
ARR = np.load(os.path.join(GOLD, 'next_kats.npz'))


def _close(a, b, tol=2e-5):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (np.abs(a - b).max(), np.abs(b).max())


@pytest.mark.parametrize('c', [1, 4])
def test_tversky_matches_reference(cuda, c):
    from ever_amd.module import loss as L
    key = f'next_loss_c{c}'
    y = torch.from_numpy(portable.integers(key + '_y', (2, 12, 10), max(c, 2)).astype(np.int64))
    y[0, :2, :3] = 255
    for alpha, beta, gamma in ((0.3, None, 1.0), (0.7, 0.5, 2.0)):
        z = torch.from_numpy(portable.uniform(key, (2, c, 12, 10), -3.0, 3.0)).to(cuda).requires_grad_()
        v = L.tversky_loss_with_logits(z, y.to(cuda), alpha, beta, gamma, smooth_value=1.0, ignore_index=255,
                                       sync_statistics=False)
        v.backward()
        tag = f'tversky_c{c}_a{alpha}_g{gamma}'
        assert abs(v.item() - KATS[tag]) <= 2e-6 * max(1.0, abs(KATS[tag])), (tag, v.item(), KATS[tag])
        _close(z.grad.cpu().contiguous().numpy(), ARR[tag + '_grad'])


def test_focal_losses_match_reference(cuda):
    from ever_amd.module import loss as L
    zf = portable.uniform('next_focal', (2, 3, 9, 7), -4.0, 4.0)
    yf = (portable.uniform01('next_focal_y', 2 * 3 * 9 * 7) > 0.6).astype(np.float32).reshape(2, 3, 9, 7)
    yt = torch.from_numpy(yf).to(cuda)
    for normalize in (False, True):
        z = torch.from_numpy(zf).to(cuda).requires_grad_()
        v = L.focal_loss(z, yt, gamma=2.0, normalize=normalize)
        v.backward()
        tag = f'focal_norm{int(normalize)}'
        assert abs(v.item() - KATS[tag]) <= 1e-5 * max(1.0, abs(KATS[tag])), (tag, v.item(), KATS[tag])
        _close(z.grad.cpu().numpy().reshape(-1), ARR[tag + '_grad'].reshape(-1))
    for alpha, gamma, red in ((-1.0, 2.0, 'mean'), (0.25, 1.5, 'sum')):
        z = torch.from_numpy(zf).to(cuda).requires_grad_()
        v = L.sigmoid_focal_loss(z, yt, alpha, gamma, red)
        v.backward()
        tag = f'sigmoid_focal_a{alpha}_g{gamma}_{red}'
        assert abs(v.item() - KATS[tag]) <= 1e-5 * max(1.0, abs(KATS[tag])), (tag, v.item(), KATS[tag])
        _close(z.grad.cpu().numpy(), ARR[tag + '_grad'])


@pytest.mark.parametrize('c', [2, 7])
def test_gpu_confusion_matrix_is_exact(cuda, c):
    from ever_amd.metric import PixelMetric
    yt = portable.integers(f'next_cm_t{c}', (3, 40, 33), c).astype(np.int64)
    yp = portable.integers(f'next_cm_p{c}', (3, 40, 33), c).astype(np.int64)
    yp = np.where(portable.uniform01(f'next_cm_m{c}', yt.size).reshape(yt.shape) < 0.6, yt, yp)
    pm = PixelMetric(c)
    b0 = pm.forward(torch.from_numpy(yt[:2]).to(cuda), torch.from_numpy(yp[:2]).to(cuda))
    pm.forward(torch.from_numpy(yt[2:]).to(cuda), torch.from_numpy(yp[2:]).to(cuda))
    assert pm.dense_cm.tolist() == KATS[f'metric_c{c}']['cm']
    assert b0.toarray().sum() == 2 * 40 * 33
    tb = pm.summary_all()
    assert [float(x) for x in tb.iou(list(range(c)))] == pytest.approx(np.round(KATS[f'metric_c{c}']['iou'], 5).tolist(), abs=1e-6)


def test_confusion_from_logits_with_ignore_and_many_classes(cuda):
    """Fused threshold / argmax counting equals the explicit prediction path; labels outside [0, C) (255) are
    skipped; a class count beyond the LDS histogram (C*C > 4096) takes the global-atomic path."""
    from ever_amd.metric import ConfusionMatrix
    g = torch.Generator().manual_seed(5)
    for cl, c in ((1, 2), (5, 5), (70, 70)):
        logits = torch.randn(2, cl, 17, 19, generator=g)
        y = torch.randint(0, c, (2, 17, 19), generator=g)
        y[0, :3] = 255
        pred = (logits[:, 0] > 0).long() if cl == 1 else logits.argmax(1)
        a, b = ConfusionMatrix(c), ConfusionMatrix(c)
        a.forward_logits(y.to(cuda), logits.to(cuda))
        b.forward(y.numpy(), pred.numpy())
        assert a.dense_cm.tolist() == b.dense_cm.tolist()
        assert a.dense_cm.sum() == (y != 255).sum().item()


@pytest.mark.parametrize('n,c,h,w,g,relu', [(2, 256, 1, 1, 32, True), (3, 128, 9, 7, 32, False), (2, 48, 33, 20, 4, True),
                                            (1, 256, 64, 64, 32, True)])
def test_group_norm_matches_torch(cuda, n, c, h, w, g, relu):
    from ever_amd.hip import functional_next as HN
    gen = torch.Generator().manual_seed(n * 1000 + c + h)
    x = torch.randn(n, c, h, w, generator=gen) * 2 + 0.7
    wt, b = torch.randn(c, generator=gen), torch.randn(c, generator=gen)
    gy = torch.randn(n, c, h, w, generator=gen)
    # fp64 reference: with 8-element groups (the 1x1 scene embedding) 1/std amplifies rounding differences
    xr, wr, br = x.double().requires_grad_(), wt.double().requires_grad_(), b.double().requires_grad_()
    yr = torch.nn.functional.group_norm(xr, g, wr, br, 1e-5)
    if relu:
        yr = torch.relu(yr)
    yr.backward(gy.double())
    xg, wg, bg = x.to(cuda).requires_grad_(), wt.to(cuda).requires_grad_(), b.to(cuda).requires_grad_()
    yg = HN.group_norm_act(xg, g, wg, bg, 1e-5, relu=relu)
    yg.backward(gy.to(cuda))
    _close(yg.detach().cpu().contiguous().numpy(), yr.detach().numpy(), 1e-5)
    _close(xg.grad.cpu().contiguous().numpy(), xr.grad.numpy(), 1e-4)
    _close(wg.grad.cpu().numpy(), wr.grad.numpy(), 1e-4)
    _close(bg.grad.cpu().numpy(), br.grad.numpy(), 1e-4)


def test_concat_and_dropout2d(cuda):
    from ever_amd.hip import functional_next as HN
    gen = torch.Generator().manual_seed(3)
    a, b = torch.randn(2, 8, 5, 7, generator=gen), torch.randn(2, 12, 5, 7, generator=gen)
    ag, bg = a.to(cuda).requires_grad_(), b.to(cuda).requires_grad_()
    out = HN.concat_channels(ag, bg)
    assert torch.equal(out.detach().cpu(), torch.cat([a, b], 1))
    gy = torch.randn(2, 20, 5, 7, generator=gen)
    out.backward(gy.to(cuda))
    assert torch.equal(ag.grad.cpu(), gy[:, :8]) and torch.equal(bg.grad.cpu(), gy[:, 8:])
    mask = (torch.rand(2, 8, generator=gen) > 0.3).float()
    xg = a.to(cuda).requires_grad_()
    y = HN.dropout2d(xg, 0.25, True, mask=mask)
    want = a * (mask / 0.75)[:, :, None, None]
    _close(y.detach().cpu().contiguous().numpy(), want.numpy(), 1e-6)
    y.backward(torch.ones_like(y))
    _close(xg.grad.cpu().contiguous().numpy(), ((mask / 0.75)[:, :, None, None]).expand_as(a).numpy(), 1e-6)
    assert HN.dropout2d(xg, 0.25, False) is xg
    # the random draw keeps whole channels and rescales the survivors
    z = HN.dropout2d(torch.ones(4, 64, 3, 3, device=cuda), 0.5, True).cpu()
    per = z.reshape(4, 64, -1)
    assert ((per == 0).all(-1) | (per == 2).all(-1)).all() and 0 < (per[:, :, 0] == 0).float().mean() < 1


@pytest.mark.parametrize('sap', [True, False])
def test_fs_relation_v2_matches_reference(cuda, sap, conv_math):
    from ever_amd.module.fs_relation import FSRelationV2
    gold = np.load(os.path.join(GOLD, f'fsrel_v2_sap{int(sap)}.npz'))
    m = FSRelationV2(128, (64, 64, 64, 64), 64, scale_aware_proj=sap)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in portable.fill_state_dict(m.state_dict()).items()})
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    m = m.to(cuda).train()
    scene = torch.from_numpy(portable.normalish('fsv2/scene', (2, 128, 1, 1))).to(cuda).requires_grad_()
    feats = [torch.from_numpy(portable.normalish(f'fsv2/f{i}', (2, 64, s, s))).to(cuda).requires_grad_()
             for i, s in enumerate((16, 8, 4, 2))]
    outs = m(scene, feats)
    gouts = [torch.from_numpy(portable.normalish(f'fsv2/g{i}', tuple(o.shape))).to(cuda) for i, o in enumerate(outs)]
    torch.autograd.backward(outs, gouts)
    for i, o in enumerate(outs):
        _close(o.detach().cpu().contiguous().numpy(), gold[f'out{i}'], 1e-4)
    _close(scene.grad.cpu().contiguous().numpy(), gold['dscene'], 2e-4)
    for i, f in enumerate(feats):
        _close(f.grad.cpu().contiguous().numpy(), gold[f'dfeat{i}'], 2e-4)
    for k, p_ in m.named_parameters():
        _close(p_.grad.cpu().contiguous().numpy(), gold['grad/' + k], 3e-4)
    sd = m.state_dict()  # flushes the lazily counted num_batches_tracked
    for k, _ in m.named_buffers():
        _close(sd[k].cpu().numpy(), gold['buf/' + k], 1e-5)
    m.eval()
    with torch.no_grad():
        for i, o in enumerate(m(scene, feats)):
            _close(o.cpu().contiguous().numpy(), gold[f'eval_out{i}'], 1e-4)


@pytest.mark.parametrize('relu,res', [(False, False), (True, False), (True, True)])
def test_sync_batchnorm_two_virtual_ranks_equal_full_batch_bn(cuda, relu, res):
    """SyncBatchNorm semantics = plain BatchNorm over the union of the ranks' batches.  Two 'ranks' are the two
    halves of a batch on one GPU; the collectives are replaced by a hook that feeds each half the other half's
    local statistics / sums (exactly what all_gather / all_reduce would deliver), everything else is the
    product path.  Reference: torch BatchNorm2d on the full batch (CPU, fp64)."""
    from ever_amd.module.sync_bn import SyncBatchNorm
    gen = torch.Generator().manual_seed(11)
    c = 32
    x = torch.randn(6, c, 9, 5, generator=gen) * 1.7 + 0.4
    x[:2] += 1.5  # make the two halves statistically different
    r = torch.randn(6, c, 9, 5, generator=gen) if res else None
    gy = torch.randn(6, c, 9, 5, generator=gen)
    wt, bs = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen)
    ref = torch.nn.BatchNorm2d(c).double()
    ref.weight.data, ref.bias.data = wt.double(), bs.double()
    xr = x.double().requires_grad_()
    rr = r.double().requires_grad_() if res else None
    yr = ref(xr)
    if res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    yr.backward(gy.double())

    halves = [slice(0, 2), slice(2, 6)]  # unequal counts on purpose
    mods, xs, rs = [], [], []
    for _ in halves:
        m = SyncBatchNorm(c).to(cuda)
        m.weight.data, m.bias.data = wt.to(cuda), bs.to(cuda)
        mods.append(m.train())
    # pass 1: each rank's local forward statistics (what all_gather would exchange)
    local = {}

    def grab(i):
        def hook(stats, backward=False):
            local[(i, backward)] = stats.clone()
            return stats[None] if not backward else stats
        return hook
    for i, sl in enumerate(halves):
        mods[i]._stats_hook = grab(i)
        xi = x[sl].to(cuda).requires_grad_()
        ri = r[sl].to(cuda).requires_grad_() if res else None
        yi = mods[i](xi, residual=ri, relu=relu)
        yi.backward(gy[sl].to(cuda))   # records this rank's (unsynchronised) backward sums
    fwd_all = torch.stack([local[(0, False)], local[(1, False)]])

    # pass 2: the synchronised run.  Forward hook returns the gathered statistics; the backward sums depend on the
    # synchronised statistics, so they are collected in a first synchronised sweep and summed for the second.
    def synced(i, bwd_total):
        def hook(stats, backward=False):
            if not backward:
                return fwd_all
            local[(i, 'sb')] = stats.clone()
            return bwd_total if bwd_total is not None else stats
        return hook
    for sweep in range(2):
        tot = (local[(0, 'sb')] + local[(1, 'sb')]) if sweep == 1 else None
        outs = []
        for i, sl in enumerate(halves):
            m = SyncBatchNorm(c).to(cuda).train()
            m.weight.data, m.bias.data = wt.to(cuda), bs.to(cuda)
            m._stats_hook = synced(i, tot)
            xi = x[sl].to(cuda).requires_grad_()
            ri = r[sl].to(cuda).requires_grad_() if res else None
            yi = m(xi, residual=ri, relu=relu)
            yi.backward(gy[sl].to(cuda))
            outs.append((m, xi, ri, yi))
    y = torch.cat([o[3].detach().cpu() for o in outs])
    dx = torch.cat([o[1].grad.cpu() for o in outs])
    _close(y.contiguous().numpy(), yr.detach().numpy(), 2e-5)
    _close(dx.contiguous().numpy(), xr.grad.numpy(), 1e-4)
    if res:
        _close(torch.cat([o[2].grad.cpu() for o in outs]).contiguous().numpy(), rr.grad.numpy(), 1e-5)
    dw = sum(o[0].weight.grad.cpu() for o in outs)  # DDP would sum (average) the local parameter gradients
    db = sum(o[0].bias.grad.cpu() for o in outs)
    _close(dw.numpy(), ref.weight.grad.numpy(), 1e-4)
    _close(db.numpy(), ref.bias.grad.numpy(), 1e-4)
    for o in outs:  # running statistics use the merged mean / unbiased variance on every rank
        _close(o[0].running_mean.cpu().numpy(), ref.running_mean.numpy(), 1e-5)
        _close(o[0].running_var.cpu().numpy(), ref.running_var.numpy(), 1e-5)


def test_ohem_matches_reference_kat_and_torch(cuda):
    from ever_amd.module import loss as L
    lo = torch.from_numpy(portable.uniform('next_ohem', (3, 50), 0.0, 2.0)).clone()
    lo[0, :7] = 0.0
    x = lo.to(cuda).requires_grad_()
    v = L.online_hard_example_mining(x, 0.4)
    v.backward()
    assert abs(v.item() - KATS['ohem_0.4']) <= 1e-6 * abs(KATS['ohem_0.4'])
    _close(x.grad.cpu().numpy(), ARR['ohem_0.4_grad'], 1e-6)
    # larger: zeros inside the kept set, negative values, many exact ties at the threshold
    g = torch.Generator().manual_seed(4)
    big = torch.randn(70001, generator=g).abs()
    big[::3] = 0.0
    big[1::7] = 0.5                         # ties
    for ratio in (0.05, 0.5, 0.9):
        k = int(ratio * big.numel())
        top = big.topk(k).values
        want = top[top != 0].mean()
        xb = big.to(cuda).requires_grad_()
        got = L.online_hard_example_mining(xb, ratio)
        got.backward()
        assert abs(got.item() - want.item()) <= 2e-6 * abs(want.item()), (ratio, got.item(), want.item())
        gsum = xb.grad.sum().item()
        assert abs(gsum - 1.0) < 1e-4                                   # the mean's weights sum to one
        assert int((xb.grad != 0).sum()) == int((top != 0).sum())       # exactly the kept non-zero elements
        kept_min = float(big[xb.grad.cpu() != 0].min())
        assert kept_min >= float(top[top != 0].min()) - 1e-7


def test_cross_entropy_per_pixel_with_ohem_pipeline(cuda):
    from ever_amd.module import loss as L
    g = torch.Generator().manual_seed(8)
    z = torch.randn(2, 6, 21, 17, generator=g)
    y = torch.randint(0, 6, (2, 21, 17), generator=g)
    y[1, :4] = 255
    zr = z.clone().requires_grad_()
    ref = torch.nn.functional.cross_entropy(zr, y, ignore_index=255, reduction='none')
    k = int(0.3 * ref.numel())
    top = ref.reshape(-1).topk(k).values
    want = top[top != 0].mean()
    want.backward()
    zg = z.to(cuda).requires_grad_()
    pix = L.cross_entropy_per_pixel(zg, y.to(cuda), ignore_index=255)
    _close(pix.detach().cpu().numpy(), ref.detach().numpy(), 1e-6)
    got = L.online_hard_example_mining(pix, 0.3)
    got.backward()
    assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item())
    _close(zg.grad.cpu().contiguous().numpy(), zr.grad.numpy(), 1e-5)


@pytest.mark.parametrize('c', [1, 5])
def test_dice_statistics_all_reduce_two_virtual_ranks_equal_full_batch(cuda, c):
    """Cross-rank dice (reference loss.py:20-23,46-48: inter and z are SUMMED over the ranks before the ratio, through an
    autograd-aware all-reduce).  Two 'ranks' = two unequal parts of one batch on one GPU; the collective is replaced by
    a hook that adds the other part's statistics / upstream gradients (what all_reduce(SUM) delivers); everything
    else is the product path.  Each rank must report the dice loss of the UNION, and d loss / d logits of its own
    pixels times the world size (the backward all-reduce sums the ranks' upstream gradients; DDP then averages)."""
    from ever_amd.hip import functional as HF
    from oracle import farseg_ref
    gen = torch.Generator().manual_seed(5)
    z = torch.randn(5, c, 12, 10, generator=gen) * 2
    y = torch.randint(0, max(c, 2), (5, 12, 10), generator=gen)
    y[0, :3, :4] = 255
    parts = [slice(0, 2), slice(2, 5)]
    zr = z.double().requires_grad_()
    ref = farseg_ref.dice_ref(zr, y)
    ref.backward()
    local = {}

    def grab(i):
        def hook(t, what):
            local[(i, what)] = t.clone()
            return 1
        return hook
    for i, sl in enumerate(parts):   # pass 1: what each rank would contribute to the forward all-reduce
        HF._rank_sum_hook = grab(i)
        try:
            HF.dice_loss_with_logits(z[sl].to(cuda), y[sl].to(cuda))
        finally:
            HF._rank_sum_hook = None
    total = local[(0, 'dice_stats')] + local[(1, 'dice_stats')]

    def synced(t, what):
        if what == 'dice_stats':
            t.copy_(total)
        else:
            t.mul_(2.0)              # every rank's upstream gradient is the same scalar: SUM over 2 ranks
        return 2
    for i, sl in enumerate(parts):
        zi = z[sl].to(cuda).requires_grad_()
        HF._rank_sum_hook = synced
        try:
            li = HF.dice_loss_with_logits(zi, y[sl].to(cuda))
            li.backward()
        finally:
            HF._rank_sum_hook = None
        assert abs(li.item() - ref.item()) <= 1e-6 * abs(ref.item()) + 1e-7, (i, li.item(), ref.item())
        ga, gb = zi.grad.cpu().contiguous().double().numpy(), 2.0 * zr.grad[sl].numpy()
        assert np.abs(ga - gb).max() <= 1e-4 * np.abs(gb).max(), (np.abs(ga - gb).max(), np.abs(gb).max())


@pytest.mark.parametrize('pw,red', [(None, 'mean'), (2.5, 'mean'), (0.4, 'sum'), (None, 'sum')])
def test_bce_pos_weight_and_reduction_match_reference_formula(cuda, pw, red):
    """reference loss.py:229-235: _masked_ignore then F.binary_cross_entropy_with_logits(reduction, pos_weight)."""
    from ever_amd.module import loss as L
    gen = torch.Generator().manual_seed(9)
    z = (torch.randn(3, 1, 17, 12, generator=gen) * 3).requires_grad_()
    y = (torch.rand(3, 17, 12, generator=gen) < 0.4).long()
    y[1, :5, :4] = 255
    valid = y.reshape(-1) != 255
    ref = torch.nn.functional.binary_cross_entropy_with_logits(
        z.double().reshape(-1)[valid], y.reshape(-1)[valid].double(), reduction=red,
        pos_weight=None if pw is None else torch.tensor(pw, dtype=torch.float64))
    (gr,) = torch.autograd.grad(ref, z)
    zg = z.detach().to(cuda).requires_grad_()
    out = L.binary_cross_entropy_with_logits(zg, y.to(cuda), reduction=red, pos_weight=None if pw is None else torch.tensor(pw))
    out.backward()
    assert abs(out.item() - ref.item()) <= 2e-6 * abs(ref.item())
    _close(zg.grad.cpu().numpy(), gr.numpy(), 1e-5)
    # reduction='none' (round 6): one value per non-ignored pixel, in pixel order (tests/test_api_rows_gpu.py holds it against
    # the imported reference's fixture)
    none = L.binary_cross_entropy_with_logits(zg, y.to(cuda), reduction='none', pos_weight=None if pw is None else torch.tensor(pw))
    refn = torch.nn.functional.binary_cross_entropy_with_logits(
        z.double().reshape(-1)[valid], y.reshape(-1)[valid].double(), reduction='none',
        pos_weight=None if pw is None else torch.tensor(pw, dtype=torch.float64))
    assert none.shape == refn.shape
    _close(none.detach().cpu().numpy(), refn.detach().numpy(), 2e-6)
    with pytest.raises(ValueError):
        L.binary_cross_entropy_with_logits(zg, y.to(cuda), reduction='median')


def test_gelu_and_dropout_kernels(cuda):
    from ever_amd.hip import functional_next as HN
    gen = torch.Generator().manual_seed(10)
    x = torch.randn(2, 8, 7, 5, generator=gen) * 2
    g = torch.randn(2, 8, 7, 5, generator=gen)
    xr = x.double().requires_grad_()
    yr = torch.nn.functional.gelu(xr)
    yr.backward(g.double())
    xg = x.to(cuda).requires_grad_()
    yg = HN.gelu(xg)
    yg.backward(g.to(cuda))
    _close(yg.detach().cpu().numpy(), yr.detach().numpy(), 2e-6)
    _close(xg.grad.cpu().numpy(), xr.grad.numpy(), 2e-6)
    # dropout with a given mask = the formula; with a drawn mask: zeros at rate ~p, survivors scaled by 1/(1-p)
    mask = (torch.rand(2, 8, 7, 5, generator=gen) < 0.7).float()
    xd = x.to(cuda).requires_grad_()
    yd = HN.dropout(xd, 0.3, True, mask=mask)
    yd.backward(g.to(cuda))
    _close(yd.detach().cpu().numpy(), (x * mask / 0.7).numpy(), 1e-6)
    _close(xd.grad.cpu().numpy(), (g * mask / 0.7).numpy(), 1e-6)
    big = torch.ones(4, 16, 64, 64, device=cuda)
    out = HN.dropout(big, 0.25, True)
    kept = float((out != 0).float().mean())
    assert abs(kept - 0.75) < 0.01 and torch.allclose(out[out != 0], torch.tensor(1 / 0.75, device=cuda))
    assert HN.dropout(big, 0.25, False) is big


def test_decoder_with_dropout_groupnorm_gelu_and_no_norm_matches_reference_structure(cuda):
    """reference fpn.py:144-193 options the FarSeg default does not use: `dropout_rate` > 0 (nn.Dropout before the
    classifier), norm_fn != BatchNorm2d (the activation becomes GELU) and norm_fn=None.  Same state-dict keys as the
    stock-torch restatement; with dropout off (eval) the outputs and gradients match it."""
    import functools
    import math
    import torch.nn as nn
    from ever_amd.module.fpn import AssymetricDecoder
    from oracle import portable

    def ref_decoder(norm_fn, c_in=32, c_out=32):
        blocks = nn.ModuleList()
        for s in (4, 8, 16, 32):
            n_up = int(math.log2(s)) - 2
            blocks.append(nn.Sequential(*[nn.Sequential(
                nn.Conv2d(c_in if i == 0 else c_out, c_out, 3, 1, 1, bias=False),
                norm_fn(num_features=c_out) if norm_fn is not None else nn.Identity(),
                nn.ReLU(True) if norm_fn == nn.BatchNorm2d else nn.GELU(),
                nn.UpsamplingBilinear2d(scale_factor=2) if n_up != 0 else nn.Identity()) for i in range(n_up if n_up else 1)]))
        cls = nn.Conv2d(c_out, 4, 1)
        return blocks, cls

    class GN(nn.GroupNorm):
        def __init__(self, num_features):
            super().__init__(8, num_features)

    for norm_fn in (GN, None):
        m = AssymetricDecoder(32, 32, norm_fn=norm_fn,
                              classifier_config=dict(scale_factor=1, num_classes=4, kernel_size=1, dropout_rate=0.2))
        blocks, cls = ref_decoder(norm_fn)
        filled = portable.fill_state_dict(m.state_dict())
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in filled.items()})
        ref_sd = {k.replace('._inner_module', ''): v for k, v in filled.items()}
        blocks.load_state_dict({k[len('blocks.'):]: torch.from_numpy(v.copy()) for k, v in ref_sd.items() if k.startswith('blocks.')})
        cls.load_state_dict({k[len('classifier.0.'):]: torch.from_numpy(v.copy()) for k, v in ref_sd.items() if k.startswith('classifier.0.')})
        feats = [torch.from_numpy(portable.normalish(f'dec/f{i}', (2, 32, s, s))) for i, s in enumerate((16, 8, 4, 2))]
        fr = [f.double().requires_grad_() for f in feats]
        blocks, cls = blocks.double(), cls.double()
        inner = [b(f) for b, f in zip(blocks, fr)]
        out_r = cls(sum(inner) / 4)
        out_r.sum().backward()
        m = m.to(cuda).eval()      # eval: dropout off, the norms here have no running statistics
        fg = [f.to(cuda).requires_grad_() for f in feats]
        out = m(fg)
        out.sum().backward()
        _close(out.detach().cpu().contiguous().numpy(), out_r.detach().numpy(), 2e-4)
        for a, b in zip(fg, fr):
            _close(a.grad.cpu().contiguous().numpy(), b.grad.numpy(), 5e-4)
        m.train()
        o1 = m([f.to(cuda) for f in feats])
        assert torch.isfinite(o1).all()
