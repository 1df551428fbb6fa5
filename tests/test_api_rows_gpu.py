"""Round 6 (VERDICT r5 "missing" 3): API rows that used to raise, each against outputs of the IMPORTED reference
(oracle/gen_golden.py holes -> tests/golden/r6_api.{npz,json}; inputs and weights regenerated from oracle/portable.py):
the FPN's top blocks (reference ever/module/fpn.py:109-141), the 'sum' / 'none' reductions of
label_smoothing_cross_entropy / binary_cross_entropy_with_logits / label_smoothing_binary_cross_entropy (loss.py:207-235) and
ResNetEncoder with a GroupNorm norm_layer (resnet.py:213-225).  Everything runs through the C-ABI (HIP layers)."""
import functools
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def kats():
    return np.load(os.path.join(GOLD, 'r6_api.npz')), json.load(open(os.path.join(GOLD, 'r6_api.json')))


def _rel(a, b):
    b = torch.as_tensor(np.asarray(b)).double()
    return float((a.detach().cpu().double() - b).abs().max() / max(float(b.abs().max()), 1e-30))


@pytest.mark.parametrize('tag', ['maxpool', 'p6p7_c5', 'p6p7_p5'])
def test_fpn_top_blocks_match_the_reference(cuda, kats, tag):
    from oracle import portable
    from ever_amd.module import fpn as F
    arr, _ = kats
    chans, sizes = (16, 32, 64, 128), (32, 16, 8, 4)
    top = {'maxpool': lambda: F.LastLevelMaxPool(), 'p6p7_c5': lambda: F.LastLevelP6P7(128, 32),
           'p6p7_p5': lambda: F.LastLevelP6P7(32, 32)}[tag]()
    m = F.FPN(chans, 32, top_blocks=top)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in portable.fill_state_dict(m.state_dict()).items()}, strict=True)
    m = m.to(cuda).train()
    xs = [torch.from_numpy(portable.normalish(f'fpn_top/x{i}', (2, c, s, s))).to(cuda).contiguous(memory_format=torch.channels_last)
          .requires_grad_() for i, (c, s) in enumerate(zip(chans, sizes))]
    outs = m(xs)
    n_out = 5 if tag == 'maxpool' else 6
    assert len(outs) == n_out
    gouts = [torch.from_numpy(portable.normalish(f'fpn_top/{tag}/g{i}', tuple(o.shape))).to(cuda) for i, o in enumerate(outs)]
    torch.autograd.backward(outs, gouts)
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert tuple(o.shape) == arr[f'fpn_{tag}/out{i}'].shape
        assert _rel(o, arr[f'fpn_{tag}/out{i}']) < 2e-5, (tag, 'out', i)
    for i, x in enumerate(xs):
        assert _rel(x.grad, arr[f'fpn_{tag}/dx{i}']) < 5e-5, (tag, 'dx', i)
    for k, p in m.named_parameters():
        assert _rel(p.grad, arr[f'fpn_{tag}/grad/{k}']) < 5e-5, (tag, k)
    if tag == 'maxpool':    # a one-pixel window at stride 2 is a selection: bit-exact
        assert torch.equal(outs[-1], outs[-2][:, :, ::2, ::2])


def test_fpn_rejects_unknown_top_blocks(cuda):
    from ever_amd.module import fpn as F
    with pytest.raises(TypeError):
        F.FPN((16, 32), 32, top_blocks=torch.nn.Identity())


def test_label_smoothing_cross_entropy_reductions(cuda, kats):
    from oracle import portable
    from ever_amd.module import loss as L
    arr, vals = kats
    z0 = torch.from_numpy(portable.uniform('ls_ce', (2, 5, 12, 10), -3.0, 3.0))
    y = torch.from_numpy(portable.integers('ls_ce_y', (2, 12, 10), 5).astype(np.int64))
    y[1, 3:6, 2:9] = 255
    for red in ('mean', 'sum'):
        z = z0.to(cuda).requires_grad_()
        v = L.label_smoothing_cross_entropy(z, y.to(cuda), eps=0.1, reduction=red, ignore_index=255)
        v.backward()
        assert abs(float(v) - vals[f'ls_ce_{red}']) <= 2e-6 * abs(vals[f'ls_ce_{red}']), red
        assert _rel(z.grad, arr[f'ls_ce_{red}_grad']) < 1e-5, red
    # every pixel ignored: the sum over nothing is 0 (the mean is NaN, as in the reference)
    z = z0.to(cuda).requires_grad_()
    v = L.label_smoothing_cross_entropy(z, torch.full_like(y, 255).to(cuda), eps=0.1, reduction='sum', ignore_index=255)
    assert float(v) == 0.0
    # 'none' where the reference's own expression is defined: flat logits, nothing ignored
    zf = torch.from_numpy(portable.uniform('ls_ce_flat', (40, 6), -3.0, 3.0)).to(cuda).requires_grad_()
    yf = torch.from_numpy(portable.integers('ls_ce_flat_y', (40,), 6).astype(np.int64)).to(cuda)
    v = L.label_smoothing_cross_entropy(zf, yf, eps=0.2, reduction='none', ignore_index=255)
    assert tuple(v.shape) == arr['ls_ce_none'].shape
    v.backward(torch.from_numpy(portable.normalish('ls_ce_flat_g', tuple(v.shape))).to(cuda))
    assert _rel(v, arr['ls_ce_none']) < 2e-6
    assert _rel(zf.grad, arr['ls_ce_none_grad']) < 1e-5


def test_binary_cross_entropy_reduction_none(cuda, kats):
    from oracle import portable
    from ever_amd.module import loss as L
    arr, _ = kats
    z0 = torch.from_numpy(portable.uniform('bce_none', (2, 1, 9, 11), -4.0, 4.0))
    yb = torch.from_numpy((portable.uniform01('bce_none_y', 2 * 9 * 11) > 0.6).astype(np.int64).reshape(2, 9, 11))
    yb[0, :3, 4:] = 255
    yt = yb.reshape(2, 1, 9, 11).float().to(cuda)
    for tag, fn in (('bce_none', lambda z: L.binary_cross_entropy_with_logits(z, yt, 'none', 255)),
                    ('bce_none_pw', lambda z: L.binary_cross_entropy_with_logits(z, yt, 'none', 255, pos_weight=torch.tensor(2.5))),
                    ('lsbce_none', lambda z: L.label_smoothing_binary_cross_entropy(z, yt, 0.1, 'none', 255))):
        z = z0.to(cuda).requires_grad_()
        v = fn(z)
        assert tuple(v.shape) == arr[tag].shape, tag          # one value per non-ignored pixel, in pixel order
        v.backward(torch.from_numpy(portable.normalish(tag + '_g', tuple(v.shape))).to(cuda))
        assert _rel(v, arr[tag]) < 2e-6, tag
        assert _rel(z.grad, arr[tag + '_grad']) < 1e-5, tag


def test_resnet_encoder_with_group_norm(cuda, kats):
    """ResNetEncoder(norm_layer=partial(nn.GroupNorm, 8)) — reference resnet.py:213-225 hands the factory to every block"""
    from oracle import portable
    import ever_amd as er
    arr, vals = kats
    enc = er.module.ResNetEncoder(dict(resnet_type='resnet18', in_channels=4, pretrained=False,
                                       norm_layer=functools.partial(torch.nn.GroupNorm, 8)))
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in portable.fill_state_dict(enc.state_dict()).items()}, strict=True)
    enc = enc.to(cuda).train()
    x = torch.from_numpy(portable.normalish('gn_enc/x', (2, 4, 64, 64))).to(cuda)
    feats = enc(x)
    gouts = [torch.from_numpy(portable.normalish(f'gn_enc/g{i}', tuple(o.shape))).to(cuda) for i, o in enumerate(feats)]
    torch.autograd.backward(feats, gouts)
    torch.cuda.synchronize()
    for i, o in enumerate(feats):
        assert _rel(o, arr[f'gn_enc/out{i}']) < 1e-4, i
    dig = vals['gn_enc_grad_digest']
    worst = 0.0
    for k, p in enc.named_parameters():
        if k not in dig:
            continue
        g = p.grad.detach().double().reshape(-1).cpu()
        ref_norm = dig[k][0]
        if ref_norm < 1e-8:
            continue
        proj = float((g.numpy() * portable.sign_vector(k, g.numel())).sum())
        worst = max(worst, abs(float(g.norm()) - ref_norm) / ref_norm, abs(proj - dig[k][6]) / ref_norm)
    # (no batch statistics and no ReLU-bit handshakes on this path, but 20 layers of ReLU decisions: fp32 rounding moves the
    # gradient norms at the 1e-4 level)
    assert worst < 2e-3, worst
