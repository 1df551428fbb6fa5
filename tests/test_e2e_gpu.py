"""End-to-end parity of ever_amd.module.FarSeg (HIP kernels through the C-ABI) on the MI355X:
  (1) against the golden vectors captured from the imported reference (tests/golden/e2e_*.npz),
  (2) against the oracle restatement run on the box's CPU at a second, larger size.
Bar (BASELINE.json north_star): logits within 1e-3 relative fp32, argmax / threshold masks bit-exact.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import farseg_ref, portable

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _hip_model(meta, dev):
    from ever_amd.module import FarSeg
    widths = (64, 128, 256, 512) if meta['resnet_type'] in ('resnet18', 'resnet34') else (256, 512, 1024, 2048)
    m = FarSeg(dict(
        encoder=dict(resnet_type=meta['resnet_type'], in_channels=meta['in_channels']),
        head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                  fs_relation=dict(scene_embedding_channels=widths[-1], in_channels_list=(256,) * 4, out_channels=256,
                                   scale_aware_proj=True),
                  fpn_decoder=dict(in_channels=256, out_channels=meta['decoder_channels'],
                                   classifier_config=dict(scale_factor=4.0, num_classes=meta['num_classes'],
                                                          kernel_size=meta['classifier_kernel'])))))
    filled = portable.fill_state_dict(m.state_dict())
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in filled.items()}, strict=True)
    return m.to(dev)


def _rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _masks_equal(lg, ref, num_classes):
    if num_classes == 1:
        return np.array_equal(lg > 0, ref > 0)
    return np.array_equal(lg.argmax(1), ref.argmax(1))


@pytest.mark.parametrize('name', ['r18_4band_64', 'r50_3band_64', 'r50_3band_64_c16'])
def test_farseg_matches_reference_golden(cuda, name):
    with open(os.path.join(GOLD, f'e2e_{name}.json')) as f:
        meta = json.load(f)
    gold = np.load(os.path.join(GOLD, f'e2e_{name}.npz'))
    m = _hip_model(meta, cuda)
    x, y = portable.synthetic_batch(name, meta['n'], meta['in_channels'], meta['hw'], meta['hw'], meta['num_classes'])
    x, y = torch.from_numpy(x).to(cuda), torch.from_numpy(y).to(cuda)
    m.train()
    lg = m.head(m.en(x))
    losses = m.loss(lg, y)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    lg_np = lg.detach().cpu().contiguous().numpy()
    assert _rel_err(lg_np, gold['logits']) < 1e-3, f'logits rel err {_rel_err(lg_np, gold["logits"]):.2e}'
    assert _masks_equal(lg_np, gold['logits'], meta['num_classes']), 'prediction masks must be bit-exact'
    for k, v in meta['losses'].items():
        assert abs(losses[k].item() - v) <= 1e-3 * abs(v), (k, losses[k].item(), v)
    bad = []
    for k, p in m.named_parameters():
        ref = meta['grads'][k]
        gn = float(p.grad.double().norm())
        if abs(gn - ref[0]) > 2e-3 * ref[0] + 1e-7:
            bad.append((k, gn, ref[0]))
    assert not bad, f'{len(bad)} gradient norms off: {bad[:5]}'
    # running statistics after one step, then eval-mode logits
    sd = m.state_dict()
    for k, (s, nrm) in meta['running'].items():
        assert abs(float(sd[k].double().norm()) - nrm) <= 1e-3 * nrm + 1e-6, k
    m.eval()
    with torch.no_grad():
        lg_eval = m.head(m.en(x)).cpu().contiguous().numpy()
    assert _rel_err(lg_eval, gold['logits_eval']) < 1e-3
    assert _masks_equal(lg_eval, gold['logits_eval'], meta['num_classes'])


def test_farseg_matches_oracle_larger_tile(cuda):
    """R50, 3x128x128, batch 2: HIP path vs the oracle on this box's CPU, every parameter gradient."""
    meta = dict(resnet_type='resnet50', in_channels=3, num_classes=1, decoder_channels=256, classifier_kernel=1)
    m = _hip_model(meta, cuda)
    ora = farseg_ref.FarSegRef('resnet50', 3, 1)
    farseg_ref.load_portable_weights(ora, portable.fill_state_dict(ora.state_dict()))
    x, y = portable.synthetic_batch('oracle128', 2, 3, 128, 128, 1)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    ora.train()
    lo = ora(xt, yt)
    sum(lo.values()).backward()
    lg_o = ora.logits(xt).detach().numpy()
    m.train()
    out = m(xt.to(cuda), yt.to(cuda))
    sum(out.values()).backward()
    lg = m.head(m.en(xt.to(cuda))).detach().cpu().contiguous().numpy()
    assert _rel_err(lg, lg_o) < 1e-3
    assert np.array_equal(lg > 0, lg_o > 0)
    for k in lo:
        assert abs(out[k].item() - lo[k].item()) <= 1e-3 * abs(lo[k].item())
    worst = 0.0
    for (k, p), (k2, q) in zip(m.named_parameters(), ora.named_parameters()):
        assert k == k2
        e = _rel_err(p.grad.cpu().contiguous().numpy(), q.grad.numpy())
        worst = max(worst, e)
        assert e < 5e-3, f'{k}: grad rel err {e:.2e}'
    print('worst grad rel err', worst)
