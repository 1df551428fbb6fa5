"""CPU: the oracle restatement (oracle/farseg_ref.py) against the committed golden vectors that
oracle/gen_golden.py captured from the imported reference.  In the build container this is
bit-exact; elsewhere the CPU BLAS path may differ in the last bits, hence a tight tolerance."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import farseg_ref, portable

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
CASES = ['r18_4band_64', 'r50_3band_64', 'r50_3band_128', 'r50_3band_64_c16', 'pp_r50_4band_64']  # (r50_3band_256: GPU test)


def _load(name):
    with open(os.path.join(GOLD, f'e2e_{name}.json')) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLD, f'e2e_{name}.npz'))


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_golden(name):
    meta, gold = _load(name)
    torch.manual_seed(0)
    m = farseg_ref.FarSegRef(meta['resnet_type'], meta['in_channels'], meta['num_classes'], meta['decoder_channels'],
                             meta['classifier_kernel'], relation_version=meta.get('relation_version', 'v1'), dropout=0.0)
    filled = portable.fill_state_dict(m.state_dict())
    filled['head.fpn_decoder.classifier.0.bias'] = np.asarray(meta['classifier_bias'], dtype=np.float32)
    farseg_ref.load_portable_weights(m, filled)
    x, y = portable.synthetic_batch(name, meta['n'], meta['in_channels'], meta['hw'], meta['hw'], meta['num_classes'])
    x, y = torch.from_numpy(x), torch.from_numpy(y)
    m.train()
    lg = m.logits(x)  # ONE training forward (running statistics move once, as in gen_golden.py)
    losses = m.loss_from_logits(lg, y)
    sum(losses.values()).backward()
    np.testing.assert_allclose(lg.detach().numpy(), gold['logits'], rtol=1e-4, atol=1e-5)
    # the fixture's decision margin (gen_golden.py placed the classifier bias in the widest empty interval): recomputed
    # from the committed logits, and the oracle's masks equal the reference's bit for bit
    rng = np.abs(gold['logits']).max()
    margin = portable.mask_margin(gold['logits'])
    assert abs(margin.min() / rng - meta['min_margin_rel']) <= 1e-6
    if meta['num_classes'] == 1:
        assert np.array_equal(lg.detach().numpy() > 0, gold['logits'] > 0)
    else:
        assert np.array_equal(lg.detach().numpy().argmax(1), gold['logits'].argmax(1))
    for k, v in meta['losses'].items():
        assert abs(losses[k].item() - v) <= 1e-5 * max(1.0, abs(v)), k
    for k, p in m.named_parameters():
        ref = meta['grads'][k]
        assert abs(float(p.grad.double().norm()) - ref[0]) <= 1e-3 * ref[0] + 1e-9, k
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m.logits(x).numpy(), gold['logits_eval'], rtol=1e-4, atol=1e-5)


def test_oracle_losses_match_reference_kats():
    with open(os.path.join(GOLD, 'op_kats.json')) as f:
        k = json.load(f)
    lg = torch.tensor([[.5, -1.], [2., 0.]]).reshape(1, 1, 2, 2)
    yb = torch.tensor([[1, 0], [1, 255]]).reshape(1, 2, 2)
    assert abs(farseg_ref.bce_ref(lg, yb).item() - k['bce']) < 1e-7
    assert abs(farseg_ref.dice_ref(lg, yb).item() - k['dice_binary']) < 1e-7
    l3 = torch.tensor([[[1., 0.], [0., 2.]], [[0., 1.], [0., 0.]], [[-1., 0.], [3., 0.]]]).reshape(1, 3, 2, 2)
    y3 = torch.tensor([[0, 1], [2, 255]]).reshape(1, 2, 2)
    assert abs(farseg_ref.dice_ref(l3, y3).item() - k['dice_3class']) < 1e-7
    assert abs(farseg_ref.dice_ref(l3, y3, ignore_channel=0).item() - k['dice_3class_ignore_ch0']) < 1e-7
    assert abs(farseg_ref.ce_ref(l3, y3).item() - k['ce_3class']) < 1e-7
    assert abs(farseg_ref.ls_ce_ref(l3, y3, ignore_index=255).item() - k['ls_ce_3class']) < 1e-7
    # SURVEY §8 c3 pins, restated literally
    assert abs(k['bce'] - 0.3047555983) < 1e-9 and abs(k['dice_binary'] - 0.1604470611) < 1e-9
    assert abs(k['dice_3class'] - 0.1912899017) < 1e-9 and abs(k['ce_3class'] - 0.3513244689) < 1e-9
    assert abs(k['ls_ce_3class'] - 0.4735466838) < 1e-9


def test_oracle_blocks_match_reference_vectors():
    gold = np.load(os.path.join(GOLD, 'blocks.npz'))
    feats = [torch.from_numpy(portable.normalish(f'blk/f{i}', (2, c, s, s))).requires_grad_()
             for i, (c, s) in enumerate([(64, 16), (128, 8), (256, 4), (512, 2)])]
    fpn = farseg_ref.FPNRef((64, 128, 256, 512), 64)
    rel = farseg_ref.FSRelationRef(512, (64,) * 4, 64, True)
    dec = farseg_ref.AssymetricDecoderRef(64, 32, classifier_config=dict(scale_factor=4.0, num_classes=3, kernel_size=3))
    for m in (fpn, rel, dec):
        farseg_ref.load_portable_weights(m, portable.fill_state_dict(m.state_dict()))
        m.train()
    p = fpn(feats)
    r = rel(torch.nn.functional.adaptive_avg_pool2d(feats[-1], 1), p)
    o = dec(r)
    w = torch.from_numpy(portable.normalish('blk/w', tuple(o.shape)))
    (o * w).sum().backward()
    for i in range(4):
        np.testing.assert_allclose(p[i].detach().numpy(), gold[f'fpn{i}'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(r[i].detach().numpy(), gold[f'rel{i}'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(feats[i].grad.numpy(), gold[f'dfeat{i}'], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(o.detach().numpy(), gold['dec'], rtol=1e-4, atol=1e-5)


def test_portable_generator_is_stable():
    """The hash generator must never change: fixtures depend on it."""
    u = portable.uniform01('en.resnet.conv1.weight', 4)
    assert u.dtype == np.float64
    np.testing.assert_array_equal(np.round(u, 12), np.round(portable.uniform01('en.resnet.conv1.weight', 4), 12))
    assert abs(float(portable.normalish('x', (100000,)).std()) - 1.0) < 0.02
