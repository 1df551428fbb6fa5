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
    from ever_amd.module import FarSeg, FarSegPP
    widths = (64, 128, 256, 512) if meta['resnet_type'] in ('resnet18', 'resnet34') else (256, 512, 1024, 2048)
    cls = FarSegPP if meta.get('relation_version', 'v1') == 'v2' else FarSeg
    m = cls(dict(
        encoder=dict(resnet_type=meta['resnet_type'], in_channels=meta['in_channels']),
        head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                  fs_relation=dict(scene_embedding_channels=widths[-1], in_channels_list=(256,) * 4, out_channels=256,
                                   scale_aware_proj=True),
                  fpn_decoder=dict(in_channels=256, out_channels=meta['decoder_channels'],
                                   classifier_config=dict(scale_factor=4.0, num_classes=meta['num_classes'],
                                                          kernel_size=meta['classifier_kernel'])))))
    filled = portable.fill_state_dict(m.state_dict())
    if meta.get('classifier_bias') is not None:   # the fixture's bias: no pixel near a decision boundary (gen_golden.py)
        filled['head.fpn_decoder.classifier.0.bias'] = np.asarray(meta['classifier_bias'], dtype=np.float32)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in filled.items()}, strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0                           # as in the golden run (deterministic training-mode forward)
    return m.to(dev)


def _rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _check_masks(lg, ref, num_classes, what, exact=False, pinned=None):
    """Prediction masks (threshold 0 for one logit channel, argmax otherwise).

    exact=True (the committed training-mode goldens): BIT-EXACT, zero flipped pixels.  Those fixtures' classifier
    bias was placed by gen_golden.py in the widest empty interval of the reference's logits, so their smallest
    decision margin (printed; `min_margin_rel` in the fixture) is far above the kernels' rounding differences.  Any
    continuous logit field over 1e4..1e5 pixels has tens of pixels inside +-1e-3 of the range, so 'no pixel inside
    the contract margin' is not attainable by choosing inputs; 'no pixel inside the rounding noise' is, and is pinned.
    exact=False (eval-mode logits, live-oracle sizes — no margin was engineered): identical on every pixel the
    reference decides by more than the contract tolerance (1e-3 of the logit range); flips inside are reported.
    pinned=N: additionally at most N flipped pixels — the count observed on MI355X when the pin was
    set (round 3: 0 on every deterministic fixture under the default arithmetic), a regression value for the pixels INSIDE
    the margin."""
    tol = 1e-3 * np.abs(ref).max()
    if num_classes == 1:
        ma, mb = lg > 0, ref > 0
        margin = np.abs(ref)
    else:
        ma, mb = lg.argmax(1), ref.argmax(1)
        srt = np.sort(ref, axis=1)
        margin = srt[:, -1] - srt[:, -2]
    decided = (margin > tol).reshape(ma.shape)
    assert np.array_equal(ma[decided], mb[decided]), f'{what}: masks differ outside the tie margin'
    flips = int((ma != mb).sum())
    ties = int((~decided).sum())
    if exact:
        assert flips == 0, (f'{what}: {flips} mask pixels differ from the reference (smallest reference margin '
                            f'{margin.min() / np.abs(ref).max():.2e} of the logit range)')
    assert flips <= ties, f'{what}: {flips} mask flips but only {ties} pixels within the tie margin'
    if pinned is not None:
        assert flips <= pinned, f'{what}: {flips} mask flips inside the tie margin, pinned at {pinned}'
    print(f'{what}: masks identical on {int(decided.sum())}/{decided.size} decided pixels; '
          f'{ties} pixels inside the 1e-3 tie margin, {flips} of them flipped; smallest reference margin '
          f'{margin.min() / np.abs(ref).max():.2e}, largest logit difference {np.abs(lg - ref).max() / np.abs(ref).max():.2e} '
          f'of the range')
    return flips


# The 131072-pixel fixture (R50, 2 x 3 x 256 x 256) carries the TIGHT gradient bound: every per-tensor deviation (norm,
# hashed +-1 projection, the 4 stored samples) within TWICE what fp32 rounding alone does to the reference's own
# gradients on that input (its fp32-vs-fp64 digest difference: 1.0e-2 / 5.4e-2 / 9e-3 — even at 256^2 the deepest
# BatchNorms see 8 x 8 maps), plus 2e-3.  The small fixtures use six times their own noise.
TIGHT_FACTOR, TIGHT_ABS = 2.0, 2e-3


@pytest.mark.parametrize('name', ['r18_4band_64', 'r50_3band_64', 'r50_3band_128', 'r50_3band_64_c16', 'r50_3band_256',
                                  'pp_r50_4band_64'])
def test_farseg_matches_reference_golden(cuda, name, conv_math):
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
    # bit-exact wherever the fixture's smallest margin is above the rounding noise (every golden but the 131072-pixel
    # gradient fixture, whose widest empty interval is only 6e-5 of the range)
    # (the 256^2 fixture: 99 of 131072 pixels inside the margin; flips observed: 0 under f16x2 and f32, 2 under bf16x3)
    _check_masks(lg_np, gold['logits'], meta['num_classes'], name, exact=meta['min_margin_rel'] >= 1e-4,
                 pinned={'f16x2': 0, 'bf16x3': 2, 'f32': 0}[conv_math])
    for k, v in meta['losses'].items():
        assert abs(losses[k].item() - v) <= 1e-3 * abs(v), (k, losses[k].item(), v)
    # Gradients.  The backward of a ReLU / max-pool network is DISCONTINUOUS in its activations: a
    # pre-activation within rounding distance of 0 flips its mask bit between two correct fp32
    # evaluations, and on these tiny tiles (4x4 .. 2x2 maps, 8..32-sample BatchNorm statistics) one
    # flip moves a BN gradient by percents.  The reference's OWN fp32 gradients sit up to 1.2e-2
    # (relative, per-tensor norm) from its fp64 gradients on these inputs (gen_golden.py stores both).
    # So this is a 2e-2 bug detector on per-tensor norms; tight element-wise gradient parity is
    # established per kernel in test_ops_gpu.py, and globally on the 256x256 tile below.
    # Which tensor a given decision flip lands in is arbitrary, so the case's conditioning number is the
    # WORST relative fp32-vs-fp64 deviation of the reference over all tensors (r50 @ 64x64: 1.2e-2).
    case_dev = max(abs(meta['grads'][k][0] - v) / v for k, v in meta['grad_norm_fp64'].items() if v > 1e-6)
    bad = []
    for k, p in m.named_parameters():
        ref32, ref64 = meta['grads'][k][0], meta['grad_norm_fp64'][k]
        gn = float(p.grad.double().norm())
        tol = max(2e-2, 6.0 * case_dev) * ref64 + 1e-7
        if abs(gn - ref64) > tol:
            bad.append((k, gn, ref32, ref64))
    assert not bad, f'{len(bad)} gradient norms off: {bad[:5]}'
    # the stored samples / projection of every gradient tensor (gen_golden.py:grad_digest), measured against the
    # tensor's own scale: e = |hip - ref32| / (fp64 norm of the tensor).  A wrong layout, a dropped tap or a missing
    # term shows up as e ~ 1 on the samples and the projection even when the norm happens to agree.  The yardstick is
    # the SAME quantity between the reference's own fp32 and fp64 runs (grads_fp64): what rounding alone does to it.
    worst = dict(norm=0.0, proj=0.0, sample=0.0)
    noise = dict(norm=0.0, proj=0.0, sample=0.0)
    for k, p in m.named_parameters():
        d, d64, ref64 = meta['grads'][k], meta['grads_fp64'][k], meta['grad_norm_fp64'][k]
        if ref64 < 1e-6:
            continue
        g = p.grad.detach().double().reshape(-1).cpu()
        idx = np.linspace(0, g.numel() - 1, 4).astype(np.int64)
        rms8 = 8 * ref64 / np.sqrt(g.numel()) + 1e-12    # a sample is one draw of the element distribution
        dev = dict(norm=abs(float(g.norm()) - d[0]) / ref64,
                   proj=abs(float((g.numpy() * portable.sign_vector(k, g.numel())).sum()) - d[6]) / ref64,
                   sample=max(abs(float(g[i]) - d[2 + j]) for j, i in enumerate(idx)) / rms8)
        own = dict(norm=abs(d[0] - d64[0]) / ref64, proj=abs(d[6] - d64[6]) / ref64,
                   sample=max(abs(d[2 + j] - d64[2 + j]) for j in range(4)) / rms8)
        for kk in worst:
            worst[kk], noise[kk] = max(worst[kk], dev[kk]), max(noise[kk], own[kk])
    print(f'{name}: worst per-tensor gradient deviation from the reference digest '
          + ', '.join(f'{kk} {worst[kk]:.1e} (reference fp32-vs-fp64: {noise[kk]:.1e})' for kk in worst))
    if name == 'r50_3band_256':
        for kk in worst:
            bound = TIGHT_FACTOR * noise[kk] + TIGHT_ABS
            assert worst[kk] <= bound, f'{name}: gradient {kk} deviation {worst[kk]:.2e} > {bound:.2e}'
    for kk in worst:   # every fixture: within a small multiple of what fp32 rounding does to the reference itself.  (Round 4:
        # 4 x noise + 5e-3 instead of max(2e-2, 6 x noise).  The 64^2 / 128^2 fixtures cannot take the 256^2 fixture's 2 x noise:
        # their deepest BatchNorms see 2 x 2 .. 4 x 4 maps, where ONE ReLU decision that falls differently under another
        # summation order moves a tensor's gradient by more than the reference's own fp32-vs-fp64 distance — measured worst
        # case over the three arithmetics: r50_3band_64 norm 4.4e-2 against a noise of 1.2e-2.)
        lim = 4.0 * noise[kk] + 5e-3
        assert worst[kk] <= lim, f'{name}: gradient {kk} deviation {worst[kk]:.2e} > {lim:.2e}'
    # running statistics after one step, then eval-mode logits
    sd = m.state_dict()
    for k, (s, nrm) in meta['running'].items():
        assert abs(float(sd[k].double().norm()) - nrm) <= 1e-3 * nrm + 1e-6, k
    m.eval()
    with torch.no_grad():
        lg_eval = m.head(m.en(x)).cpu().contiguous().numpy()
    assert _rel_err(lg_eval, gold['logits_eval']) < 1e-3
    _check_masks(lg_eval, gold['logits_eval'], meta['num_classes'], name + ' (eval)', pinned=0)   # observed 0, all three


def test_farseg_matches_oracle_larger_tile(cuda, conv_math):
    """R50, 3x256x256, batch 2: HIP path vs the oracle run on this box's CPU in fp32 AND fp64.

    Forward: 1e-3 relative + masks.  Backward: for this random-init network the gradient is badly
    conditioned (ReLU / max-pool decision flips, small-sample BatchNorm): the fp32 ORACLE itself sits
    ~3e-2 (relative L2, per tensor) from the fp64 oracle here.  The HIP gradients are therefore held
    to the same yardstick: per tensor, distance to the fp64 truth at most 3x the fp32 oracle's own
    distance (+2e-3), and the full gradient vector must agree in direction and length."""
    meta = dict(resnet_type='resnet50', in_channels=3, num_classes=1, decoder_channels=256, classifier_kernel=1)
    m = _hip_model(meta, cuda)
    x, y = portable.synthetic_batch('oracle256', 2, 3, 256, 256, 1)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    grads, logits, losses = {}, {}, {}
    for dt in (torch.float32, torch.float64):
        ora = farseg_ref.FarSegRef('resnet50', 3, 1)
        farseg_ref.load_portable_weights(ora, portable.fill_state_dict(ora.state_dict()))
        ora = ora.to(dt).train()
        lg_o = ora.logits(xt.to(dt))
        lo = ora.loss_from_logits(lg_o, yt)
        sum(lo.values()).backward()
        grads[dt] = {k: p.grad.double().numpy() for k, p in ora.named_parameters()}
        logits[dt] = lg_o.detach().double().numpy()
        losses[dt] = {k: float(v) for k, v in lo.items()}
    m.train()
    lg = m.head(m.en(xt.to(cuda)))
    out = m.loss(lg, yt.to(cuda))
    sum(out.values()).backward()
    lg = lg.detach().cpu().contiguous().numpy()
    assert _rel_err(lg, logits[torch.float32]) < 1e-3
    # NATURAL classifier bias (nothing placed in a gap of the reference's logits): 217 of 131072 pixels lie inside the 1e-3 tie
    # margin (smallest reference margin 3.8e-6 of the range).  Flips observed per arithmetic are pinned with a little room for a
    # changed summation order (a new kernel choice moves one or two): masks are identical OUTSIDE the tie band, and inside it
    # this many pixels decide differently from the fp32 CPU reference — not "bit-exact masks" on an un-engineered input.
    _check_masks(lg, logits[torch.float32], 1, f'oracle256[{conv_math}]', pinned={'f16x2': 7, 'bf16x3': 7, 'f32': 4}[conv_math])   # observed 5 / 5 / 2
    e_hip, e_o32 = _rel_err(lg, logits[torch.float64]), _rel_err(logits[torch.float32], logits[torch.float64])
    print(f'logits vs fp64 truth: HIP {e_hip:.2e}, fp32 oracle {e_o32:.2e}')
    for k, v in losses[torch.float32].items():
        assert abs(out[k].item() - v) <= 1e-3 * abs(v)
    worst, dot, na, nb = 0.0, 0.0, 0.0, 0.0
    g32, g64 = grads[torch.float32], grads[torch.float64]
    for k, p in m.named_parameters():
        a = p.grad.cpu().contiguous().numpy().astype(np.float64)
        scale = np.linalg.norm(g64[k])
        e_hip, e_o32 = np.linalg.norm(a - g64[k]), np.linalg.norm(g32[k] - g64[k])
        # conv biases feeding a BatchNorm have an analytically ZERO gradient: absolute floor
        assert e_hip <= 3.0 * e_o32 + 2e-3 * scale + 1e-6, \
            f'{k}: HIP grad is {e_hip / max(scale, 1e-30):.2e} from fp64 truth, fp32 oracle is {e_o32 / max(scale, 1e-30):.2e}'
        if scale > 1e-5:
            worst = max(worst, e_hip / scale)
            dot, na, nb = dot + float((a * g64[k]).sum()), na + float((a * a).sum()), nb + float((g64[k] ** 2).sum())
    cos = dot / np.sqrt(na * nb)
    print(f'worst per-tensor grad rel L2 err vs fp64 {worst:.2e}; global cosine {cos:.6f}; norm ratio {np.sqrt(na / nb):.5f}')
    assert cos > 0.999 and abs(np.sqrt(na / nb) - 1) < 5e-3


def test_folded_batchnorm_inference_matches_unfolded_and_reference(cuda, conv_math):
    """fold_batchnorm: eval-mode logits of the folded model = the unfolded model's (1e-5: only the place of the
    per-channel scaling changes) and stay within the 1e-3 contract of the reference's eval golden."""
    from ever_amd.module.fold import fold_batchnorm
    name = 'r50_3band_128'
    with open(os.path.join(GOLD, f'e2e_{name}.json')) as f:
        meta = json.load(f)
    gold = np.load(os.path.join(GOLD, f'e2e_{name}.npz'))
    m = _hip_model(meta, cuda)
    x, y = portable.synthetic_batch(name, meta['n'], meta['in_channels'], meta['hw'], meta['hw'], meta['num_classes'])
    x, y = torch.from_numpy(x).to(cuda), torch.from_numpy(y).to(cuda)
    m.train()
    sum(m.loss(m.head(m.en(x)), y).values()).backward()     # one step's running statistics, as in the golden
    m.eval()
    with torch.no_grad():
        plain = m.head(m.en(x)).cpu().contiguous().numpy()
    fold_batchnorm(m)
    assert m._folded_pairs >= 53 + 8                         # every ResNet-50 BatchNorm + the head's conv-BN pairs
    with torch.no_grad():
        folded = m.head(m.en(x)).cpu().contiguous().numpy()
    assert _rel_err(folded, plain) < 2e-5, _rel_err(folded, plain)
    assert _rel_err(folded, gold['logits_eval']) < 1e-3
    _check_masks(folded, gold['logits_eval'], meta['num_classes'], name + ' (folded eval)', pinned=0)
    m.train()                                                # training is untouched by the folded copies
    lg = m.head(m.en(x))
    assert _rel_err(lg.detach().cpu().contiguous().numpy(), gold['logits']) < 5e-2  # second step: statistics moved on
    # that training-mode forward moved the running statistics through raw pointers, with no optimiser step
    # ("precise BN" re-calibration): the folded copies must notice and fold again from the new statistics
    m.eval()
    with torch.no_grad():
        refolded = m.head(m.en(x)).cpu().contiguous().numpy()
        from ever_amd.module.fold import unfold_batchnorm
        unfold_batchnorm(m)
        plain2 = m.head(m.en(x)).cpu().contiguous().numpy()
    assert _rel_err(plain2, plain) > 1e-4                     # the statistics did move
    assert _rel_err(refolded, plain2) < 2e-5, _rel_err(refolded, plain2)
    # the plane cache of the folded weights is bounded — a re-derived fold evicts the planes of the copy it
    # replaces, unfold_batchnorm evicts everything it drops (it grew by one entry per folded convolution and re-fold before)
    import gc
    from ever_amd.module import fold as _fold
    gc.collect()                                              # (models of earlier tests: their folded copies die with them)
    mine = lambda: sum(1 for v in _fold._PLANES.values() if v[2].device == x.device and any(
        getattr(c, '_folded', None) is not None and c._folded.weight is v[2] for c in m.modules()))
    assert mine() == 0                                        # unfold_batchnorm above dropped every folded copy
    base = len(_fold._PLANES)
    fold_batchnorm(m)
    with torch.no_grad():
        m.head(m.en(x))
    n0 = len(_fold._PLANES)
    assert n0 - base <= m._folded_pairs and mine() == n0 - base
    assert n0 > base or conv_math == 'f32'                    # (the exact-fp32 kernels read the folded weight itself: no planes)
    for _ in range(3):                                        # train / eval alternation: every eval re-derives every fold
        m.train()
        m.head(m.en(x))
        m.eval()
        with torch.no_grad():
            m.head(m.en(x))
        assert len(_fold._PLANES) == n0, (len(_fold._PLANES), n0)
