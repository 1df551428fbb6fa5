"""ChangeStar's ChangeMixin on the HIP kernels vs its stock-torch restatement (oracle/changestar_ref.py,
"parity unpinned": the reference tree holds no definition), and the bitemporal model at configuration C4's tile size
through size-independent properties (date-swap symmetry, agreement of the semantic branch with plain FarSeg)."""
import numpy as np
import pytest
import torch

from oracle import changestar_ref, portable

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_changemixin_matches_torch_restatement(cuda, conv_math):
    from ever_amd.module import ChangeMixin
    m = ChangeMixin(in_channels=128, inner_channels=16, num_convs=3, scale_factor=4.0)
    ora = changestar_ref.ChangeMixinRef(in_channels=128, inner_channels=16, num_convs=3, scale_factor=4.0)
    assert list(m.state_dict().keys()) == list(ora.state_dict().keys())
    filled = portable.fill_state_dict(m.state_dict())
    for mod in (m, ora):
        mod.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in filled.items()}, strict=True)
    m, ora = m.to(cuda).train(), ora.train()
    t1 = torch.from_numpy(portable.normalish('cm/t1', (2, 64, 12, 20)))
    t2 = torch.from_numpy(portable.normalish('cm/t2', (2, 64, 12, 20)))
    g12 = torch.from_numpy(portable.normalish('cm/g12', (2, 1, 48, 80)))
    g21 = torch.from_numpy(portable.normalish('cm/g21', (2, 1, 48, 80)))
    a1, a2 = t1.clone().requires_grad_(), t2.clone().requires_grad_()
    r12, r21 = ora(a1, a2)
    ((r12 * g12).sum() + (r21 * g21).sum()).backward()
    b1, b2 = t1.to(cuda).requires_grad_(), t2.to(cuda).requires_grad_()
    c12, c21 = m(b1, b2)
    ((c12 * g12.to(cuda)).sum() + (c21 * g21.to(cuda)).sum()).backward()
    assert _rel(c12.detach().cpu().contiguous(), r12.detach()) < 1e-4 and _rel(c21.detach().cpu().contiguous(), r21.detach()) < 1e-4
    assert _rel(b1.grad.cpu().contiguous(), a1.grad) < 2e-3 and _rel(b2.grad.cpu().contiguous(), a2.grad) < 2e-3
    for (k, p), (_, q) in zip(m.named_parameters(), ora.named_parameters()):
        g, r = p.grad.cpu().contiguous().double().numpy(), q.grad.double().numpy()
        # a convolution bias in front of a BatchNorm has an analytically ZERO gradient (both sides hold rounding noise
        # of ~1e-5 there): absolute floor
        assert np.abs(g - r).max() <= 2e-3 * np.abs(r).max() + 5e-5, k
    sd, so = m.state_dict(), ora.state_dict()     # (num_batches_tracked is materialised by state_dict())
    for k in so:
        if 'running_' in k or 'num_batches' in k:
            assert _rel(sd[k].double().cpu(), so[k].double()) < 1e-5, k   # BatchNorm statistics over BOTH orders at once


def test_changestar_date_swap_symmetry_and_semantic_branch_at_c4_tile_size(cuda):
    """Eval mode (frozen statistics: every sample independent).  Swapping the dates must swap t1 <-> t2 and the two
    change orders — the change probability is symmetric — and the semantic branch must equal plain FarSeg with the
    same weights on each date.  R50, two 3x512x512 dates, batch 2."""
    import ever_amd as er
    torch.manual_seed(11)
    m = er.module.ChangeStarFarSeg(dict()).to(cuda).eval()
    x = torch.randn(2, 6, 512, 512, device=cuda)
    xs = torch.cat([x[:, 3:], x[:, :3]], dim=1)
    a, b = m(x), m(xs)
    assert torch.allclose(a['t1'], b['t2'], atol=1e-6) and torch.allclose(a['t2'], b['t1'], atol=1e-6)
    assert torch.allclose(a['change'], b['change'], atol=1e-6)
    assert a['change'].shape == (2, 1, 512, 512)
    fs = er.module.FarSeg(dict()).to(cuda).eval()
    fs.en.load_state_dict(m.en.state_dict())
    fs.head.load_state_dict(m.head.state_dict())
    # batch 2 instead of 4: other tile shapes / kernels, i.e. another accumulation order — rounding times this random-init
    # network's conditioning (test_fullsize_gpu.py), inside the 1e-3 contract
    assert torch.allclose(fs(x[:, :3].contiguous()), a['t1'], atol=1e-3)
    assert torch.allclose(fs(x[:, 3:].contiguous()), a['t2'], atol=1e-3)


def test_changestar_training_step_is_finite_and_order_losses_are_symmetric(cuda):
    import ever_amd as er
    torch.manual_seed(12)
    m = er.module.ChangeStarFarSeg(dict(encoder=dict(resnet_type='resnet18'),
                                        head=dict(fpn=dict(in_channels_list=(64, 128, 256, 512)),
                                                  fs_relation=dict(scene_embedding_channels=512)))).to(cuda).train()
    x = torch.randn(4, 6, 128, 128, device=cuda)
    y = dict(cls=(torch.rand(4, 128, 128, device=cuda) < 0.3).long(), cls2=(torch.rand(4, 128, 128, device=cuda) < 0.3).long())
    y['change'] = (y['cls'] != y['cls2']).long()
    y['change'][:, :8, :8] = 255
    losses = m(x, y)
    assert set(losses) == {'t1_bce_loss', 't1_dice_loss', 't2_bce_loss', 't2_dice_loss', 'change12_bce_loss', 'change21_bce_loss'}
    sum(losses.values()).backward()
    assert all(torch.isfinite(v) for v in losses.values())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    # identical dates: the two orders see the same input, so their logits (and losses) coincide
    xi = torch.cat([x[:, :3], x[:, :3]], dim=1)
    m.zero_grad(set_to_none=True)
    li = m(xi, dict(change=y['change']))
    assert torch.allclose(li['change12_bce_loss'], li['change21_bce_loss'], rtol=1e-6)


def test_config_c4_batch8_training_step_and_order_symmetry(cuda):
    """BASELINE configuration C4 at its stated per-GPU size: ChangeStar(FarSeg-R50 + ChangeMixin), bitemporal
    2 x (3 x 512 x 512), BATCH 8, one full training step (forward, six losses, backward, fused SGD).  Properties at
    that size: every loss / gradient / updated parameter is finite; the semantic losses of the two dates swap when the
    dates swap (training-mode BatchNorm sees the same 16 images either way); with identical dates the two change
    orders coincide."""
    import ever_amd as er
    torch.manual_seed(13)
    m = er.module.ChangeStarFarSeg(dict()).to(cuda).train()
    opt = er.opt.FusedSGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator(device='cpu').manual_seed(77)
    x = torch.randn(8, 6, 512, 512, generator=g).to(cuda)
    y = dict(cls=(torch.rand(8, 512, 512, generator=g) < 0.3).long().to(cuda),
             cls2=(torch.rand(8, 512, 512, generator=g) < 0.3).long().to(cuda))
    y['change'] = (y['cls'] != y['cls2']).long()
    y['change'][:, :8, :8] = 255
    losses = m(x, y)
    assert set(losses) == {'t1_bce_loss', 't1_dice_loss', 't2_bce_loss', 't2_dice_loss', 'change12_bce_loss',
                           'change21_bce_loss'}
    sum(losses.values()).backward()
    assert all(torch.isfinite(v) for v in losses.values())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    first = {k: float(v) for k, v in losses.items()}
    m.zero_grad(set_to_none=True)
    # swapped dates (same weights, same 16 images in the BatchNorm batches, other order along the batch)
    xs = torch.cat([x[:, 3:], x[:, :3]], dim=1)
    ys = dict(cls=y['cls2'], cls2=y['cls'], change=y['change'])
    with torch.no_grad():
        swapped = {k: float(v) for k, v in m(xs, ys).items()}
    for a, b in (('t1_bce_loss', 't2_bce_loss'), ('t1_dice_loss', 't2_dice_loss'), ('change12_bce_loss', 'change21_bce_loss')):
        assert abs(first[a] - swapped[b]) <= 1e-3 * abs(first[a]), (a, first[a], swapped[b])
        assert abs(first[b] - swapped[a]) <= 1e-3 * abs(first[b]), (b, first[b], swapped[a])
    # the optimizer step on the first batch's gradients
    losses = m(x, y)
    sum(losses.values()).backward()
    opt.step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in m.parameters())
    xi = torch.cat([x[:, :3], x[:, :3]], dim=1)
    with torch.no_grad():
        li = m(xi, dict(change=y['change']))
    assert torch.allclose(li['change12_bce_loss'], li['change21_bce_loss'], rtol=1e-6)
