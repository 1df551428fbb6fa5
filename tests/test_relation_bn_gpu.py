"""FS-Relation with BatchNorm + ReLU of its two branches inside the relation kernels (hip/pointwise.py:fs_relation_bn,
include/ever_hip.h: evk_relation_bn_*; reference fs_relation.py:39-53,61-71).  Same function as the layer-by-layer path
(EVK_RELATION_BN=0): outputs, input / scene gradients, every parameter gradient and the BatchNorm running statistics agree
to fp32 rounding (the partial sums are formed by other workgroups in another order); the layer-by-layer path is kept where
a hook could observe the tensors that no longer exist, for SyncBatchNorm and in eval mode."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(cuda, v2=False):
    import ever_amd as er
    torch.manual_seed(9)
    cls = er.module.fs_relation.FSRelationV2 if v2 else er.module.fs_relation.FSRelation
    m = cls(128, (64, 64, 64, 64), 64).to(cuda).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5)
            torch.nn.init.uniform_(mod.bias, -0.3, 0.3)
    return m


def _run(m, scene, feats, ws, monkeypatch, fused):
    from ever_amd.hip import functional as HF
    monkeypatch.setenv('EVK_RELATION_BN', '1' if fused else '0')
    for mod in m.modules():
        if hasattr(mod, 'reset_running_stats'):
            mod.reset_running_stats()
            mod._nbt_pending = 0
    m.zero_grad(set_to_none=True)
    torch.manual_seed(21)            # (FSRelationV2 draws Dropout2d masks)
    sc = scene.clone().requires_grad_()
    fs = [f.clone().requires_grad_() for f in feats]
    outs = m(sc, fs)
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    torch.cuda.synchronize()
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return [o.detach() for o in outs], sc.grad, [f.grad for f in fs], {k: p.grad.clone() for k, p in m.named_parameters()}, state


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('v2', [False, True], ids=['FSRelation', 'FSRelationV2'])
def test_fused_relation_is_the_same_function(cuda, monkeypatch, v2, conv_math):
    from ever_amd.hip import functional as HF
    m = _build(cuda, v2)
    g = torch.Generator().manual_seed(3)
    scene = (torch.randn(3, 128, 1, 1, generator=g) * 0.1).to(cuda)      # keeps the sigmoid off saturation
    feats = [(torch.randn(3, 64, 48 >> i, 64 >> i, generator=g) + 0.2).to(cuda).contiguous(memory_format=torch.channels_last)
             for i in range(4)]
    ws = [torch.randn(3, 64, 48 >> i, 64 >> i, generator=g).to(cuda) for i in range(4)]
    o1, s1, f1, p1, st1 = _run(m, scene, feats, ws, monkeypatch, True)
    o0, s0, f0, p0, st0 = _run(m, scene, feats, ws, monkeypatch, False)
    for a, b in zip(o1, o0):
        assert _rel(a, b) < 1e-5
    assert _rel(s1, s0) < 2e-5
    for a, b in zip(f1, f0):
        assert _rel(a, b) < 5e-5
    for k in p0:
        if k.endswith('.0.bias') and ('content_encoders' in k or 'feature_reencoders' in k):
            # bias of a convolution in front of a training-mode BatchNorm: its true gradient is exactly zero (the BatchNorm
            # backward's output sums to zero over the pixels); both paths leave rounding noise there
            wk = k[:-len('bias')] + 'weight'
            assert float(p0[wk].abs().max()) > 0 and float(p1[k].abs().max()) < 1e-3 * float(p0[wk].abs().max()), k
            continue
        assert _rel(p1[k], p0[k]) < 1e-4 or float(p0[k].abs().max()) < 1e-6, (k, _rel(p1[k], p0[k]))
    for k in st0:        # running statistics and num_batches_tracked moved the same way
        if st0[k].dtype.is_floating_point:
            assert _rel(st1[k], st0[k]) < 1e-5, k
        else:
            assert torch.equal(st1[k], st0[k]), k
    assert int(st1['content_encoders.0.1.num_batches_tracked']) == 1


def test_layerwise_path_is_kept_where_the_tensors_could_be_seen(cuda, monkeypatch):
    from ever_amd.module import fs_relation as R
    m = _build(cuda)
    assert R._fusable_bn(m.content_encoders[0]) and R._fusable_bn(m.feature_reencoders[2])
    h = m.content_encoders[0][1].register_forward_hook(lambda mod, i, o: None)
    assert not R._fusable_bn(m.content_encoders[0])
    h.remove()
    m.eval()
    assert not R._fusable_bn(m.content_encoders[0])        # running statistics: the folded / eval path
    m.train()
    from ever_amd.module.sync_bn import SyncBatchNorm
    m.content_encoders[1][1] = SyncBatchNorm(64).to(cuda)
    assert not R._fusable_bn(m.content_encoders[1])
    scene = torch.randn(2, 128, 1, 1, device=cuda)
    feats = [torch.randn(2, 64, 32 >> i, 32 >> i, device=cuda, requires_grad=True) for i in range(4)]
    seen = []
    hk = m.feature_reencoders[3][1].register_forward_hook(lambda mod, i, o: seen.append(tuple(o.shape)))
    outs = m(scene, feats)            # level 1 (SyncBatchNorm, one process) and level 3 (hooked) run layer by layer
    sum(o.sum() for o in outs).backward()
    hk.remove()
    assert seen == [(2, 64, 4, 4)] and all(f.grad is not None for f in feats)


@pytest.mark.parametrize('c,hw', [(64, (13, 9)), (128, (20, 12)), (192, (7, 31)), (320, (8, 8))])
def test_fused_relation_on_ragged_shapes(cuda, monkeypatch, c, hw):
    """channel counts that leave lanes idle or need two chunks per lane, odd pixel counts per image"""
    import ever_amd as er
    torch.manual_seed(c)
    m = er.module.fs_relation.FSRelation(96, (c,), c).to(cuda).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5)
            torch.nn.init.uniform_(mod.bias, -0.3, 0.3)
    scene = (torch.randn(3, 96, 1, 1) * 0.1).to(cuda)
    feats = [(torch.randn(3, c, *hw) + 0.2).to(cuda).contiguous(memory_format=torch.channels_last)]
    ws = [torch.randn(3, c, *hw).to(cuda)]
    o1, s1, f1, p1, st1 = _run(m, scene, feats, ws, monkeypatch, True)
    o0, s0, f0, p0, st0 = _run(m, scene, feats, ws, monkeypatch, False)
    assert _rel(o1[0], o0[0]) < 1e-5 and _rel(s1, s0) < 5e-5 and _rel(f1[0], f0[0]) < 1e-4
    for k in p0:
        if k.endswith('.0.bias') and ('content_encoders' in k or 'feature_reencoders' in k):
            continue
        assert _rel(p1[k], p0[k]) < 2e-4 or float(p0[k].abs().max()) < 1e-6, (k, _rel(p1[k], p0[k]))
