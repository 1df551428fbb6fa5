"""The decoder applies its 1x1 classifier BEFORE the last bilinear x2 of every branch and the mean (module/fpn.py:
conv1x1(mean_i up(r_i)) + b = mean_i up(conv1x1(r_i) + b) — both maps are linear, one per pixel and one per channel).
Same function as the reference order (ever/module/fpn.py:186-193): outputs, input gradients and every parameter gradient
agree to fp32 rounding; anything that could observe the skipped tensors (hooks, active dropout, a 3x3 classifier) keeps the
reference order."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _decoder(cuda, num_classes=1, kernel_size=1, dropout=-1, norm=torch.nn.BatchNorm2d):
    import ever_amd as er
    torch.manual_seed(4)
    return er.module.AssymetricDecoder(64, 64, norm_fn=norm, classifier_config=dict(
        scale_factor=4, num_classes=num_classes, kernel_size=kernel_size, dropout_rate=dropout)).to(cuda).train()


def _run(dec, feats, w, monkeypatch, commute):
    monkeypatch.setenv('EVK_DECODER_COMMUTE', '1' if commute else '0')
    fs = [f.clone().requires_grad_() for f in feats]
    dec.zero_grad(set_to_none=True)
    for m in dec.modules():        # same BatchNorm running statistics going in
        if hasattr(m, 'reset_running_stats'):
            m.reset_running_stats()
    out = dec(fs)
    (out * w).sum().backward()
    torch.cuda.synchronize()
    return out.detach(), [f.grad for f in fs], {k: p.grad.clone() for k, p in dec.named_parameters()}


@pytest.mark.parametrize('num_classes,norm', [(1, torch.nn.BatchNorm2d), (5, torch.nn.BatchNorm2d),
                                              (1, lambda num_features: torch.nn.GroupNorm(32, num_features))])
def test_commuted_classifier_is_the_same_function(cuda, monkeypatch, num_classes, norm):
    dec = _decoder(cuda, num_classes=num_classes, norm=norm)
    g = torch.Generator().manual_seed(2)
    feats = [torch.randn(2, 64, 64 >> i, 64 >> i, generator=g).to(cuda).contiguous(memory_format=torch.channels_last)
             for i in range(4)]
    w = torch.randn(2, num_classes, 256, 256, generator=g).to(cuda)
    assert dec._classifier_commutes()
    o1, g1, p1 = _run(dec, feats, w, monkeypatch, True)
    o0, g0, p0 = _run(dec, feats, w, monkeypatch, False)
    assert o1.shape == o0.shape == (2, num_classes, 256, 256)

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max())
    assert rel(o1, o0) < 2e-6, rel(o1, o0)
    for a, b in zip(g1, g0):
        assert rel(a, b) < 2e-5, rel(a, b)
    for k in p0:
        assert rel(p1[k], p0[k]) < 2e-5, (k, rel(p1[k], p0[k]))


def test_reference_order_is_kept_where_the_difference_could_be_seen(cuda):
    dec = _decoder(cuda)
    assert dec._classifier_commutes()
    h = dec.classifier[0].register_forward_hook(lambda m, i, o: None)       # somebody watches the classifier's input
    assert not dec._classifier_commutes()
    h.remove()
    h = list(dec.blocks[3][-1])[-1].register_forward_hook(lambda m, i, o: None)   # ... or a branch's last upsampling
    assert not dec._classifier_commutes()
    h.remove()
    assert dec._classifier_commutes()
    assert not _decoder(cuda, kernel_size=3)._classifier_commutes()
    drop = _decoder(cuda, dropout=0.5)
    assert not drop._classifier_commutes()         # training-mode dropout acts on the mean
    assert drop.eval()._classifier_commutes()
    assert not _decoder(cuda, num_classes=64)._classifier_commutes()     # nothing to gain: as many classes as channels


@pytest.mark.parametrize('c,k,hw', [(64, 1, (17, 23)), (128, 3, (32, 20)), (192, 2, (9, 11)), (256, 5, (16, 16)), (320, 4, (8, 24))])
def test_bn_relu_classifier_pass_on_ragged_shapes(cuda, c, k, hw):
    """HF.bn_relu_dot (BatchNorm + ReLU + narrow 1x1 convolution as one consumer of a convolution output) against the three
    layers one by one, on channel counts that leave lanes idle, several chunks per lane, odd pixel counts, 1..5 classes"""
    import ever_amd as er
    from ever_amd.hip import functional as HF
    from ever_amd.module.layers import BatchNorm2d, Conv2d
    torch.manual_seed(c + k)
    conv = Conv2d(32, c, 3, 1, 1, bias=False).to(cuda)
    bn = BatchNorm2d(c).to(cuda).train()
    cls = Conv2d(c, k, 1).to(cuda)
    torch.nn.init.uniform_(bn.weight, 0.5, 1.5)
    torch.nn.init.uniform_(bn.bias, -0.3, 0.3)
    x = torch.randn(3, 32, *hw, device=cuda).contiguous(memory_format=torch.channels_last)
    g = torch.randn(3, k, *hw, device=cuda)
    res = []
    for fused in (True, False):
        for m in (conv, bn, cls):
            m.zero_grad(set_to_none=True)
        bn.reset_running_stats()
        xi = x.clone().requires_grad_()
        z = conv(xi, bn_stats=True)
        out = HF.bn_relu_dot(z, bn, cls) if fused else None
        if out is None:
            assert not fused, 'the fused form should take this shape'
            out = cls(bn(z, relu=True))
        (out * g).sum().backward()
        torch.cuda.synchronize()
        res.append((out.detach(), xi.grad, conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                    cls.weight.grad.clone(), cls.bias.grad.clone(), bn.running_mean.clone(), bn.running_var.clone()))

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max())
    names = ('out', 'dx', 'dconv', 'dgamma', 'dbeta', 'dcls_w', 'dcls_b', 'running_mean', 'running_var')
    for name, a, b in zip(names, res[0], res[1]):
        assert rel(a, b) < 5e-5, (name, rel(a, b))
