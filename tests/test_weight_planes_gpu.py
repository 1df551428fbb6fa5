"""Weight-plane cache of the split-arithmetic convolutions (ever_amd/hip/weight_planes.py): one launch per weight
update must give exactly the planes the per-convolution split gives, and no stale plane may ever be read."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _split_arithmetic():
    """the plane cache belongs to the bf16x3 convolutions: run these tests under it whatever EVK_CONV_MATH says"""
    from ever_amd.hip import functional as F
    old = F.get_conv_math()
    F.set_conv_math('bf16x3')
    yield
    F.set_conv_math(old)


def _net(dev, seed=0):
    from ever_amd.module.layers import Conv2d
    torch.manual_seed(seed)
    convs = [Conv2d(16, 64, 3, padding=1, bias=False), Conv2d(64, 64, 1, stride=2, bias=False),
             Conv2d(64, 128, 3, stride=2, padding=1, bias=False), Conv2d(128, 32, 3, padding=1, dilation=1, bias=True)]
    return torch.nn.Sequential(*convs).to(dev)


def _run(net, x):
    net.zero_grad(set_to_none=True)
    xx = x.clone().requires_grad_()
    y = net(xx)
    y.square().mean().backward()
    return [y.detach().clone(), xx.grad.clone()] + [p.grad.clone() for p in net.parameters()]


def _same(a, b):
    return all(torch.equal(u, v) for u, v in zip(a, b))


def test_cached_planes_equal_per_call_split_and_follow_every_kind_of_update(cuda):
    from ever_amd.hip import functional as F, weight_planes as wp
    assert F.get_conv_math() == 'bf16x3'
    net = _net(cuda)
    x = torch.randn(4, 16, 32, 48, device=cuda).contiguous(memory_format=torch.channels_last)
    wp.clear()
    wp._ENABLED = False
    try:
        ref0 = _run(net, x)
    finally:
        wp._ENABLED = True
    s0 = dict(wp.stats)
    got0 = _run(net, x)                      # first sight: every (weight, layout) split alone
    assert _same(ref0, got0)
    assert wp.stats['single'] - s0['single'] == 8      # 4 forward layouts + 4 data gradients (x requires grad)
    got1 = _run(net, x)                      # nothing changed: pure hits, no launch at all
    assert _same(ref0, got1) and wp.stats['single'] - s0['single'] == 8 and wp.stats['multi'] == s0['multi']

    # (1) torch-side in-place update (version counter)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.25)
    got2 = _run(net, x)
    assert wp.stats['multi'] - s0['multi'] == 1 and wp.stats['single'] - s0['single'] == 8   # ONE launch for all
    wp._ENABLED = False
    try:
        ref2 = _run(net, x)
    finally:
        wp._ENABLED = True
    assert _same(ref2, got2) and not torch.equal(ref2[0], ref0[0])

    # (2) the HIP optimiser (raw-pointer writes: epoch)
    from ever_amd.opt.optimizer import FusedSGD
    opt = FusedSGD(net.parameters(), lr=0.5, momentum=0.9)
    _run(net, x)
    opt.step()
    got3 = _run(net, x)
    assert wp.stats['multi'] - s0['multi'] == 2
    wp._ENABLED = False
    try:
        ref3 = _run(net, x)
    finally:
        wp._ENABLED = True
    assert _same(ref3, got3) and not torch.equal(ref3[0], ref2[0])


def test_new_model_at_a_recycled_address_never_sees_old_planes(cuda):
    from ever_amd.hip import weight_planes as wp
    x = torch.randn(2, 16, 16, 16, device=cuda).contiguous(memory_format=torch.channels_last)
    outs = []
    for seed in (1, 2, 3):
        net = _net(cuda, seed)
        got = _run(net, x)
        wp._ENABLED = False
        try:
            ref = _run(net, x)
        finally:
            wp._ENABLED = True
        assert _same(ref, got)
        outs.append(got[0])
        del net
    assert not torch.equal(outs[0], outs[1])


def test_second_geometry_shares_or_adds_layouts(cuda):
    """A different input size reuses the planes when the kernels' layout is the same and adds an entry when it is not
    (3x3 'same' convolutions switch between the halo layout and the generic one with the map size)."""
    from ever_amd.hip import weight_planes as wp
    net = _net(cuda, 5)
    for hw in ((16, 16), (64, 64), (16, 16)):
        x = torch.randn(2, 16, *hw, device=cuda).contiguous(memory_format=torch.channels_last)
        got = _run(net, x)
        wp._ENABLED = False
        try:
            ref = _run(net, x)
        finally:
            wp._ENABLED = True
        assert _same(ref, got), hw
