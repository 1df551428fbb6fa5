"""Grouped convolution and the ResNeXt bodies (reference ever/module/_resnets.py:21-24, :88-112, :291-324; registered in
resnet.py:30-32).  A grouped convolution runs as a dense one with the block-diagonal weight (exact zeros outside the
groups): checked against torch's own grouped convolution in fp64 — forward, input gradient, weight gradient — and a
ResNeXt bottleneck / encoder against the same block built from stock torch.nn layers."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize('cin,cout,groups,k,stride', [(128, 128, 32, 3, 1), (256, 256, 32, 3, 2), (64, 96, 4, 1, 1),
                                                      (32, 32, 32, 3, 1)])
def test_grouped_conv_matches_torch(cuda, cin, cout, groups, k, stride):
    import ever_amd as er
    torch.manual_seed(cin + cout + groups)
    conv = er.module.layers.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=True).to(cuda)
    x = torch.randn(2, cin, 20, 24)
    g = torch.randn(2, cout, (20 + 2 * (k // 2) - k) // stride + 1, (24 + 2 * (k // 2) - k) // stride + 1)
    xr = x.double().requires_grad_()
    wr = conv.weight.detach().cpu().double().requires_grad_()
    br = conv.bias.detach().cpu().double().requires_grad_()
    yr = torch.nn.functional.conv2d(xr, wr, br, stride=stride, padding=k // 2, groups=groups)
    yr.backward(g.double())
    xg = x.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = conv(xg)
    y.backward(g.to(cuda).contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    assert conv.weight.grad.shape == conv.weight.shape == (cout, cin // groups, k, k)
    assert _rel(y.detach(), yr.detach()) < 2e-6
    assert _rel(xg.grad, xr.grad) < 2e-6
    assert _rel(conv.weight.grad, wr.grad) < 5e-6
    assert _rel(conv.bias.grad, br.grad) < 2e-6


def _stock_bottleneck(inplanes, planes, groups, base_width, stride):
    nn = torch.nn
    width = int(planes * (base_width / 64.)) * groups

    class B(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(width)
            self.conv2 = nn.Conv2d(width, width, 3, stride, 1, groups=groups, bias=False)
            self.bn2 = nn.BatchNorm2d(width)
            self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
            self.bn3 = nn.BatchNorm2d(planes * 4)
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))

        def forward(self, x):
            out = torch.relu(self.bn1(self.conv1(x)))
            out = torch.relu(self.bn2(self.conv2(out)))
            return torch.relu(self.bn3(self.conv3(out)) + self.downsample(x))
    return B()


def test_resnext_bottleneck_matches_stock_torch(cuda):
    from ever_amd.module import _resnets
    from ever_amd.module.layers import BatchNorm2d, Conv2d, HipSequential
    torch.manual_seed(1)
    ref = _stock_bottleneck(64, 64, 32, 4, 2).double().train()
    down = HipSequential(Conv2d(64, 256, 1, 2, bias=False), BatchNorm2d(256))
    blk = _resnets.Bottleneck(64, 64, stride=2, downsample=down, groups=32, base_width=4).to(cuda).train()
    for p in ref.parameters():
        torch.nn.init.normal_(p, 0.0, 0.3) if p.dim() > 1 else torch.nn.init.uniform_(p, 0.5, 1.5)
    blk.load_state_dict({k: v.float() for k, v in ref.state_dict().items()}, strict=True)
    assert blk.conv2.weight.shape == (128, 4, 3, 3)
    x = torch.randn(4, 64, 32, 32)
    g = torch.randn(4, 256, 16, 16)
    xr = x.double().requires_grad_()
    yr = ref(xr)
    yr.backward(g.double())
    xg = x.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = blk(xg)
    y.backward(g.to(cuda).contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    assert _rel(y.detach(), yr.detach()) < 2e-5
    assert _rel(xg.grad, xr.grad) < 2e-4
    for (k, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert _rel(p.grad, q.grad) < 5e-4, k
    # inference: the grouped pair is not folded, the others are
    from ever_amd.module.fold import fold_batchnorm
    ref.eval()
    fold_batchnorm(blk)
    with torch.no_grad():
        ye = blk(x.to(cuda).contiguous(memory_format=torch.channels_last))
    assert getattr(blk.conv2, '_folded', None) is None and getattr(blk.conv1, '_folded', None) is not None
    assert _rel(ye, ref(x.double()).detach()) < 2e-5


def test_resnext_encoder_builds_with_reference_keys(cuda):
    import ever_amd as er
    enc = er.module.ResNetEncoder(dict(resnet_type='resnext50_32x4d', in_channels=3)).to(cuda).train()
    sd = enc.state_dict()
    assert sd['resnet.layer1.0.conv2.weight'].shape == (128, 4, 3, 3)      # torchvision's resnext50_32x4d shapes
    assert sd['resnet.layer4.2.conv2.weight'].shape == (1024, 32, 3, 3)
    assert sd['resnet.layer4.2.conv3.weight'].shape == (2048, 1024, 1, 1)
    x = torch.randn(2, 3, 64, 64, device=cuda)
    feats = enc(x)
    assert [tuple(f.shape) for f in feats] == [(2, 256, 16, 16), (2, 512, 8, 8), (2, 1024, 4, 4), (2, 2048, 2, 2)]
    sum(f.sum() for f in feats).backward()
    torch.cuda.synchronize()
    gw = enc.resnet.layer2[0].conv2.weight.grad
    assert gw is not None and gw.shape == (256, 8, 3, 3) and bool(torch.isfinite(gw).all()) and float(gw.abs().max()) > 0
    assert 'resnext101_32x8d' in er.registry.MODEL and 'resnext101_32x4d' in er.registry.MODEL
