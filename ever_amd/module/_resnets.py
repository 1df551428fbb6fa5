"""ResNet v1.5 bodies (reference ever/module/_resnets.py:32-227) on the HIP layers.

Stride sits on the 3x3 conv of a bottleneck; conv init kaiming_normal(fan_out, relu); BN weight 1 /
bias 0.  Attribute names (conv1, bn1, layer1.0.downsample.0 ...) equal torchvision's so pretrained
and reference state dicts load.  Each residual block issues fused kernels:
conv -> [BN+ReLU] -> conv -> [BN+ReLU] -> conv -> [BN + identity add + ReLU].
"""
import torch.nn as nn

from ..hip import functional as HF
from .fold import _takes_epilogue_stats, _use_folded, conv_bn
from .layers import AdaptiveAvgPool2d, BatchNorm2d, Conv2d, GroupNorm, HipSequential, MaxPool2d, ReLU

__all__ = ['ResNet', 'BasicBlock', 'Bottleneck', 'resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152',
           'resnext50_32x4d', 'resnext101_32x4d', 'resnext101_32x8d', 'resnet50_v1c', 'resnet101_v1c']


def conv3x3(cin, cout, stride=1, groups=1, dilation=1):
    return Conv2d(cin, cout, kernel_size=3, stride=stride, padding=dilation, groups=groups, bias=False,
                  dilation=dilation)


def conv1x1(cin, cout, stride=1):
    return Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False)


def _norm(norm_layer):
    """The factory `norm_layer(num_features)` of reference _resnets.py:72-75 / resnet.py:213-225 on the HIP layers:
    BatchNorm2d (default), or any callable that builds an nn.GroupNorm (e.g. functools.partial(nn.GroupNorm, 32)) or a
    BatchNorm2d — instances of the stock classes are retargeted to this package's; anything else has no kernel."""
    if norm_layer is None or norm_layer is nn.BatchNorm2d or norm_layer is BatchNorm2d:
        return BatchNorm2d

    def make(num_features):
        m = norm_layer(num_features)
        if isinstance(m, nn.GroupNorm):
            m.__class__ = GroupNorm
        elif isinstance(m, nn.BatchNorm2d) and not isinstance(m, nn.SyncBatchNorm):
            m.__class__ = BatchNorm2d
        elif not isinstance(m, nn.SyncBatchNorm):
            raise NotImplementedError(f'ever_amd ResNet: norm layer {type(m).__name__} has no HIP kernel (BatchNorm2d, '
                                      f'SyncBatchNorm and GroupNorm are implemented)')
        return m
    return make


def _plain_norm(m):
    """a norm without the BatchNorm-only fusions (statistics from the convolution epilogue, folded inference, ReLU bits):
    the block runs layer by layer — convolution, norm (+ fused ReLU), add, ReLU"""
    return not isinstance(m, BatchNorm2d)


def _norm_act(norm, x, relu):
    return norm(x, relu=relu) if isinstance(norm, GroupNorm) else (HF.relu(norm(x)) if relu else norm(x))


def _inference(bn):
    """eval mode without autograd: the blocks take the folded-convolution path (module/fold.py)"""
    import torch
    return (not bn.training) and not torch.is_grad_enabled()


def _fork(block, x):
    """(conv1(x), shortcut) — the block input feeds both; one autograd node so that the two input
    gradients are summed inside the data-gradient kernel (hip/conv.py:_ConvForkFn)."""
    if block.downsample is None:
        return HF.conv2d_fork(x, block.conv1, bn_stats=(_takes_epilogue_stats(block.bn1), False))
    h, s = HF.conv2d_fork(x, block.conv1, block.downsample[0],
                          bn_stats=(_takes_epilogue_stats(block.bn1), _takes_epilogue_stats(block.downsample[1])))
    return h, block.downsample[1](s)


def _lazy_ok(block, bn):
    """May the block's last BatchNorm hand the shortcut's gradient on UNMASKED with its ReLU bits (hip/functional.py:
    batch_norm_act, lazy_res)?  Only when both possible readers of that gradient are known to understand the bits: the
    block input's fork node (conv1 [+ the shortcut convolution]) and, with a shortcut convolution, ITS BatchNorm — so the
    last BatchNorm and the shortcut's must be this package's own plain BatchNorm2d (a SyncBatchNorm or another norm put
    there by hand would read an unmasked gradient as if it were masked), and no forward hook may sit on the modules in
    between (a hook can hang a tensor hook on the shortcut, which would see the unmasked values).  ADVICE r3."""
    from .layers import BatchNorm2d
    if type(bn) is not BatchNorm2d or bn._forward_hooks:
        return False
    mods = [block.conv1]
    ds = block.downsample
    if ds is not None:
        if len(ds) != 2 or type(ds[1]) is not BatchNorm2d:
            return False
        mods += [ds, ds[0], ds[1]]
    return not any(m._forward_hooks or m._forward_pre_hooks for m in mods)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        norm_layer = _norm(norm_layer)
        if groups != 1 or base_width != 64:
            raise ValueError('BasicBlock only supports groups=1 and base_width=64')
        if dilation > 1:
            raise NotImplementedError('Dilation > 1 not supported in BasicBlock')
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if _plain_norm(self.bn1):
            shortcut = x if self.downsample is None else self.downsample(x)
            out = _norm_act(self.bn1, self.conv1(x), True)
            return HF.relu(HF.add(_norm_act(self.bn2, self.conv2(out), False), shortcut))
        if _inference(self.bn1):
            shortcut = x if self.downsample is None else conv_bn(self.downsample[0], self.downsample[1], x)
            out = conv_bn(self.conv1, self.bn1, x, relu=True)
            return conv_bn(self.conv2, self.bn2, out, residual=shortcut, relu=True)
        h, shortcut = _fork(self, x)
        out = self.bn1(h, relu=True, conv_only=True)       # read by conv2 alone
        # (the shortcut's gradient reaches only the fork node or the down-sampling BatchNorm: lazy_res)
        return conv_bn(self.conv2, self.bn2, out, residual=shortcut, relu=True, lazy_res=_lazy_ok(self, self.bn2))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        norm_layer = _norm(norm_layer)
        width = int(planes * (base_width / 64.)) * groups
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if _plain_norm(self.bn1):
            shortcut = x if self.downsample is None else self.downsample(x)
            out = _norm_act(self.bn1, self.conv1(x), True)
            out = _norm_act(self.bn2, self.conv2(out), True)
            return HF.relu(HF.add(_norm_act(self.bn3, self.conv3(out), False), shortcut))
        if _inference(self.bn1):
            shortcut = x if self.downsample is None else conv_bn(self.downsample[0], self.downsample[1], x)
            out = conv_bn(self.conv1, self.bn1, x, relu=True)
            out = conv_bn(self.conv2, self.bn2, out, relu=True)
            return conv_bn(self.conv3, self.bn3, out, residual=shortcut, relu=True)
        h, shortcut = _fork(self, x)
        out = self.bn1(h, relu=True, conv_only=True)       # read by conv2 alone
        out = conv_bn(self.conv2, self.bn2, out, relu=True, conv_only=True)   # ... and this by conv3 alone
        return conv_bn(self.conv3, self.bn3, out, residual=shortcut, relu=True, lazy_res=_lazy_ok(self, self.bn3))


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, zero_init_residual=False, groups=1, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None, deep_stem=False):
        super().__init__()
        norm_layer = _norm(norm_layer)
        self._norm_layer = norm_layer
        self.inplanes = 64
        self.dilation = 1
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        if len(replace_stride_with_dilation) != 3:
            raise ValueError('replace_stride_with_dilation should be None or a 3-element tuple, got {}'.format(
                replace_stride_with_dilation))
        self.groups = groups
        self.base_width = width_per_group
        self.deep_stem = deep_stem
        if deep_stem:  # v1c: three 3x3 convs instead of the 7x7
            half = self.inplanes // 2
            self.stem = HipSequential(
                Conv2d(3, half, 3, 2, 1, bias=False), BatchNorm2d(half), ReLU(inplace=True),
                Conv2d(half, half, 3, 1, 1, bias=False), BatchNorm2d(half), ReLU(inplace=True),
                Conv2d(half, self.inplanes, 3, 1, 1, bias=False), BatchNorm2d(self.inplanes), ReLU(inplace=True))
        else:
            self.conv1 = Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
            self.bn1 = norm_layer(self.inplanes)
            self.relu = ReLU(inplace=True)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dilate=replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dilate=replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dilate=replace_stride_with_dilation[2])
        self.avgpool = AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        norm_layer = self._norm_layer
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       norm_layer(planes * block.expansion))
        stack = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width, previous_dilation,
                       norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            stack.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width,
                               dilation=self.dilation, norm_layer=norm_layer))
        return nn.Sequential(*stack)

    def stem_forward(self, x):
        if self.deep_stem:
            return self.stem(x)
        if _plain_norm(self.bn1):
            h = HF.stem_conv7x7s2(x, self.conv1.weight) if HF.stem_conv_applicable(x, self.conv1) else self.conv1(x)
            return _norm_act(self.bn1, h, True)
        if HF.stem_conv_applicable(x, self.conv1) and not _use_folded(self.conv1, self.bn1):
            # 7x7 / stride 2 on a 3- or 4-band image: space-to-depth form on the split-MFMA kernels (csrc/stem_s2d.hip)
            return self.bn1(HF.stem_conv7x7s2(x, self.conv1.weight, bn_stats=_takes_epilogue_stats(self.bn1)), relu=True)
        return conv_bn(self.conv1, self.bn1, x, relu=True)

    def stem_pool_forward(self, x):
        """maxpool(stem_forward(x)); BatchNorm + ReLU + max-pool as one pass each way where the 7x7 stem runs in its
        space-to-depth form under a training-mode BatchNorm2d (hip/norm.py:batch_norm_relu_max_pool)."""
        if (not self.deep_stem and type(self.maxpool) is MaxPool2d and _takes_epilogue_stats(self.bn1)
                and HF.stem_conv_applicable(x, self.conv1) and not _use_folded(self.conv1, self.bn1)):
            return self.bn1.forward_relu_pool(HF.stem_conv7x7s2(x, self.conv1.weight, bn_stats=True))
        return self.maxpool(self.stem_forward(x))

    def forward(self, x):
        x = self.stem_pool_forward(x)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x)
        return self.fc(x.reshape(x.size(0), -1))


_URLS = {
    'resnet18': 'https://download.pytorch.org/models/resnet18-5c106cde.pth',
    'resnet34': 'https://download.pytorch.org/models/resnet34-333f7ec4.pth',
    'resnet50': 'https://download.pytorch.org/models/resnet50-19c8e357.pth',
    'resnet101': 'https://download.pytorch.org/models/resnet101-5d3b4d8f.pth',
    'resnet152': 'https://download.pytorch.org/models/resnet152-b121ed2d.pth',
    'resnext50_32x4d': 'https://download.pytorch.org/models/resnext50_32x4d-7cdf4587.pth',
    'resnext101_32x8d': 'https://download.pytorch.org/models/resnext101_32x8d-8ba56ff5.pth',
    'resnext101_32x4d': 'https://s3.ap-northeast-2.amazonaws.com/open-mmlab/pretrain/third_party/resnext101_32x4d-a5af3160.pth',
    'resnet50_v1c': 'https://download.openmmlab.com/pretrain/third_party/resnet50_v1c-2cccc1ad.pth',
    'resnet101_v1c': 'https://download.openmmlab.com/pretrain/third_party/resnet101_v1c-e67eebb6.pth',
}


def _build(arch, block, layers, pretrained, progress, **kwargs):
    model = ResNet(block, layers, **kwargs)
    if pretrained:
        from torch.utils.model_zoo import load_url
        state = load_url(_URLS[arch], progress=progress)
        state = state.get('state_dict', state)
        model.load_state_dict(state, strict=False)
    return model


def resnet18(pretrained=False, progress=True, **kw):
    return _build('resnet18', BasicBlock, [2, 2, 2, 2], pretrained, progress, **kw)


def resnet34(pretrained=False, progress=True, **kw):
    return _build('resnet34', BasicBlock, [3, 4, 6, 3], pretrained, progress, **kw)


def resnet50(pretrained=False, progress=True, **kw):
    return _build('resnet50', Bottleneck, [3, 4, 6, 3], pretrained, progress, **kw)


def resnet101(pretrained=False, progress=True, **kw):
    return _build('resnet101', Bottleneck, [3, 4, 23, 3], pretrained, progress, **kw)


def resnet152(pretrained=False, progress=True, **kw):
    return _build('resnet152', Bottleneck, [3, 8, 36, 3], pretrained, progress, **kw)


# ResNeXt (reference _resnets.py:291-324): the grouped 3x3 convolution runs dense with a block-diagonal weight
# (layers.Conv2d), everything else is the bottleneck above at another width
def resnext50_32x4d(pretrained=False, progress=True, **kw):
    return _build('resnext50_32x4d', Bottleneck, [3, 4, 6, 3], pretrained, progress, **dict(kw, groups=32, width_per_group=4))


def resnext101_32x4d(pretrained=False, progress=True, **kw):
    return _build('resnext101_32x4d', Bottleneck, [3, 4, 23, 3], pretrained, progress, **dict(kw, groups=32, width_per_group=4))


def resnext101_32x8d(pretrained=False, progress=True, **kw):
    return _build('resnext101_32x8d', Bottleneck, [3, 4, 23, 3], pretrained, progress, **dict(kw, groups=32, width_per_group=8))


def resnet50_v1c(pretrained=False, progress=True, **kw):
    return _build('resnet50_v1c', Bottleneck, [3, 4, 6, 3], pretrained, progress, deep_stem=True, **kw)


def resnet101_v1c(pretrained=False, progress=True, **kw):
    return _build('resnet101_v1c', Bottleneck, [3, 4, 23, 3], pretrained, progress, deep_stem=True, **kw)
