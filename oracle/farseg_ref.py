"""TEST INFRASTRUCTURE ONLY — CPU restatement (stock torch.nn, fp32) of the reference's hot path.

Each builder cites the reference lines it follows (paths under the reference tree, ever/...).
Child names / indices are chosen so `state_dict()` keys equal the reference's and ever_amd's,
which lets one deterministic weight set drive all three (oracle/portable.py).
Pinned against the imported reference by oracle/gen_golden.py (bit-equality of logits, losses and
gradients on CPU), then frozen as fixtures under tests/golden/.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------- ResNet (v1.5) -----
# module/_resnets.py:21-29 conv3x3 / conv1x1 (no bias), :32-69 BasicBlock, :72-112 Bottleneck,
# :115-203 ResNet.__init__/_make_layer, :205-212 stem_forward.


def _c3(i, o, s=1, d=1):
    return nn.Conv2d(i, o, 3, s, d, dilation=d, bias=False)


def _c1(i, o, s=1):
    return nn.Conv2d(i, o, 1, s, bias=False)


class _Basic(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1, self.bn1 = _c3(cin, planes, stride), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _c3(planes, planes), nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        # op ORDER matters for bit-equality of accumulated gradients: main branch first, shortcut
        # second, in-place add and ReLU (_resnets.py:52-69)
        out = F.relu(self.bn1(self.conv1(x)), inplace=True)
        out = self.bn2(self.conv2(out))
        idt = x if self.downsample is None else self.downsample(x)
        out += idt
        return F.relu(out, inplace=True)


class _Bottle(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1, self.bn1 = _c1(cin, planes), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _c3(planes, planes, stride, dilation), nn.BatchNorm2d(planes)  # stride on the 3x3
        self.conv3, self.bn3 = _c1(planes, planes * 4), nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)), inplace=True)
        out = F.relu(self.bn2(self.conv2(out)), inplace=True)
        out = self.bn3(self.conv3(out))
        idt = x if self.downsample is None else self.downsample(x)  # _resnets.py:106-107
        out += idt                                                  # _resnets.py:109
        return F.relu(out, inplace=True)                            # _resnets.py:110


_ARCH = {'resnet18': (_Basic, [2, 2, 2, 2]), 'resnet34': (_Basic, [3, 4, 6, 3]),
         'resnet50': (_Bottle, [3, 4, 6, 3]), 'resnet101': (_Bottle, [3, 4, 23, 3])}


class _ResNet(nn.Module):
    def __init__(self, arch, in_channels=3):
        super().__init__()
        block, layers = _ARCH[arch]
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)  # _resnets.py:149 / resnet.py:110-113
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)                           # _resnets.py:153
        self._cin = 64
        self.layer1 = self._stage(block, 64, layers[0], 1)
        self.layer2 = self._stage(block, 128, layers[1], 2)
        self.layer3 = self._stage(block, 256, layers[2], 2)
        self.layer4 = self._stage(block, 512, layers[3], 2)

    def _stage(self, block, planes, n, stride):
        ds = None
        if stride != 1 or self._cin != planes * block.expansion:  # _resnets.py:188-192
            ds = nn.Sequential(_c1(self._cin, planes * block.expansion, stride), nn.BatchNorm2d(planes * block.expansion))
        mods = [block(self._cin, planes, stride, ds)]
        self._cin = planes * block.expansion
        mods += [block(self._cin, planes) for _ in range(1, n)]
        return nn.Sequential(*mods)


class ResNetEncoderRef(nn.Module):
    """module/resnet.py:72-225 with defaults (output_stride 32, no freeze, no checkpointing)."""

    def __init__(self, resnet_type='resnet50', in_channels=3, output_stride=32):
        super().__init__()
        self.resnet = _ResNet(resnet_type, in_channels)
        if output_stride == 16:
            self.resnet.layer4.apply(lambda m: _dilate(m, 2))
        elif output_stride == 8:
            self.resnet.layer3.apply(lambda m: _dilate(m, 2))
            self.resnet.layer4.apply(lambda m: _dilate(m, 4))

    def forward(self, x):
        r = self.resnet
        x = r.maxpool(F.relu(r.bn1(r.conv1(x)), inplace=True))  # resnet.py:185-186
        c2 = r.layer1(x)
        c3 = r.layer2(c2)
        c4 = r.layer3(c3)
        c5 = r.layer4(c4)
        return [c2, c3, c4, c5]


def _dilate(m, dilate):  # module/resnet.py:236-251
    if isinstance(m, nn.Conv2d):
        if m.stride == (2, 2):
            m.stride = (1, 1)
            if m.kernel_size == (3, 3):
                m.dilation = m.padding = (dilate // 2, dilate // 2)
        elif m.kernel_size == (3, 3):
            m.dilation = m.padding = (dilate, dilate)


# --------------------------------------------------------------------------- FPN ---------------
class FPNRef(nn.Module):
    """module/fpn.py:40-115: lateral 1x1 and output 3x3 convs WITHOUT bias/BN/ReLU, each wrapped as
    ConvBlock(conv, Identity, Identity) -> keys `fpn_inner{i}.0.weight`."""

    def __init__(self, in_channels_list, out_channels):
        super().__init__()
        self.n = len(in_channels_list)
        for i, c in enumerate(in_channels_list, 1):
            self.add_module(f'fpn_inner{i}', nn.Sequential(nn.Conv2d(c, out_channels, 1, bias=False), nn.Identity(), nn.Identity()))
            self.add_module(f'fpn_layer{i}', nn.Sequential(nn.Conv2d(out_channels, out_channels, 3, 1, 1, bias=False), nn.Identity(), nn.Identity()))

    def forward(self, xs):
        last = getattr(self, f'fpn_inner{self.n}')(xs[-1])
        outs = [getattr(self, f'fpn_layer{self.n}')(last)]
        for i in range(self.n - 1, 0, -1):
            top = F.interpolate(last, scale_factor=2, mode='nearest')   # fpn.py:100
            last = getattr(self, f'fpn_inner{i}')(xs[i - 1]) + top       # fpn.py:104-105
            outs.insert(0, getattr(self, f'fpn_layer{i}')(last))
        return tuple(outs)


# --------------------------------------------------------------------------- FS-Relation -------
class FSRelationRef(nn.Module):
    """module/fs_relation.py:8-73 (scale_aware_proj=True as FarSegHead defaults, :193)."""

    def __init__(self, scene_embedding_channels, in_channels_list, out_channels, scale_aware_proj=True):
        super().__init__()
        self.scale_aware_proj = scale_aware_proj

        def mlp():
            return nn.Sequential(nn.Conv2d(scene_embedding_channels, out_channels, 1), nn.ReLU(True),
                                 nn.Conv2d(out_channels, out_channels, 1))

        def enc(c):
            return nn.Sequential(nn.Conv2d(c, out_channels, 1), nn.BatchNorm2d(out_channels), nn.ReLU(True))

        self.scene_encoder = nn.ModuleList([mlp() for _ in in_channels_list]) if scale_aware_proj else mlp()
        self.content_encoders = nn.ModuleList([enc(c) for c in in_channels_list])
        self.feature_reencoders = nn.ModuleList([enc(c) for c in in_channels_list])

    def forward(self, scene, feats):
        contents = [e(f) for e, f in zip(self.content_encoders, feats)]
        scenes = [e(scene) for e in self.scene_encoder] if self.scale_aware_proj else [self.scene_encoder(scene)] * len(feats)
        rel = [torch.sigmoid((s * c).sum(dim=1, keepdim=True)) for s, c in zip(scenes, contents)]  # :61-66
        ps = [e(f) for e, f in zip(self.feature_reencoders, feats)]
        return [r * p for r, p in zip(rel, ps)]                                                      # :71


class FSRelationV2Ref(nn.Module):
    """module/fs_relation.py:76-163 (FarSeg++): GroupNorm(32) scene MLP(s), sigmoid relation, re-encoded feature
    weighted by it and concatenated with the pyramid feature, 1x1 (no bias) -> BN -> ReLU -> Dropout2d(0.1)."""

    def __init__(self, scene_embedding_channels, in_channels_list, out_channels, scale_aware_proj=True, dropout=0.1):
        super().__init__()
        self.scale_aware_proj = scale_aware_proj

        def mlp():
            return nn.Sequential(nn.Conv2d(scene_embedding_channels, out_channels, 1), nn.GroupNorm(32, out_channels),
                                 nn.ReLU(True), nn.Conv2d(out_channels, out_channels, 1),
                                 nn.GroupNorm(32, out_channels), nn.ReLU(True))

        def proj():
            return nn.Sequential(nn.Conv2d(out_channels * 2, out_channels, 1, bias=False), nn.BatchNorm2d(out_channels),
                                 nn.ReLU(True), nn.Dropout2d(p=dropout))

        def enc(c):
            return nn.Sequential(nn.Conv2d(c, out_channels, 1), nn.BatchNorm2d(out_channels), nn.ReLU(True))

        # registration order as the reference (:86-120): scene_encoder, project, content_encoders, feature_reencoders
        if scale_aware_proj:
            self.scene_encoder = nn.ModuleList([mlp() for _ in in_channels_list])
            self.project = nn.ModuleList([proj() for _ in in_channels_list])
        else:
            self.scene_encoder = mlp()
            self.project = proj()
        self.content_encoders = nn.ModuleList([enc(c) for c in in_channels_list])
        self.feature_reencoders = nn.ModuleList([enc(c) for c in in_channels_list])

    def forward(self, scene, feats):
        contents = [e(f) for e, f in zip(self.content_encoders, feats)]
        if self.scale_aware_proj:
            scenes = [e(scene) for e in self.scene_encoder]
        else:
            scenes = [self.scene_encoder(scene)] * len(feats)
        rel = [torch.sigmoid((s * c).sum(dim=1, keepdim=True)) for s, c in zip(scenes, contents)]   # :143-151
        ps = [e(f) for e, f in zip(self.feature_reencoders, feats)]
        refined = [torch.cat([r * p, o], dim=1) for r, p, o in zip(rel, ps, feats)]                 # :155
        if self.scale_aware_proj:
            return [op(x) for op, x in zip(self.project, refined)]
        return [self.project(x) for x in refined]


# --------------------------------------------------------------------------- decoder -----------
class _Wrap(nn.Module):  # key-compatible stand-in for ops.Bf16compatible (module/ops.py:152-166), fp32 no-op
    def __init__(self, m):
        super().__init__()
        self._inner_module = m

    def forward(self, x):
        return self._inner_module(x)


class AssymetricDecoderRef(nn.Module):
    """module/fpn.py:144-193."""

    def __init__(self, in_channels, out_channels, in_feat_output_strides=(4, 8, 16, 32), out_feat_output_stride=4,
                 classifier_config=None):
        super().__init__()
        self.blocks = nn.ModuleList()
        for s in in_feat_output_strides:
            n_up = int(math.log2(int(s))) - int(math.log2(int(out_feat_output_stride)))
            n = n_up if n_up != 0 else 1
            self.blocks.append(nn.Sequential(*[nn.Sequential(
                nn.Conv2d(in_channels if i == 0 else out_channels, out_channels, 3, 1, 1, bias=False),
                nn.BatchNorm2d(out_channels), nn.ReLU(True),
                _Wrap(nn.UpsamplingBilinear2d(scale_factor=2)) if n_up != 0 else nn.Identity()) for i in range(n)]))
        cfg = classifier_config or {}
        k = cfg.get('kernel_size', 1)
        sf = cfg.get('scale_factor', 1)
        self.classifier = nn.Sequential(
            nn.Conv2d(out_channels, cfg.get('num_classes', 1), k, padding=(k - 1) // 2),
            _Wrap(nn.UpsamplingBilinear2d(scale_factor=sf)) if sf > 1 else nn.Identity())

    def forward(self, feats):
        inner = [b(f) for b, f in zip(self.blocks, feats)]
        out = sum(inner) / len(inner)   # fpn.py:189
        return self.classifier(out)


class FarSegHeadRef(nn.Module):
    """module/fs_relation.py:166-206 with its default config."""

    def __init__(self, in_channels_list=(256, 512, 1024, 2048), fpn_channels=256, decoder_channels=256,
                 num_classes=1, classifier_kernel=1, relation_version='v1', dropout=0.1):
        super().__init__()
        self.fpn = FPNRef(in_channels_list, fpn_channels)
        if relation_version == 'v2':   # FarSeg++ composition: the same head with FSRelationV2 (:76-163)
            self.fs_relation = FSRelationV2Ref(in_channels_list[-1], (fpn_channels,) * 4, fpn_channels, True, dropout)
        else:
            self.fs_relation = FSRelationRef(in_channels_list[-1], (fpn_channels,) * 4, fpn_channels, True)
        self.fpn_decoder = AssymetricDecoderRef(fpn_channels, decoder_channels, classifier_config=dict(
            scale_factor=4.0, num_classes=num_classes, kernel_size=classifier_kernel))

    def forward(self, feats):
        fpn = self.fpn(feats)
        scene = F.adaptive_avg_pool2d(feats[-1], 1)   # GAP of c5, fs_relation.py:176-177
        return self.fpn_decoder(self.fs_relation(scene, fpn))


# --------------------------------------------------------------------------- losses ------------
def bce_ref(y_pred, y_true, ignore_index=255):
    """module/loss.py:229-235 + _masked_ignore :10-17"""
    yp, yt = y_pred.reshape(-1), y_true.reshape(-1)
    valid = yt != ignore_index
    return F.binary_cross_entropy_with_logits(yp.masked_select(valid).float(), yt.masked_select(valid).float())


def dice_ref(y_pred, y_true, smooth=1.0, ignore_index=255, ignore_channel=-1):
    """module/loss.py:26-37 (select), :40-51 (dice_coeff), :54-75; single process (no all-reduce)."""
    c = y_pred.size(1)
    yp = y_pred.permute(0, 2, 3, 1).reshape(-1, c)
    yt = y_true.reshape(-1)
    valid = yt != ignore_index
    yp, yt = yp[valid, :], yt[valid]
    w = torch.ones(c, dtype=torch.bool)
    if c == 1:
        prob, tgt = yp.sigmoid(), yt.reshape(-1, 1)
    else:
        prob = yp.log_softmax(dim=1).exp()
        tgt = F.one_hot(yt.long(), num_classes=c).type_as(yp)
        if ignore_channel != -1:
            w[ignore_channel] = False
    prob, tgt = prob[:, w], tgt[:, w]
    inter = torch.sum(prob * tgt, dim=0)
    z = prob.sum(dim=0) + tgt.sum(dim=0) + smooth
    return 1. - ((2 * inter + smooth) / z).mean()


def ce_ref(y_pred, y_true, ignore_index=255):
    return F.cross_entropy(y_pred, y_true, ignore_index=ignore_index)


def ls_ce_ref(output, target, eps=0.1, ignore_index=-1):
    """module/loss.py:207-219"""
    c = output.size(1)
    logp = F.log_softmax(output, dim=1)
    loss = -logp.sum(dim=1)
    valid = target.reshape(-1) != ignore_index
    loss = loss.reshape(-1).masked_select(valid).float().mean()
    return loss * eps / c + (1 - eps) * F.nll_loss(logp, target, reduction='mean', ignore_index=ignore_index)


# --------------------------------------------------------------------------- whole model -------
class FarSegRef(nn.Module):
    """encoder + head + loss; state-dict prefixes `en.` / `head.` as ever_amd.module.FarSeg."""

    def __init__(self, resnet_type='resnet50', in_channels=3, num_classes=1, decoder_channels=256, classifier_kernel=1,
                 ignore_index=255, relation_version='v1', dropout=0.1):
        super().__init__()
        self.en = ResNetEncoderRef(resnet_type, in_channels)
        widths = (64, 128, 256, 512) if resnet_type in ('resnet18', 'resnet34') else (256, 512, 1024, 2048)
        self.head = FarSegHeadRef(widths, 256, decoder_channels, num_classes, classifier_kernel, relation_version,
                                  dropout)
        self.num_classes = num_classes
        self.ignore_index = ignore_index

    def logits(self, x):
        return self.head(self.en(x))

    def forward(self, x, y=None):
        lg = self.logits(x)
        if y is None:
            return lg
        return self.loss_from_logits(lg, y)

    def loss_from_logits(self, lg, y):
        if self.num_classes == 1:
            return dict(bce_loss=bce_ref(lg, y, self.ignore_index), dice_loss=dice_ref(lg, y, ignore_index=self.ignore_index))
        return dict(cls_loss=ce_ref(lg, y, self.ignore_index))


def load_portable_weights(model, filled):
    """copy {name: ndarray} (oracle.portable.fill_state_dict) into a torch module, strict."""
    sd = {k: torch.from_numpy(v.copy()) for k, v in filled.items()}
    model.load_state_dict(sd, strict=True)
    return model
