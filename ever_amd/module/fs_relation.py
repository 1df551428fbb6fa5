"""Foreground-Scene relation module and FarSegHead (API of reference ever/module/fs_relation.py:8-73,166-206).

scene embedding = GAP(c5) -> per-scale 2-layer 1x1 MLP (bias, ReLU between);
content / re-encode = 1x1 conv (bias) -> BN -> ReLU on each pyramid level;
relation r = sigmoid(<scene, content>_channels);  output = r * re-encoded feature  (one fused kernel).
"""
import torch.nn as nn

from ..core import registry
from ..hip import functional as HF
from ..interface import ERModule
from .fpn import FPN, AssymetricDecoder
from .layers import BatchNorm2d, Conv2d, Dropout2d, GroupNorm, HipSequential, ReLU

__all__ = ['FSRelation', 'FSRelationV2', 'FarSegHead', 'FarSegPPHead']


def _mlp(cin, cout):
    return HipSequential(Conv2d(cin, cout, 1), ReLU(True), Conv2d(cout, cout, 1))


def _conv_bn_relu(cin, cout):
    return HipSequential(Conv2d(cin, cout, 1), BatchNorm2d(cout), ReLU(True))


def _fusable_bn(seq):
    """Conv2d -> BatchNorm2d (training, batch statistics from the convolution's epilogue) -> ReLU, nothing watching"""
    from .fold import _takes_epilogue_stats
    if not (isinstance(seq, nn.Sequential) and len(seq) == 3 and isinstance(seq[0], Conv2d) and isinstance(seq[2], nn.ReLU)):
        return False
    bn = seq[1]
    if not (_takes_epilogue_stats(bn) and bn.training and bn.momentum is not None):
        return False
    return not any(m._forward_hooks or m._forward_pre_hooks for m in (seq, seq[0], bn, seq[2]))


class _NoBranches:
    """stand-in for HF.HeadBranches when the levels run in plain order"""

    def level(self, i):
        return self

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def join(self):
        pass


_PLAIN = _NoBranches()


def _relation_levels(content_encoders, feature_reencoders, scenes, features, branches=None):
    """[relation(scene_i, content_i(f_i), reencode_i(f_i))].  Training, plain BatchNorm: the two 1x1 convolutions of a level
    run as one fork node and BatchNorm + ReLU of both branches run INSIDE the relation kernels (HF.fs_relation_bn:
    the normalised maps are never written, the BatchNorm backward sums come out of the relation backward);
    EVK_RELATION_BN=0, hooks, SyncBatchNorm, eval mode or foreign layer stacks take the layers one by one."""
    import os
    import torch
    fuse = (os.environ.get('EVK_RELATION_BN', '1') != '0' and torch.is_grad_enabled() and not HF.observers_active())
    br = branches if branches is not None else _PLAIN     # (HF.HeadBranches: levels 1.. on the head's branch stream)
    outs = []
    for i, (ce, fr, s, f) in enumerate(zip(content_encoders, feature_reencoders, scenes, features)):
        with br.level(i):
            out = None
            if fuse and f.requires_grad and _fusable_bn(ce) and _fusable_bn(fr):
                zc, zf = HF.conv2d_fork(f, ce[0], fr[0], bn_stats=(True, True))
                out = HF.fs_relation_bn(s, zc, zf, ce[1], fr[1])
                if out is None:     # the convolutions left no statistics records: finish the level layer by layer
                    from .layers import run_sequence
                    out = HF.fs_relation(s, run_sequence(list(ce)[1:], zc), run_sequence(list(fr)[1:], zf))
                else:
                    for bn in (ce[1], fr[1]):
                        if bn.track_running_stats and bn.num_batches_tracked is not None:
                            bn._nbt_pending = getattr(bn, '_nbt_pending', 0) + 1
            if out is None:
                contents, feats = _content_and_reencoded([ce], [fr], [f])
                out = HF.fs_relation(s, contents[0], feats[0])
            outs.append(out)
    return outs


def _content_and_reencoded(content_encoders, feature_reencoders, features):
    """[content_i(f_i)], [reencode_i(f_i)].  Each pyramid level feeds both 1x1 convolutions: under autograd the pair
    runs as one node whose backward sums the two input gradients in the second data-gradient's epilogue (no add
    pass over the 256-channel maps); without autograd (or for foreign layer stacks) the Sequentials run as they are,
    which keeps the folded-BatchNorm inference path."""
    import torch
    from .layers import run_sequence
    contents, feats = [], []
    for ce, fr, f in zip(content_encoders, feature_reencoders, features):
        pairable = (torch.is_grad_enabled() and f.requires_grad and isinstance(ce, nn.Sequential)
                    and isinstance(fr, nn.Sequential) and len(ce) > 0 and len(fr) > 0
                    and isinstance(ce[0], Conv2d) and isinstance(fr[0], Conv2d))
        if pairable:
            from .fold import _takes_epilogue_stats
            want = tuple(len(seq) > 1 and isinstance(seq[1], BatchNorm2d) and _takes_epilogue_stats(seq[1]) for seq in (ce, fr))
            yc, yf = HF.conv2d_fork(f, ce[0], fr[0], bn_stats=want)
            contents.append(run_sequence(list(ce)[1:], yc))
            feats.append(run_sequence(list(fr)[1:], yf))
        else:
            contents.append(ce(f))
            feats.append(fr(f))
    return contents, feats


class FSRelation(nn.Module):
    def __init__(self, scene_embedding_channels, in_channels_list, out_channels, scale_aware_proj=False):
        super().__init__()
        self.scale_aware_proj = scale_aware_proj
        if scale_aware_proj:
            self.scene_encoder = nn.ModuleList([_mlp(scene_embedding_channels, out_channels)
                                                for _ in range(len(in_channels_list))])
        else:
            self.scene_encoder = _mlp(scene_embedding_channels, out_channels)
        self.content_encoders = nn.ModuleList([_conv_bn_relu(c, out_channels) for c in in_channels_list])
        self.feature_reencoders = nn.ModuleList([_conv_bn_relu(c, out_channels) for c in in_channels_list])
        self.normalizer = nn.Sigmoid()  # parameter-free; the sigmoid runs inside the relation kernel

    def forward(self, scene_feature, features, branches=None):
        """branches: a HF.HeadBranches session of the calling head (levels 1.. then run on its branch stream and the CALLER
        joins it in front of the first consumer of all levels); None = plain order on the current stream"""
        if self.scale_aware_proj:
            scenes = [enc(scene_feature) for enc in self.scene_encoder]
        else:
            scenes = [self.scene_encoder(scene_feature)] * len(features)
        return _relation_levels(self.content_encoders, self.feature_reencoders, scenes, features, branches)


class FSRelationV2(nn.Module):
    """FarSeg++ relation module (reference fs_relation.py:76-163): GroupNorm scene MLP, relation-weighted
    re-encoded feature concatenated with the pyramid feature, 1x1 conv-BN-ReLU-Dropout2d projection.
    Same child names as the reference => same state-dict keys."""

    def __init__(self, scene_embedding_channels, in_channels_list, out_channels, scale_aware_proj=False):
        super().__init__()
        self.scale_aware_proj = scale_aware_proj

        def scene_mlp():
            return HipSequential(Conv2d(scene_embedding_channels, out_channels, 1), GroupNorm(32, out_channels),
                                 ReLU(True), Conv2d(out_channels, out_channels, 1), GroupNorm(32, out_channels),
                                 ReLU(True))

        def project():
            return HipSequential(Conv2d(out_channels * 2, out_channels, 1, bias=False), BatchNorm2d(out_channels),
                                 ReLU(True), Dropout2d(p=0.1))

        if scale_aware_proj:
            self.scene_encoder = nn.ModuleList([scene_mlp() for _ in range(len(in_channels_list))])
            self.project = nn.ModuleList([project() for _ in range(len(in_channels_list))])
        else:
            self.scene_encoder = scene_mlp()
            self.project = project()
        self.content_encoders = nn.ModuleList([_conv_bn_relu(c, out_channels) for c in in_channels_list])
        self.feature_reencoders = nn.ModuleList([_conv_bn_relu(c, out_channels) for c in in_channels_list])
        self.normalizer = nn.Sigmoid()

    def forward(self, scene_feature, features, branches=None):
        from ..hip import functional_next as HN
        if self.scale_aware_proj:
            scenes = [enc(scene_feature) for enc in self.scene_encoder]
        else:
            scenes = [self.scene_encoder(scene_feature)] * len(features)
        br = branches if branches is not None else _PLAIN
        related = _relation_levels(self.content_encoders, self.feature_reencoders, scenes, features, branches)
        projects = self.project if self.scale_aware_proj else [self.project] * len(related)
        outs = []
        for i, (rel, o, op) in enumerate(zip(related, features, projects)):
            with br.level(i):
                outs.append(op(HN.concat_channels(rel, o)))
        return outs


_RELATIONS = {'v1': FSRelation, 'v2': FSRelationV2}


@registry.MODEL.register(verbose=False)
class FarSegHead(ERModule):
    def __init__(self, config):
        super().__init__(config)
        self.fpn = FPN(**self.config.fpn)
        # `relation_version`: 'v1' = FSRelation (reference fs_relation.py:8-73, what the reference FarSegHead builds,
        # :174); 'v2' = FSRelationV2 (:76-163), the FarSeg++ relation module, which the reference ships without a head
        # that composes it.  The key lives beside the module's constructor arguments, not inside them.
        version = str(self.config.get('relation_version', 'v1')).lower()
        if version not in _RELATIONS:
            raise ValueError(f"FarSegHead: relation_version must be 'v1' or 'v2', got {version!r}")
        self.fs_relation = _RELATIONS[version](**self.config.fs_relation)
        self.fpn_decoder = AssymetricDecoder(**self.config.fpn_decoder)

    def refined(self, feature_list, branches=None):
        fpn_feats = self.fpn(feature_list)
        scene = HF.global_avg_pool(feature_list[-1])  # GAP of c5 (encoder output), not of P5
        if branches is None:
            return self.fs_relation(scene, fpn_feats)
        return self.fs_relation(scene, fpn_feats, branches=branches)

    def _branches(self, feature_list):
        """the pyramid levels behind the FPN on two streams (HF.HeadBranches), when nothing watches the modules in between"""
        if type(self.fs_relation) not in (FSRelation, FSRelationV2) or type(self.fpn_decoder) is not AssymetricDecoder:
            return None
        if any(m._forward_hooks or m._forward_pre_hooks for m in (self.fs_relation, self.fpn_decoder)):
            return None
        return HF.head_branches(feature_list[-1])

    def forward(self, feature_list):
        br = self._branches(feature_list)
        if br is None:
            return self.fpn_decoder(self.refined(feature_list))
        try:
            return self.fpn_decoder(self.refined(feature_list, br), branches=br)
        finally:
            br.join()       # (the decoder joins in front of its mean; this one only acts when an exception left the fork open)

    def features(self, feature_list):
        """decoder feature map before the classifier (stride out_feat_output_stride)"""
        br = self._branches(feature_list)
        if br is None:
            return self.fpn_decoder.features(self.refined(feature_list))
        try:
            return self.fpn_decoder.features(self.refined(feature_list, br), branches=br)
        finally:
            br.join()

    def set_default_config(self):
        self.config.update(dict(
            relation_version='v1',
            fpn=dict(in_channels_list=(256, 512, 1024, 2048), out_channels=256),
            fs_relation=dict(scene_embedding_channels=2048, in_channels_list=(256, 256, 256, 256), out_channels=256,
                             scale_aware_proj=True),
            fpn_decoder=dict(in_channels=256, out_channels=256, in_feat_output_strides=(4, 8, 16, 32),
                             out_feat_output_stride=4,
                             classifier_config=dict(scale_factor=4.0, num_classes=1, kernel_size=1)),
        ))


@registry.MODEL.register(verbose=False)
class FarSegPPHead(FarSegHead):
    """FarSeg++ head (BASELINE config C3): FPN -> FSRelationV2 -> AssymetricDecoder.  Same state-dict prefixes as
    FarSegHead (`fpn.`, `fs_relation.` incl. `fs_relation.project.*`, `fpn_decoder.`)."""

    def set_default_config(self):
        super().set_default_config()
        self.config.update(dict(relation_version='v2'))
