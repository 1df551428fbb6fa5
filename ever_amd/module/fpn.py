"""FPN and the asymmetric decoder of FarSeg (API of reference ever/module/fpn.py:40-115,144-193).

FPN convs carry NO bias / BN / ReLU (kaiming_uniform a=1); the top-down path is nearest x2 fused with
the lateral add; decoder branches are [conv3x3 -> BN -> ReLU -> bilinear x2(align_corners)] x n, the
four branches are averaged, the classifier conv has a bias and is followed by bilinear x scale.
Module names / Sequential indices equal the reference's, so state-dict keys are identical
(`fpn_inner1.0.weight`, `blocks.3.2.0.weight`, `classifier.0.bias`, ...).
"""
import math
import os

import torch.nn as nn

from ..hip import functional as HF
from .layers import BatchNorm2d, Conv2d, Dropout, GELU, GroupNorm, HipSequential, ReLU, UpsamplingBilinear2d
from .ops import Bf16compatible, ConvBlock

__all__ = ['FPN', 'AssymetricDecoder', 'LastLevelMaxPool', 'LastLevelP6P7', 'conv_with_kaiming_uniform',
           'default_conv_block', 'conv_bn_block', 'conv_bn_relu_block']


def init_conv(m):
    if isinstance(m, nn.Conv2d):
        nn.init.kaiming_uniform_(m.weight, a=1)


def conv_with_kaiming_uniform(use_bn=False, use_relu=False):
    def make_conv(in_channels, out_channels, kernel_size, stride=1, dilation=1):
        return ConvBlock(in_channels, out_channels, kernel_size, stride,
                         padding=ConvBlock.same_padding(kernel_size, dilation), dilation=dilation, bias=False,
                         bn=use_bn, relu=use_relu, init_fn=init_conv)

    return make_conv


default_conv_block = conv_with_kaiming_uniform(use_bn=False, use_relu=False)
conv_bn_block = conv_with_kaiming_uniform(use_bn=True, use_relu=False)
conv_bn_relu_block = conv_with_kaiming_uniform(use_bn=True, use_relu=True)


class FPN(nn.Module):
    def __init__(self, in_channels_list, out_channels, conv_block=default_conv_block, top_blocks=None):
        super().__init__()
        if top_blocks is not None and not isinstance(top_blocks, (LastLevelMaxPool, LastLevelP6P7)):
            # (the reference's forward ignores any other module, fpn.py:109-114; saying so beats a silent no-op)
            raise TypeError('ever_amd FPN: top_blocks must be a LastLevelMaxPool or a LastLevelP6P7 (ever_amd.module.fpn)')
        self.inner_blocks, self.layer_blocks = [], []
        for idx, cin in enumerate(in_channels_list, 1):
            inner, layer = f'fpn_inner{idx}', f'fpn_layer{idx}'
            if cin == 0:
                continue
            self.add_module(inner, conv_block(cin, out_channels, 1))
            self.add_module(layer, conv_block(out_channels, out_channels, 3, 1))
            self.inner_blocks.append(inner)
            self.layer_blocks.append(layer)
        self.top_blocks = top_blocks

    def forward(self, x):
        """x: feature maps, highest resolution first -> tuple of FPN maps, highest resolution first."""
        last_inner = getattr(self, self.inner_blocks[-1])(x[-1])
        out, top = self._output_conv(self.layer_blocks[-1], last_inner, len(self.inner_blocks) > 1)
        results = [out]
        names = list(zip(x[:-1][::-1], self.inner_blocks[:-1][::-1], self.layer_blocks[:-1][::-1]))
        for i, (feat, inner, layer) in enumerate(names):
            lateral = getattr(self, inner)(feat)
            last_inner = HF.upsample_nearest2x_add(top, lateral)  # lateral + nearest_x2(top), one pass
            out, top = self._output_conv(layer, last_inner, i + 1 < len(names))
            results.insert(0, out)
        # reference fpn.py:109-114: extra, coarser levels appended behind the last (smallest-resolution) FPN output
        if isinstance(self.top_blocks, LastLevelP6P7):
            results.extend(self.top_blocks(x[-1], results[-1]))
        elif isinstance(self.top_blocks, LastLevelMaxPool):
            results.extend(self.top_blocks(results[-1]))
        return tuple(results)

    def _output_conv(self, layer, last_inner, has_finer_level):
        """(P_k, the tensor to hand to the next finer level).  last_inner_k feeds its 3x3 output convolution AND, later in the
        forward, the top-down path of level k - 1; backward runs that later consumer first, so its gradient is parked in a
        GradSlot and added inside the output convolution's data-gradient epilogue (hip/conv.py: GradSlot) instead of by
        autograd's own add pass over a 256-channel map — which also left a tensor without an operand scale behind, i.e. a
        stand-alone absmax pass in front of the two gradient kernels that read it."""
        block = getattr(self, layer)
        conv = block[0] if isinstance(block, nn.Sequential) and len(block) == 3 else None
        plain = (conv is not None and type(conv) is Conv2d and isinstance(block[1], nn.Identity) and isinstance(block[2], nn.Identity)
                 and conv.groups == 1 and not (block._forward_hooks or block._forward_pre_hooks or conv._forward_hooks
                                               or conv._forward_pre_hooks))
        if not (has_finer_level and plain and HF.grad_slots_enabled() and last_inner.requires_grad):
            return block(last_inner), last_inner
        slot = HF.GradSlot()
        out = HF.conv2d(last_inner, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, grad_slot=slot)
        return out, (HF.slot_output(last_inner, slot) if slot.claimed else last_inner)


class LastLevelMaxPool(nn.Module):
    """reference fpn.py:118-120: `F.max_pool2d(x, 1, 2, 0)` of the coarsest FPN map — a one-pixel window at stride 2"""

    def forward(self, x):
        return [HF.max_pool1x1s2(x)]


class LastLevelP6P7(nn.Module):
    """reference fpn.py:123-141 (RetinaNet's P6, P7): two 3x3 / stride-2 convolutions with bias, a ReLU between them, fed by
    P5 when the channel counts agree and by C5 otherwise; same parameter names (`p6.weight`, `p7.bias`, ...)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.p6 = Conv2d(in_channels, out_channels, 3, 2, 1)
        self.p7 = Conv2d(out_channels, out_channels, 3, 2, 1)
        for module in (self.p6, self.p7):
            nn.init.kaiming_uniform_(module.weight, a=1)
            nn.init.constant_(module.bias, 0)
        self.use_P5 = in_channels == out_channels

    def forward(self, c5, p5):
        x = p5 if self.use_P5 else c5
        p6 = self.p6(x)
        p7 = self.p7(HF.relu(p6))
        return [p6, p7]


class AssymetricDecoder(nn.Module):
    def __init__(self, in_channels, out_channels, in_feat_output_strides=(4, 8, 16, 32), out_feat_output_stride=4,
                 norm_fn=nn.BatchNorm2d, classifier_config=None):
        super().__init__()
        self.cls_cfg = classifier_config
        # reference fpn.py:163-167: norm_fn(num_features=out_channels) (Identity when None), ReLU after BatchNorm2d and
        # GELU after anything else.  Norms with a HIP kernel: BatchNorm2d / SyncBatchNorm / GroupNorm (instances of
        # the stock classes are retargeted); any other norm layer is refused here rather than run on ATen.
        is_bn = norm_fn in (nn.BatchNorm2d, BatchNorm2d)

        def make_norm():
            if norm_fn is None:
                return nn.Identity()
            if is_bn:
                return BatchNorm2d(num_features=out_channels)
            m = norm_fn(num_features=out_channels)
            if isinstance(m, nn.GroupNorm):
                m.__class__ = GroupNorm
            elif isinstance(m, nn.BatchNorm2d) and not isinstance(m, nn.SyncBatchNorm):
                m.__class__ = BatchNorm2d
            elif not isinstance(m, (GroupNorm, BatchNorm2d, nn.SyncBatchNorm, nn.Identity)):
                raise NotImplementedError(f'ever_amd AssymetricDecoder: norm layer {type(m).__name__} has no HIP kernel '
                                          f'(BatchNorm2d, SyncBatchNorm, GroupNorm and None are implemented)')
            return m
        self.blocks = nn.ModuleList()
        for os_in in in_feat_output_strides:
            n_up = int(math.log2(int(os_in))) - int(math.log2(int(out_feat_output_stride)))
            n_layers = n_up if n_up != 0 else 1
            self.blocks.append(HipSequential(*[
                HipSequential(
                    Conv2d(in_channels if i == 0 else out_channels, out_channels, 3, 1, 1, bias=False),
                    make_norm(),
                    ReLU(True) if is_bn else GELU(),
                    Bf16compatible(UpsamplingBilinear2d(scale_factor=2)) if n_up != 0 else nn.Identity(),
                ) for i in range(n_layers)]))
        if self.cls_cfg:
            scale_factor = classifier_config.get('scale_factor', 1)
            num_classes = classifier_config.get('num_classes', -1)
            kernel_size = classifier_config.get('kernel_size', 1)
            dropout_rate = classifier_config.get('dropout_rate', -1)
            self.dropout = Dropout(dropout_rate) if dropout_rate > 0 else nn.Identity()
            self.classifier = HipSequential(
                Conv2d(out_channels, num_classes, kernel_size, padding=(kernel_size - 1) // 2),
                Bf16compatible(UpsamplingBilinear2d(scale_factor=scale_factor)) if scale_factor > 1 else nn.Identity())

    def features(self, feat_list, branches=None):
        """mean of the per-level decoder outputs, before the classifier (what ChangeStar's ChangeMixin consumes).
        branches: the calling head's HF.HeadBranches session — branches 1.. run on its stream, joined here in front of the mean"""
        inner = []
        for i, block in enumerate(self.blocks):
            if branches is None:
                inner.append(block(feat_list[i]))
            else:
                with branches.level(i):
                    inner.append(block(feat_list[i]))
        if branches is not None:
            branches.join()
        if len(inner) == 4:
            return HF.mean4(*inner)
        out = inner[0]  # generic: running add then scale (same left-to-right association as python sum)
        for t in inner[1:]:
            out = HF.add(out, t)
        return _Scale.apply(out, 1.0 / len(inner))

    def forward(self, feat_list, branches=None):
        if self.cls_cfg and self._classifier_commutes():
            return self._forward_commuted(feat_list, branches)
        out = self.features(feat_list, branches)
        if self.cls_cfg:
            out = self.classifier(self.dropout(out))
        return out

    # ---- classifier before the last upsampling -------------------------------------------------------------------
    # Reference fpn.py:186-193: out = mean_i(branch_i(feat_i)); logits = up4(conv1x1(out) + b).  Every branch ends in
    # `ReLU -> bilinear x2` (or in the ReLU, at the output stride), the mean is linear, and a 1x1 convolution (per pixel,
    # across channels) commutes with a bilinear interpolation (per channel, across pixels; its weights sum to one, so the
    # bias passes through as well):
    #     conv1x1(mean_i up(r_i)) + b  =  mean_i up(conv1x1(r_i) + b)
    # The right-hand side upsamples num_classes channels instead of out_channels, never forms the full-width mean, and
    # its backward hands each branch a gradient at the branch's own resolution: at FarSeg-R50 512^2 x 16 this takes three
    # 268 MB upsampled tensors, the 1.34 GB mean pass, the classifier's read of the mean and their backward counterparts
    # off the step.  Same function, same parameters and gradients up to fp32 rounding (tests/test_decoder_commute_gpu.py);
    # taken only when nothing could observe the tensors that no longer exist (hooks), EVK_DECODER_COMMUTE=0 disables it.
    def _classifier_commutes(self):
        if os.environ.get('EVK_DECODER_COMMUTE', '1') == '0':
            return False
        conv = self.classifier[0]
        if tuple(conv.kernel_size) != (1, 1) or tuple(conv.stride) != (1, 1) or conv.groups != 1:
            return False
        if conv.out_channels >= conv.in_channels:
            return False
        if self.training and not isinstance(self.dropout, nn.Identity):
            return False
        if HF.observers_active():
            return False
        watched = [self.classifier, conv, self.dropout]
        for block in self.blocks:
            last = list(block)[-1]
            watched += [block, last, list(last)[-1], getattr(list(last)[-1], '_inner_module', None)]
        return not any(m is not None and (m._forward_hooks or m._forward_pre_hooks) for m in watched)

    def _bn_relu_classifier(self, last, x, conv):
        """`classifier(relu(bn(conv3x3(x))))` of a branch's last block with BatchNorm + ReLU + the 1x1 classifier as ONE
        consumer of the 3x3 convolution's output (HF.bn_relu_dot: the normalised 256-channel map is never written and the
        classifier's rank-K gradient never formed); None = run the layers one by one (eval / folded BatchNorm,
        GroupNorm or no norm, hooks, EVK_BN_DOT=0)."""
        from .fold import _takes_epilogue_stats
        import torch
        if os.environ.get('EVK_BN_DOT', '1') == '0' or not torch.is_grad_enabled():
            return None
        if not (len(last) == 4 and isinstance(last[0], Conv2d) and isinstance(last[2], nn.ReLU)):
            return None
        bn = last[1]
        if not (_takes_epilogue_stats(bn) and bn.training and bn.momentum is not None):
            return None
        if any(m._forward_hooks or m._forward_pre_hooks for m in (last[0], bn, last[2])):
            return None
        z = last[0](x, bn_stats=True)
        out = HF.bn_relu_dot(z, bn, conv)
        if out is None:      # no statistics records on z: finish the block layer by layer
            from .layers import run_sequence
            return conv(run_sequence(last[1:3], z))
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            bn._nbt_pending = getattr(bn, '_nbt_pending', 0) + 1
        return out

    def _commuted_branch(self, block, x, conv):
        from .layers import run_sequence
        subs = list(block)
        for sub in subs[:-1]:
            x = sub(x)
        last = list(subs[-1])
        up = last[-1]
        has_up = not isinstance(up, nn.Identity)
        z = self._bn_relu_classifier(last, x, conv)
        if z is None:
            x = run_sequence(last[:-1] if has_up else last, x)
            z = conv(x)
        return up(z) if has_up else z

    def _forward_commuted(self, feat_list, branches=None):
        from .layers import run_sequence
        conv = self.classifier[0]
        zs = []
        for i, block in enumerate(self.blocks):
            if branches is None:
                zs.append(self._commuted_branch(block, feat_list[i], conv))
            else:
                with branches.level(i):
                    zs.append(self._commuted_branch(block, feat_list[i], conv))
        if branches is not None:
            branches.join()
        if len(zs) == 4:
            out = HF.mean4(*zs)
        else:
            out = zs[0]
            for t in zs[1:]:
                out = HF.add(out, t)
            out = _Scale.apply(out, 1.0 / len(zs))
        return run_sequence(list(self.classifier)[1:], out)


class _Scale:
    @staticmethod
    def apply(x, alpha):
        import torch
        from .. import _C

        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, t):
                o = torch.empty_like(t)
                _C.call('evk_scale', t.data_ptr(), alpha, o.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
                return o

            @staticmethod
            def backward(ctx, g):
                g = HF.as_nhwc(g)
                o = torch.empty_like(g)
                _C.call('evk_scale', g.data_ptr(), alpha, o.data_ptr(), g.numel(), torch.cuda.current_stream().cuda_stream)
                return o

        return Fn.apply(x)
