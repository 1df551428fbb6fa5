"""Layer blocks of the FarSeg path (reference ever/module/ops.py:45-64,152-181), HIP-backed.
Child indices ('0' conv, '1' bn/identity, '2' relu/identity) match the reference so state-dict keys
such as `fpn.fpn_inner1.0.weight` are identical."""
import torch.nn as nn

from .layers import BatchNorm2d, Conv2d, HipSequential, ReLU, UpsamplingBilinear2d

__all__ = ['ConvBlock', 'Bf16compatible', 'ConvUpsampling']


class ConvBlock(HipSequential):
    """conv -> (BN) -> (ReLU); reference ops.py:45-64."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=False, bn=True, relu=True, init_fn=None):
        super().__init__(
            Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias),
            BatchNorm2d(out_channels) if bn else nn.Identity(),
            ReLU(True) if relu else nn.Identity(),
        )
        if init_fn:
            self.apply(init_fn)

    @staticmethod
    def same_padding(kernel_size, dilation):
        return dilation * (kernel_size - 1) // 2


class Bf16compatible(nn.Module):
    """Reference ops.py:152-166 runs the wrapped (upsampling) module in fp32 under bf16 autocast.  The
    HIP path computes in fp32 throughout, so this only preserves the `_inner_module` key structure."""

    def __init__(self, module):
        super().__init__()
        self._inner_module = module

    def forward(self, x):
        return self._inner_module(x)


class ConvUpsampling(HipSequential):
    """conv (+bias) -> bilinear(align_corners=True) ; reference ops.py:169-181."""

    def __init__(self, in_channels, out_channels, scale_factor, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__(
            Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation),
            Bf16compatible(UpsamplingBilinear2d(scale_factor=scale_factor)),
        )
