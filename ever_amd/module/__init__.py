from . import loss  # noqa: F401
from .changestar import ChangeMixin, ChangeStarFarSeg
from .farseg import FarSeg, FarSegPP
from .fpn import FPN, AssymetricDecoder
from .freenet import FreeNet
from .fs_relation import FarSegHead, FarSegPPHead, FSRelation, FSRelationV2
from .layers import (AdaptiveAvgPool2d, BatchNorm2d, Conv2d, ConvTranspose2d, HipSequential, MaxPool2d, ReLU, UpsamplingBilinear2d,
                     to_hip)
from .ops import Bf16compatible, ConvBlock, ConvUpsampling
from .resnet import ResNetEncoder

__all__ = ['ResNetEncoder', 'FPN', 'AssymetricDecoder', 'FSRelation', 'FSRelationV2', 'FarSegHead', 'FarSegPPHead', 'FarSeg', 'FarSegPP', 'FreeNet', 'ChangeMixin', 'ChangeStarFarSeg', 'ConvBlock',
           'Bf16compatible', 'ConvUpsampling', 'Conv2d', 'ConvTranspose2d', 'BatchNorm2d', 'ReLU', 'MaxPool2d', 'UpsamplingBilinear2d',
           'AdaptiveAvgPool2d', 'HipSequential', 'to_hip', 'loss']
