"""TEST INFRASTRUCTURE — stock-PyTorch CPU restatement of ever_amd/module/changestar.py:ChangeMixin (same state-dict
keys).  There is no ChangeMixin definition in the reference tree (external ChangeStar project), so this oracle has
nothing to be pinned against: PARITY UNPINNED.  It states the published module with torch.nn layers; the GPU tests
compare the HIP module with it on identical hash-generated weights."""
import torch
import torch.nn as nn


class ChangeMixinRef(nn.Module):
    def __init__(self, in_channels=512, inner_channels=16, num_convs=4, scale_factor=4.0):
        super().__init__()
        layers = [nn.Sequential(nn.Conv2d(in_channels, inner_channels, 3, 1, 1), nn.BatchNorm2d(inner_channels), nn.ReLU(True))]
        layers += [nn.Sequential(nn.Conv2d(inner_channels, inner_channels, 3, 1, 1), nn.BatchNorm2d(inner_channels), nn.ReLU(True))
                   for _ in range(num_convs - 1)]
        layers.append(nn.Conv2d(inner_channels, 1, 3, 1, 1))
        layers.append(nn.UpsamplingBilinear2d(scale_factor=scale_factor))
        self.convs = nn.Sequential(*layers)

    def forward(self, t1, t2):
        n = t1.shape[0]
        out = self.convs(torch.cat([torch.cat([t1, t2], dim=1), torch.cat([t2, t1], dim=1)], dim=0))
        return out[:n], out[n:]
