"""TEST INFRASTRUCTURE — stock-PyTorch CPU restatement of ever_amd/module/freenet.py (same state-dict keys).
There is no FreeNet definition in the reference tree (external project), so this oracle has nothing to be pinned
against: PARITY UNPINNED.  It states the same published architecture with torch.nn layers; the GPU tests compare the
HIP model with it on identical hash-generated weights."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _cgr(cin, cout, g):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.GroupNorm(g, cout), nn.ReLU(inplace=True))


class FreeNetRef(nn.Module):
    def __init__(self, in_channels=200, num_classes=16, num_blocks=(1, 1, 1, 1), reduction_ratio=1.0):
        super().__init__()
        r = int(16 * reduction_ratio)
        chans = [int(c * reduction_ratio / r) * r for c in (96, 128, 192, 256)]
        ops = [_cgr(in_channels, chans[0], r), nn.Sequential(*[_cgr(chans[0], chans[0], r) for _ in range(num_blocks[0])]),
               nn.Identity()]
        for i in range(1, 4):
            ops += [nn.Sequential(nn.Conv2d(chans[i - 1], chans[i], 3, 2, 1), nn.ReLU(inplace=True)),
                    nn.Sequential(*[_cgr(chans[i], chans[i], r) for _ in range(num_blocks[i])]), nn.Identity()]
        self.feature_ops = nn.ModuleList(ops)
        inner = int(128 * reduction_ratio)
        self.reduce_1x1convs = nn.ModuleList([nn.Conv2d(c, inner, 1) for c in chans])
        self.fuse_3x3convs = nn.ModuleList([nn.Conv2d(inner, inner, 3, 1, 1) for _ in range(4)])
        self.cls_pred_conv = nn.Conv2d(inner, num_classes, 1)

    def logits(self, x):
        feats = []
        for op in self.feature_ops:
            x = op(x)
            if isinstance(op, nn.Identity):
                feats.append(x)
        inner = [c(f) for c, f in zip(self.reduce_1x1convs, feats)]
        inner.reverse()
        out = inner[0]
        for i in range(len(inner) - 1):
            out = self.fuse_3x3convs[i](F.interpolate(out, scale_factor=2.0, mode='nearest') + inner[i + 1])
        return self.cls_pred_conv(out)

    def loss(self, logit, y, weight=None):
        t = y.long() - 1
        if weight is not None:
            t = torch.where(weight > 0, t, torch.full_like(t, -1))
        return F.cross_entropy(logit, t, ignore_index=-1)
