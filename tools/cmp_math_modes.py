"""Per-parameter gradient agreement of two convolution arithmetics on one FarSeg-R50 step (debug aid).
usage: python tools/cmp_math_modes.py [modeA] [modeB] [tile] [batch]"""
import sys
import torch
import ever_amd as er
from ever_amd.hip import functional as F

a, b = (sys.argv[1:3] + ['bf16x3', 'bf16'])[:2] if len(sys.argv) >= 3 else ('bf16x3', 'bf16')
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 256
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
x = torch.randn(batch, 3, tile, tile, generator=g).to(dev)
y = (torch.rand(batch, tile, tile, generator=g) < 0.3).long().to(dev)
res = {}
for mode in (a, b):
    F.set_conv_math(mode)
    torch.manual_seed(5)
    m = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet50'))).to(dev).train()
    out = m(x, y)
    sum(out.values()).backward()
    torch.cuda.synchronize()
    res[mode] = ({k: float(v) for k, v in out.items()}, {n: p.grad.double().cpu() for n, p in m.named_parameters() if p.grad is not None})
print(res[a][0], res[b][0])
for n in res[a][1]:
    ga, gb = res[a][1][n].flatten(), res[b][1][n].flatten()
    cos = float(torch.dot(ga, gb) / (ga.norm() * gb.norm() + 1e-300))
    print(f'{n:60s} cos {cos:8.5f}  |a| {float(ga.norm()):.3e} |b| {float(gb.norm()):.3e}')
