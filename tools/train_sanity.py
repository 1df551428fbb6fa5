"""dev tool: a few hundred optimiser steps of FarSeg on a learnable synthetic task (label = smoothed band-0 threshold):
the loss must fall and stay finite under the three fp32-grade convolution arithmetics."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
from ever_amd.hip import functional as HF
dev = torch.device('cuda:0')
for mode in ('f16x2', 'bf16x3', 'f32'):
    HF.set_conv_math(mode)
    torch.manual_seed(0)
    widths = (64, 128, 256, 512)
    m = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet18', in_channels=3),
                              head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                                        fs_relation=dict(scene_embedding_channels=512)))).to(dev).train()
    opt = er.opt.FusedSGD(m.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator(device='cpu').manual_seed(1)
    hist = []
    for step in range(300):
        x = torch.randn(8, 3, 128, 128, generator=g)
        y = (torch.nn.functional.avg_pool2d(x[:, :1], 9, 1, 4)[:, 0] > 0.05).long()
        out = m(x.to(dev), y.to(dev))
        loss = sum(out.values())
        loss.backward()
        opt.step(); opt.zero_grad()
        if step % 50 == 0 or step == 299:
            hist.append(round(loss.item(), 4))
    print(mode, hist, 'finite' if all(h == h for h in hist) else 'NaN!')
