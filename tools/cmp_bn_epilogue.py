"""Per-parameter gradient agreement with / without BatchNorm statistics from the conv epilogue (debug aid)."""
import sys
import torch
import ever_amd as er
from ever_amd.hip import functional as F
from tests import plumbing_common as pc

dev = torch.device('cuda:0')
widths = (64, 128, 256, 512)
res = {}
loader = torch.utils.data.DataLoader(pc.ToyTiles(), batch_size=2, shuffle=False)
x, y = next(iter(loader))
x = x.to(dev)
y = {k: v.to(dev) for k, v in y.items()} if isinstance(y, dict) else y.to(dev)
for epi in (False, True):
    F._BN_EPILOGUE = epi
    model = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet18', in_channels=4),
                                  head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                                            fs_relation=dict(scene_embedding_channels=512))))
    ora = pc.OracleFarSeg(dict())
    model.load_state_dict(ora.state_dict(), strict=True)
    model = model.to(dev).train()
    out = model(x, y)
    sum(out.values()).backward()
    torch.cuda.synchronize()
    res[epi] = ({k: float(v) for k, v in out.items()},
                {n: p.grad.double().cpu() for n, p in model.named_parameters() if p.grad is not None},
                {n: b.double().cpu().clone() for n, b in model.named_buffers() if 'running' in n})
print(res[False][0], res[True][0])
for n in res[False][1]:
    a, b = res[False][1][n].flatten(), res[True][1][n].flatten()
    rel = float((a - b).norm() / (a.norm() + 1e-300))
    if rel > 1e-4:
        print(f'{n:55s} rel {rel:.3e} |a| {float(a.norm()):.3e}')
for n in res[False][2]:
    a, b = res[False][2][n].flatten(), res[True][2][n].flatten()
    rel = float((a - b).abs().max() / (a.abs().max() + 1e-300))
    if rel > 1e-5:
        print(f'BUF {n:51s} rel {rel:.3e}')

# ---- forward activations, module by module
acts = {}
for epi in (False, True):
    F._BN_EPILOGUE = epi
    model = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet18', in_channels=4),
                                  head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                                            fs_relation=dict(scene_embedding_channels=512))))
    model.load_state_dict(pc.OracleFarSeg(dict()).state_dict(), strict=True)
    model = model.to(dev).train()
    rec = {}
    grec = {}
    def mk(name):
        def hook(m, i, o):
            if torch.is_tensor(o):
                rec[name] = o.detach().double().cpu()
                if o.requires_grad:
                    o.register_hook(lambda g, _n=name: grec.__setitem__(_n, g.detach().double().cpu()))
        return hook
    for n, m in model.named_modules():
        m.register_forward_hook(mk(n))
    out = model(x, y)
    sum(out.values()).backward()
    torch.cuda.synchronize()
    acts[epi] = (rec, grec)
print('---- forward outputs that differ')
for n in acts[False][0]:
    a, b = acts[False][0][n], acts[True][0][n]
    rel = float((a - b).abs().max() / (a.abs().max() + 1e-300))
    if rel > 2e-6:
        print(f'{n:55s} {tuple(a.shape)} rel {rel:.3e}')
print('---- output gradients that differ')
for n in acts[False][1]:
    if n not in acts[True][1]:
        continue
    a, b = acts[False][1][n], acts[True][1][n]
    rel = float((a - b).abs().max() / (a.abs().max() + 1e-300))
    if rel > 1e-4:
        print(f'{n:55s} {tuple(a.shape)} rel {rel:.3e}')
