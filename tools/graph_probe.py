"""dev tool: which part of the step survives hipGraph capture.  usage: python tools/graph_probe.py <stage>  (fwd|bwd|opt|tiny)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
from ever_amd.hip import functional as HF
stage = sys.argv[1]
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet18', in_channels=4),
                          head=dict(fpn=dict(in_channels_list=(64, 128, 256, 512), out_channels=256),
                                    fs_relation=dict(scene_embedding_channels=512)))).to(dev).train()
opt = er.opt.FusedSGD(m.parameters(), lr=0.01, momentum=0.9)
x = torch.randn(2, 4, 128, 128, device=dev); y = (torch.rand(2, 128, 128, device=dev) < 0.3).long()
def full():
    out = m(x, y); sum(out.values()).backward(); opt.step(); opt.zero_grad(set_to_none=True)
if stage.startswith('gts'):
    from ever_amd.core.graph import GraphedTrainStep
    def step_fn(x_, y_):
        out = m(x_, y_); sum(out.values()).backward()
        if 'noclip' not in stage: opt.fused_clip(max_norm=35)
        opt.step(); opt.zero_grad(set_to_none=True)
        return out if 'noout' not in stage else {}
    gs = GraphedTrainStep(step_fn, opt, modules=(m,))
    for i in range(6):
        out = gs(x, y)
        torch.cuda.synchronize()
        print('step', i, {k: float(v) for k, v in out.items()}, flush=True)
    print('replayed', stage, flush=True)
    sys.exit(0)
for _ in range(3): full()
opt.use_device_lr(dev)
full()
HF._ZERO_POOL.clear()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if stage == 'tiny':
    a = torch.zeros(1024, device=dev)
    with torch.cuda.graph(g):
        b = HF.relu(a.view(1, 4, 16, 16))
elif stage == 'conv':
    w = m.en.resnet.layer1[0].conv1.weight
    a = torch.randn(2, 64, 32, 32, device=dev)
    with torch.no_grad(), torch.cuda.graph(g):
        b = HF.conv2d(a, w, None, padding=1)
elif stage == 'enc':
    with torch.no_grad(), torch.cuda.graph(g):
        b = m.en(x)
elif stage == 'fwd':
    with torch.no_grad(), torch.cuda.graph(g):
        out = m(x, y)
elif stage == 'fwdgrad':
    with torch.cuda.graph(g):
        out = m(x, y)
elif stage == 'bwd':
    with torch.cuda.graph(g):
        out = m(x, y); sum(out.values()).backward()
elif stage == 'clip':
    def full_clip():
        out = m(x, y); sum(out.values()).backward(); opt.fused_clip(max_norm=35); opt.step(); opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        full_clip()
elif stage == 'cliponly':
    out = m(x, y); sum(out.values()).backward()
    with torch.cuda.graph(g):
        opt.fused_clip(max_norm=35)
else:
    with torch.cuda.graph(g):
        full()
print('captured', stage, flush=True)
g.replay(); torch.cuda.synchronize()
print('replayed', stage, flush=True)
