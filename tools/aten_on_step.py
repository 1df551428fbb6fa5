"""dev tool (GPU box): which stock ATen kernels (copies, adds, fills) still run inside a training step, with the Python
frames that issue them.  usage: python tools/aten_on_step.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
from ever_amd import _C
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0'); torch.cuda.set_device(dev); _C.load()
torch.manual_seed(2333)
model = er.module.FarSeg(dict()).to(dev).train()
opt = er.opt.FusedSGD(model.parameters(), lr=0.007, momentum=0.9, weight_decay=1e-4)
import bench
x, y = bench.make_batch(dev, 16, 0)
def step():
    out = model(x, y)
    sum(v for k, v in out.items() if k.endswith('loss')).backward()
    opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
from collections import Counter
c = Counter()
for ev in prof.events():
    if ev.name.startswith('aten::') and ev.name in ('aten::copy_', 'aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_', 'aten::mul', 'aten::clone',
                                                    'aten::contiguous', 'aten::cat', 'aten::sum', 'aten::div', 'aten::mul_', 'aten::to', 'aten::_to_copy'):
        stack = [f for f in (ev.stack or []) if 'ever_amd' in f or 'bench' in f or 'tools' in f]
        shapes = str(ev.input_shapes)[:60]
        c[(ev.name, shapes, ' <- '.join(s.split('/')[-1] for s in stack[:3]))] += 1
for (n, sh, st), k in sorted(c.items(), key=lambda kv: -kv[1])[:60]:
    print(k, n, sh, '|', st)
