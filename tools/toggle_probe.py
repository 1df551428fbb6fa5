"""dev tool (GPU box): per-step GPU time when the weight-gradient side stream is switched off for single steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
from ever_amd import _C
from ever_amd.hip import functional as HF
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
for _ in range(6): step()
torch.cuda.synchronize()
def timed(tag, on):
    HF.set_wgrad_stream(on)
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); h = time.perf_counter() - t0; torch.cuda.synchronize()
    print(f'{tag}: gpu+host {(time.perf_counter() - t0) * 1e3:.1f} ms, host {h * 1e3:.1f}', torch.cuda.memory_stats()['num_device_alloc'])
for on in (1, 1, 0, 0, 0, 1, 1, 0, 1, 0, 1, 1):
    timed('on ' if on else 'off', bool(on))
