"""Quick wall-clock of FarSeg-R50 fwd+bwd on one GPU (dev tool; bench.py is the contract)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd.module import FarSeg  # noqa: E402

B = int(os.environ.get('BATCH', 16))
HW = int(os.environ.get('HW', 512))
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = FarSeg(dict()).to(dev).train()
x = torch.randn(B, 3, HW, HW, device=dev)
y = (torch.rand(B, HW, HW, device=dev) < 0.3).long()
y[:, :8, :8] = 255


def step():
    out = m(x, y)
    sum(out.values()).backward()
    m.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.time()
N = 5
for _ in range(N):
    step()
torch.cuda.synchronize()
dt = (time.time() - t0) / N
print(f'batch {B} @ {HW}: {dt*1e3:.1f} ms/step, {B/dt:.1f} tiles/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
if os.environ.get('PROFILE'):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=30, max_name_column_width=60))
