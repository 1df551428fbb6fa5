"""dev tool: host-side (enqueue) time per training step vs GPU time: how far the CPU runs ahead of the GPU."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = er.module.FarSeg(dict()).to(dev).train()
opt = er.opt.FusedSGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
x = torch.randn(16, 3, 512, 512, device=dev)
y = (torch.rand(16, 512, 512, device=dev) > 0.5).long()
def step():
    loss = sum(m(x, y).values()); loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(5): step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'enqueue {1e3*(t1-t0)/K:.2f} ms/step, total {1e3*(t2-t0)/K:.2f} ms/step (GPU-bound if enqueue < total)')
# pure host cost: the same loop with the GPU kept idle-free is not separable; report CPU time of the process instead
import resource
r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize()
r1 = resource.getrusage(resource.RUSAGE_SELF); t1 = time.perf_counter()
print(f'process CPU time {1e3*((r1.ru_utime-r0.ru_utime)+(r1.ru_stime-r0.ru_stime))/K:.2f} ms/step over {1e3*(t1-t0)/K:.2f} ms wall')

# host-bound regime: the same graph on a 2x3x64x64 batch (kernels of a few microseconds): wall time per step ~ host cost
xs = torch.randn(2, 3, 64, 64, device=dev); ys = (torch.rand(2, 64, 64, device=dev) > 0.5).long()
def small():
    loss = sum(m(xs, ys).values()); loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(5): small()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): small()
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f'tiny batch (host-bound): {1e3*(t1-t0)/K:.2f} ms/step')
