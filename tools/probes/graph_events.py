"""probe (GPU box): do HIP events recorded inside a captured graph (torch.cuda.Event(external=True)) time the kernels of a replay?"""
import torch
dev = torch.device('cuda:0')
a = torch.randn(4096, 4096, device=dev)
s = torch.cuda.Stream()
evs = []
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True, external=True); e0.record()
        b = a @ a
        e1 = torch.cuda.Event(enable_timing=True, external=True); e1.record()
        evs.append((e0, e1))
for rep in range(3):
    g.replay()
    torch.cuda.synchronize()
    print('replay', rep, [round(x.elapsed_time(y) * 1e3, 1) for x, y in evs], 'us')
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); b = a @ a; t1.record(); torch.cuda.synchronize(); print('eager matmul', round(t0.elapsed_time(t1) * 1e3, 1), 'us')
