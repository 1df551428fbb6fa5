"""dev tool: per-launch table of the conv kernels inside one FarSeg-R50 training step (HIP-event timed):
family, GFLOP, us, TFLOP/s, sorted by time — shows which layers sit furthest below the big-layer rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
from ever_amd.hip import timing
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = er.module.FarSeg(dict()).to(dev).train()
x = torch.randn(16, 3, 512, 512, device=dev)
y = (torch.rand(16, 512, 512, device=dev) > 0.5).long()
for it in range(3):
    t = timing.KernelTimer() if it == 2 else None
    if t: t.__enter__()
    loss = sum(m(x, y).values()); loss.backward()
    if t: t.__exit__()
    m.zero_grad(set_to_none=True)
torch.cuda.synchronize()
rows = [(f, fl, s.elapsed_time(e) * 1e3) for f, fl, nb, s, e in t.records if fl > 0]
tot = sum(r[2] for r in rows)
agg = {}
for f, fl, us in rows:
    k = (f, round(fl / 1e9, 2))
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += us
print(f'conv launches {len(rows)}, total {tot/1e3:.2f} ms')
cum = 0
for (f, gf), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    cum += us
    print(f'{f:16s} {gf:8.2f} GF x{n:3d}  {us/n:8.1f} us  {gf*n/us*1e-3:7.1f} TF  share {us/tot*100:5.1f}% cum {cum/tot*100:5.1f}%')
