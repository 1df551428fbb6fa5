"""dev tool: per-launch table of the conv kernels inside one FarSeg-R50 training step (HIP-event timed):
family, GFLOP, us, TFLOP/s, sorted by time — shows which layers sit furthest below the big-layer rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
from ever_amd.hip import timing
from ever_amd.hip import functional as _HF0
import bench
_HF0.set_wgrad_stream(False)     # every launch alone on the chip (durations under overlap say how two streams share it)
dev = torch.device('cuda:0')
torch.manual_seed(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'      # BASELINE.json configs[1..4] as bench.py --config builds them
m, inputs = bench.make_workload(er, cfg, dev, bench.BATCH, 0)[:2]
m = m.to(dev).train()
print(f'config {cfg}')
for it in range(3):
    t = timing.KernelTimer() if it == 2 else None
    if t: t.__enter__()
    loss = sum(v for k, v in m(*inputs).items() if k.endswith('loss')); loss.backward()
    if t: t.__exit__()
    m.zero_grad(set_to_none=True)
torch.cuda.synchronize()
from ever_amd.hip import functional as _HF
_PASSES = {'f16x2': 3, 'bf16x3': 6, 'bf16': 1}.get(_HF.get_conv_math())
PEAK_TF = 2500e12 / _PASSES if _PASSES else 157.3e12   # 16-bit MFMA peak / partial products of the arithmetic in use
PEAK_HBM = 8.0e12                                      # HBM3E peak
rows = [(f, fl, nb, s.elapsed_time(e) * 1e3) for f, fl, nb, s, e, _sc in t.records if fl > 0]
tot = sum(r[3] for r in rows)
agg = {}
for f, fl, nb, us in rows:
    k = (f, round(fl / 1e9, 3), round(nb / 1e6, 1))
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += us
print(f'conv launches {len(rows)}, total {tot/1e3:.2f} ms')
print(f'arithmetic {_HF.get_conv_math()}: bound = max(flop / {PEAK_TF / 1e12:.1f} TF, algorithmic bytes / 8 TB/s) per launch; frac = bound / measured')
cum, bound_tot = 0, 0.0
for (f, gf, mb), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    cum += us
    t_mfma, t_hbm = gf * 1e9 / PEAK_TF * 1e6, mb * 1e6 / PEAK_HBM * 1e6
    bound = max(t_mfma, t_hbm)
    bound_tot += bound * n
    print(f'{f:16s} {gf:8.3f} GF {mb:7.1f} MB x{n:3d}  {us/n:8.1f} us  {gf*n/us*1e3:7.1f} TF  '
          f'{"hbm " if t_hbm > t_mfma else "mfma"} bound {bound:7.1f} us frac {bound*n/us:4.2f}  '
          f'share {us/tot*100:5.1f}% cum {cum/tot*100:5.1f}%')
print(f'sum of per-launch bounds {bound_tot/1e3:.2f} ms = {bound_tot/tot:.2f} of the measured conv time')

# HBM-bound families: (family, MB) -> launches, us, TB/s against algorithmic bytes
rows = [(f, nb, s.elapsed_time(e) * 1e3) for f, fl, nb, s, e, _sc in t.records if fl == 0 and nb > 0]
tot = sum(r[2] for r in rows)
agg = {}
for f, nb, us in rows:
    a = agg.setdefault((f, round(nb / 1e6, 1)), [0, 0.0]); a[0] += 1; a[1] += us
print(f'HBM-bound launches {len(rows)}, total {tot/1e3:.2f} ms (a launch = one C-ABI call, e.g. bn_bwd = partial + final + apply)')
for (f, mb), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f'{f:16s} {mb:8.1f} MB x{n:3d}  {us/n:8.1f} us  {mb*n/us/1e6*1e6:7.2f} TB/s  share {us/tot*100:5.1f}%')
