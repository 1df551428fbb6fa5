import sys, os, torch, time
sys.path.insert(0, os.getcwd())
import bench, ever_amd as er
from ever_amd.hip import functional as HF
cfg = sys.argv[1]
dev = torch.device('cuda:0')
torch.manual_seed(2333)
model, inputs, *_ = bench.make_workload(er, cfg, dev, bench.BATCH, 0)
model = model.to(dev).train()
opt = er.opt.FusedSGD(model.parameters(), lr=0.007, momentum=0.9, weight_decay=1e-4)
def step():
    out = model(*inputs)
    sum(v for k, v in out.items() if k.endswith('loss')).backward()
    opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
for k in HF.wgrad_stream_stats: HF.wgrad_stream_stats[k] = 0
t = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print(cfg, 'ms/step', (time.perf_counter() - t) * 100, 'wgrad stats per step', {k: v / 10 for k, v in HF.wgrad_stream_stats.items()},
      'absmax', dict(HF.absmax_stats), 'selftest', getattr(HF, '_SIDE_SELFTEST', None))
