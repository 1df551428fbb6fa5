"""dev tool (GPU box): one workload's step time in THIS tree, to be run from two trees on one box — the comparison that an
in-build switch cannot make (a change to a shared epilogue alters the registers of every kernel that instantiates it, on
both sides of the switch: DESIGN 2.9).  Workflow (build container):
    git worktree add -f _r03 <older commit>; make -C _r03/ever_amd/csrc -j; cp tools/ab_tree.py _r03/tools/
    gpurun -- 'for c in c2 c5; do (cd _r03 && python tools/ab_tree.py $c); python tools/ab_tree.py $c; done'
(_r03/ is in .git/info/exclude; its built library travels with the snapshot.)  usage: python tools/ab_tree.py c2|c3|c4|c5"""
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
