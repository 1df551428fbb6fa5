"""dev tool (GPU box): a longer run of the bench training step — the loss stays finite and the allocator stops growing.
usage: python tools/soak.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
from ever_amd import _C
from ever_amd.hip import functional as HF
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device('cuda:0'); torch.cuda.set_device(dev); _C.load()
torch.manual_seed(2333)
model = er.module.FarSeg(dict()).to(dev).train()
opt = er.opt.FusedSGD(model.parameters(), lr=0.007, momentum=0.9, weight_decay=1e-4)
import bench
batches = [bench.make_batch(dev, 16, s) for s in range(4)]
def mem():
    m = torch.cuda.memory_stats()
    return (m['num_device_alloc'], round(m['reserved_bytes.all.current'] / 2**30, 2), round(m['allocated_bytes.all.peak'] / 2**30, 2))
t0 = time.perf_counter()
for i in range(n):
    x, y = batches[i % 4]
    out = model(x, y)
    loss = sum(v for k, v in out.items() if k.endswith('loss'))
    loss.backward()
    opt.fused_clip(max_norm=35)
    opt.step(); opt.zero_grad(set_to_none=True)
    if i in (50, 200, n - 1) or i % 500 == 0:
        print(f'step {i}: loss {float(loss):.5f} allocator (device allocs, reserved GB, peak GB) {mem()} side/main {HF.wgrad_stream_stats}', flush=True)
torch.cuda.synchronize()
print(f'{n} steps in {time.perf_counter() - t0:.1f} s')
