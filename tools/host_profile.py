"""dev tool (GPU box): cProfile of the host side of the bench training step (where the ~16-20 ms of Python / ctypes /
autograd time per step go; the GPU needs ~36 ms, the launch queue holds about two steps).
usage: python tools/host_profile.py [steps]"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
from ever_amd import _C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
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
t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize()
print(f'host enqueue of one step on an empty queue: {(t1 - t0) * 1e3:.1f} ms')
def mem():
    m = torch.cuda.memory_stats()
    return {k: m.get(k) for k in ('num_device_alloc', 'num_device_free', 'num_alloc_retries', 'num_ooms')} | {
        'reserved_GB': round(m['reserved_bytes.all.current'] / 2**30, 2), 'allocated_GB': round(m['allocated_bytes.all.current'] / 2**30, 2),
        'peak_alloc_GB': round(m['allocated_bytes.all.peak'] / 2**30, 2)}
print('allocator', mem(), 'env', {k: v for k, v in os.environ.items() if 'ALLOC' in k or 'PYTORCH' in k})
for i in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f'step {i}: host {(t1 - t0) * 1e3:.1f} ms', mem())
for i in range(3):   # no synchronisation in between, as the bench loop
    t0 = time.perf_counter(); step(); t1 = time.perf_counter()
    print(f'unsynchronised step {i}: host {(t1 - t0) * 1e3:.1f} ms')
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    step(); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
