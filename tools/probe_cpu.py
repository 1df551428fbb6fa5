import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import farseg_ref
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads default', torch.get_num_threads())
try:
    print(open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e:
    print('no cgroup cpu.max', e)
th = int(sys.argv[1]); b = int(sys.argv[2])
torch.set_num_threads(th)
net = farseg_ref.FarSegRef('resnet50', 3, 1).train()
x = torch.randn(b, 3, 512, 512); y = (torch.rand(b, 512, 512) < 0.3).long()
for i in range(3):
    t0 = time.time(); out = net(x, y); sum(out.values()).backward(); dt = time.time() - t0
    print(f'threads {th} batch {b}: {dt:.2f} s/step -> {b/dt:.3f} tiles/s', flush=True)
