"""dev tool: BatchNorm(+ReLU) forward/backward on the FarSeg-R50 activation shapes, HIP-event timed; GB/s against
algorithmic bytes (fwd 3|x|, bwd 5|x|)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd.hip import functional as F
dev = torch.device('cuda:0')
SH = [(16, 64, 256, 256), (16, 256, 128, 128), (16, 64, 128, 128), (16, 512, 64, 64), (16, 128, 64, 64), (16, 1024, 32, 32), (16, 256, 32, 32), (16, 2048, 16, 16)]
def ev(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e-3
tf = tb = 0
for n, c, h, w in SH:
    x = F.empty_nhwc(n, c, h, w, dev).normal_().requires_grad_()
    g = torch.ones(c, device=dev, requires_grad=True); b = torch.zeros(c, device=dev, requires_grad=True)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    dy = torch.randn_like(x)
    fwd = lambda: F.batch_norm_act(x, g, b, rm, rv, True, 0.1, 1e-5, relu=True)
    t_f = ev(fwd)
    y = fwd()
    def bwd():
        torch.autograd.grad(y, (x, g, b), dy, retain_graph=True)
    t_b = ev(bwd)
    nb = x.numel() * 4
    tf += t_f; tb += t_b
    print(f'{str((n,c,h,w)):22s} {nb/1e6:7.1f} MB  fwd {t_f*1e6:7.1f} us {3*nb/t_f/1e9:7.0f} GB/s | bwd {t_b*1e6:7.1f} us {5*nb/t_b/1e9:7.0f} GB/s')
print(f'total fwd {tf*1e3:.3f} ms bwd {tb*1e3:.3f} ms')
