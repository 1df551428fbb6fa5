"""dev tool: bilinear upsampling forward and backward on the FarSeg-R50 decoder shapes, HIP-event timed, TB/s against
algorithmic bytes (|in| + |out|), with a plain copy / fill of the same size beside it."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd.hip import functional as F
dev = torch.device('cuda:0')
def ev(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e-3
for (n, c, h, w, sc) in [(16, 128, 64, 64, 2), (16, 128, 32, 32, 2), (16, 128, 16, 16, 2), (16, 256, 64, 64, 2), (16, 256, 32, 32, 2), (16, 256, 16, 16, 2), (16, 1, 128, 128, 4)]:
    x = F.empty_nhwc(n, c, h, w, dev).normal_().requires_grad_()
    y = F.upsample_bilinear(x, sc)
    dy = torch.randn_like(y)
    tf = ev(lambda: F.upsample_bilinear(x, sc))
    tb = ev(lambda: torch.autograd.grad(y, x, dy, retain_graph=True))
    nb = (x.numel() + y.numel()) * 4
    print(f'bilinear x{sc} {(n,c,h,w)}: {nb/1e6:6.1f} MB fwd {tf*1e6:6.1f} us {nb/tf/1e12:5.2f} TB/s | bwd {tb*1e6:6.1f} us {nb/tb/1e12:5.2f} TB/s')
a = torch.empty(16 * 128 * 128 * 256, device=dev); b = torch.empty_like(a)
t = ev(lambda: b.copy_(a)); print(f'copy 268 MB: {t*1e6:.1f} us {2*a.numel()*4/t/1e12:.2f} TB/s')
t = ev(lambda: b.fill_(1.0)); print(f'fill 268 MB: {t*1e6:.1f} us {a.numel()*4/t/1e12:.2f} TB/s')
