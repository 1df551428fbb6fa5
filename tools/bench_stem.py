"""dev tool: the stem convolution (7x7 s2 p3, 3-band 512^2, batch 16) forward + weight gradient, space-to-depth form vs
the exact-fp32 fallback (EVK_STEM_S2D=0), event-timed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
from ever_amd.hip import functional as HF
dev = torch.device('cuda:0')
conv = er.module.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev)
x = torch.randn(16, 3, 512, 512, device=dev)
g = torch.randn(16, 64, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
def run(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
def fwd_only():
    with torch.no_grad():
        return HF.stem_conv7x7s2(x, conv.weight) if HF.stem_conv_applicable(x, conv) else conv(x)
def fwd_bwd():
    y = HF.stem_conv7x7s2(x, conv.weight) if HF.stem_conv_applicable(x, conv) else conv(x)
    y.backward(g)
    conv.weight.grad = None
f, fb = run(fwd_only), run(fwd_bwd)
gf = 2.0 * 16 * 256 * 256 * 64 * 147 / 1e9
print(f'S2D={os.environ.get("EVK_STEM_S2D", "1")}: forward {f:7.1f} us ({gf / f * 1e3:6.1f} TF), forward+wgrad {fb:7.1f} us, wgrad ~{fb - f:7.1f} us ({gf / (fb - f) * 1e3:6.1f} TF)')
