"""dev tool: forward time of the short-K 1x1 layer shapes (+ one 3x3 layer as a clock reference) under the current
environment switches; run once per variant and compare the ratios to the reference row."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd import _C
dev = torch.device('cuda:0'); lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
B = 16
def run(h, cin, cout, k, p, it=20):
    d = _C.ConvDesc(B, h, h, cin, h, h, cout, k, k, 1, 1, p, p, 1, 1)
    x = torch.randn(B, h, h, cin, device=dev); wt = torch.randn(cout, k, k, cin, device=dev) * 0.05
    y = torch.empty(B, h, h, cout, device=dev)
    pf = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
    _C.call('evk_conv2d_split_weight', ctypes.byref(d), wt.data_ptr(), 0, pf.data_ptr(), st)
    f = lambda: _C.call('evk_conv2d_fwd_x3', ctypes.byref(d), x.data_ptr(), pf.data_ptr(), None, y.data_ptr(), 0, st)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / it * 1e3
    gf = 2.0 * B * h * h * cin * cout * k * k / 1e9
    return us, gf / us * 1e3
ref_us, ref_tf = run(64, 256, 256, 3, 1)
print(f'{os.environ.get("TAG", "")}: ref 3x3x256@64^2 {ref_us:7.1f} us {ref_tf:6.1f} TF')
for (h, ci, co) in [(128, 64, 256), (128, 256, 64), (128, 256, 256), (64, 128, 512), (64, 512, 128), (32, 256, 1024), (32, 1024, 256), (64, 512, 256)]:
    us, tf = run(h, ci, co, 1, 0)
    print(f'   1x1 {ci:4d}->{co:4d} @{h:3d}^2  {us:7.1f} us {tf:6.1f} TF   x{us / ref_us:5.2f} of ref')
