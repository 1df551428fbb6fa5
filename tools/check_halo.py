"""dev tool: x3 kernels (incl. the LDS-halo 3x3 kernel) vs the exact-fp32 MFMA kernels on 3x3 'same' shapes."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd import _C
dev = torch.device('cuda:0'); lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
SH = [(16, 256, 256, 32, 32), (16, 128, 128, 64, 64), (16, 256, 256, 64, 64), (16, 64, 64, 128, 128), (16, 256, 128, 64, 64),
      (16, 128, 256, 32, 32), (16, 256, 256, 128, 128), (16, 512, 512, 16, 16), (16, 256, 256, 16, 16), (8, 256, 256, 32, 32),
      (16, 64, 192, 128, 128), (16, 192, 64, 128, 128), (3, 32, 64, 128, 128), (16, 48, 80, 64, 64)]
for n, cin, cout, h, w in SH:
    d = _C.ConvDesc(n, h, w, cin, h, w, cout, 3, 3, 1, 1, 1, 1, 1, 1)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(n, h, w, cin, generator=g) + 0.3).to(dev); wt = (torch.randn(cout, 3, 3, cin, generator=g) * 0.05).to(dev)
    dy = torch.randn(n, h, w, cout, generator=g).to(dev)
    y, y3, dx, dx3 = torch.empty(n, h, w, cout, device=dev), torch.empty(n, h, w, cout, device=dev), torch.empty_like(x), torch.empty_like(x)
    wpk = torch.empty(cin * 9 * cout, device=dev)
    pf = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
    pd = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 1), dtype=torch.uint8, device=dev)
    _C.call('evk_conv2d_split_weight', ctypes.byref(d), wt.data_ptr(), 0, pf.data_ptr(), st)
    _C.call('evk_conv2d_split_weight', ctypes.byref(d), wt.data_ptr(), 1, pd.data_ptr(), st)
    _C.call('evk_conv2d_pack_dgrad_weight', ctypes.byref(d), wt.data_ptr(), wpk.data_ptr(), st)
    _C.call('evk_conv2d_fwd', ctypes.byref(d), x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), 0, st)
    _C.call('evk_conv2d_fwd_x3', ctypes.byref(d), x.data_ptr(), pf.data_ptr(), None, y3.data_ptr(), 0, st)
    _C.call('evk_conv2d_dgrad', ctypes.byref(d), dy.data_ptr(), wpk.data_ptr(), None, dx.data_ptr(), st)
    _C.call('evk_conv2d_dgrad_x3', ctypes.byref(d), dy.data_ptr(), pd.data_ptr(), None, dx3.data_ptr(), st)
    torch.cuda.synchronize()
    r = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print(f'{(n, cin, cout, h, w)}: fwd {r(y3, y):.2e}  dgrad {r(dx3, dx):.2e}')
