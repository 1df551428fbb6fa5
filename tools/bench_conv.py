"""Micro-benchmark of the conv kernels on the FarSeg-R50 layer shapes, batch 16 @ 512x512.
Prints TFLOP/s per layer and direction; used to steer kernel work, not part of the bench contract."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd import _C  # noqa: E402
from ever_amd.hip.workspace import workspace  # noqa: E402

# (name, N, H, W, Cin, Cout, k, stride, pad)
B = int(os.environ.get('BATCH', 16))
SHAPES = [
    ('stem7x7', B, 512, 512, 4, 64, 7, 2, 3),
    ('l1.1x1.64-64', B, 128, 128, 64, 64, 1, 1, 0),
    ('l1.3x3.64', B, 128, 128, 64, 64, 3, 1, 1),
    ('l1.1x1.64-256', B, 128, 128, 64, 256, 1, 1, 0),
    ('l1.1x1.256-64', B, 128, 128, 256, 64, 1, 1, 0),
    ('l2.3x3.128.s2', B, 128, 128, 128, 128, 3, 2, 1),
    ('l2.3x3.128', B, 64, 64, 128, 128, 3, 1, 1),
    ('l2.1x1.128-512', B, 64, 64, 128, 512, 1, 1, 0),
    ('l2.ds.256-512.s2', B, 128, 128, 256, 512, 1, 2, 0),
    ('l3.3x3.256', B, 32, 32, 256, 256, 3, 1, 1),
    ('l3.1x1.1024-256', B, 32, 32, 1024, 256, 1, 1, 0),
    ('l4.3x3.512', B, 16, 16, 512, 512, 3, 1, 1),
    ('l4.1x1.512-2048', B, 16, 16, 512, 2048, 1, 1, 0),
    ('fpn.3x3.256@128', B, 128, 128, 256, 256, 3, 1, 1),
    ('fpn.1x1.256@128', B, 128, 128, 256, 256, 1, 1, 0),
    ('fpn.3x3.256@16', B, 16, 16, 256, 256, 3, 1, 1),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    dev = torch.device('cuda:0')
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    print(f'{"layer":22s} {"GF":>8s} | {"fwd ms":>8s} {"TF/s":>6s} | {"dgrad ms":>8s} {"TF/s":>6s} | {"wgrad ms":>8s} {"TF/s":>6s}')
    tot = [0.0, 0.0, 0.0, 0.0]
    for name, n, h, w, cin, cout, k, s, p in SHAPES:
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        d = _C.ConvDesc(n, h, w, cin, ho, wo, cout, k, k, s, s, p, p, 1, 1)
        x = torch.randn(n, h, w, cin, device=dev)
        wt = torch.randn(cout, k, k, cin, device=dev) * 0.05
        y = torch.empty(n, ho, wo, cout, device=dev)
        dy = torch.randn(n, ho, wo, cout, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(wt)
        wtt = torch.empty(cin, k, k, cout, device=dev)
        ws_bytes = lib.evk_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
        ws = workspace(dev, ws_bytes)
        gf = 2.0 * n * ho * wo * cout * cin * k * k / 1e9
        tf = timeit(lambda: _C.call('evk_conv2d_fwd', ctypes.byref(d), x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), 0, st))
        _C.call('evk_conv2d_pack_dgrad_weight', ctypes.byref(d), wt.data_ptr(), wtt.data_ptr(), st)
        td = timeit(lambda: _C.call('evk_conv2d_dgrad', ctypes.byref(d), dy.data_ptr(), wtt.data_ptr(), None, dx.data_ptr(), st))
        tw = timeit(lambda: _C.call('evk_conv2d_wgrad', ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None,
                                    ws.data_ptr(), ws_bytes, st))
        print(f'{name:22s} {gf:8.1f} | {tf*1e3:8.3f} {gf/tf/1e3:6.1f} | {td*1e3:8.3f} {gf/td/1e3:6.1f} | {tw*1e3:8.3f} {gf/tw/1e3:6.1f}')
        tot[0] += gf; tot[1] += tf; tot[2] += td; tot[3] += tw
    print(f'{"TOTAL":22s} {tot[0]:8.1f} | {tot[1]*1e3:8.3f} {tot[0]/tot[1]/1e3:6.1f} | {tot[2]*1e3:8.3f} {tot[0]/tot[2]/1e3:6.1f} | {tot[3]*1e3:8.3f} {tot[0]/tot[3]/1e3:6.1f}')


if __name__ == '__main__':
    main()
