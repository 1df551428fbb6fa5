"""dev tool: bf16-split (x3) conv kernels vs the exact-fp32 MFMA kernels: accuracy against fp64 and speed."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd import _C  # noqa: E402

B = int(os.environ.get('BATCH', 16))
SHAPES = [
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


def timeit(fn, iters=10):
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
    _C.load()
    st = torch.cuda.current_stream().cuda_stream
    check = os.environ.get('CHECK', '1') == '1'
    print(f'{"layer":20s} {"GF":>7s} | fwd f32 TF   x3 TF  err32    err_x3  | dgrad f32 TF  x3 TF  err32    err_x3 | wgrad f32 TF  x3 TF  err32    err_x3')
    t32 = t3 = 0.0
    for name, n, h, w, cin, cout, k, s, p in SHAPES:
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        d = _C.ConvDesc(n, h, w, cin, ho, wo, cout, k, k, s, s, p, p, 1, 1)
        g = torch.Generator(device='cpu').manual_seed(1)
        x = (torch.randn(n, h, w, cin, generator=g) + 0.5).to(dev)
        wt = (torch.randn(cout, k, k, cin, generator=g) * 0.05).to(dev)
        dy = torch.randn(n, ho, wo, cout, generator=g).to(dev)
        y, y3 = torch.empty(n, ho, wo, cout, device=dev), torch.empty(n, ho, wo, cout, device=dev)
        dx, dx3 = torch.empty_like(x), torch.empty_like(x)
        wpk = torch.empty(cin * k * k * cout, device=dev)
        ws_f = torch.empty(_C.load().evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
        ws_d = torch.empty(_C.load().evk_conv2d_split_weight_bytes(ctypes.byref(d), 1), dtype=torch.uint8, device=dev)
        _C.call('evk_conv2d_split_weight', ctypes.byref(d), wt.data_ptr(), 0, ws_f.data_ptr(), st)
        _C.call('evk_conv2d_split_weight', ctypes.byref(d), wt.data_ptr(), 1, ws_d.data_ptr(), st)
        _C.call('evk_conv2d_pack_dgrad_weight', ctypes.byref(d), wt.data_ptr(), wpk.data_ptr(), st)
        f32 = lambda: _C.call('evk_conv2d_fwd', ctypes.byref(d), x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), 0, st)
        fx3 = lambda: _C.call('evk_conv2d_fwd_x3', ctypes.byref(d), x.data_ptr(), ws_f.data_ptr(), None, y3.data_ptr(), 0, st)
        d32 = lambda: _C.call('evk_conv2d_dgrad', ctypes.byref(d), dy.data_ptr(), wpk.data_ptr(), None, dx.data_ptr(), st)
        dx3f = lambda: _C.call('evk_conv2d_dgrad_x3', ctypes.byref(d), dy.data_ptr(), ws_d.data_ptr(), None, dx3.data_ptr(), st)
        lib = _C.load()
        dw, dw3 = torch.empty_like(wt), torch.empty_like(wt)
        wsb = max(lib.evk_conv2d_wgrad_workspace_bytes(ctypes.byref(d)), lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d)))
        wsp = torch.empty(wsb, dtype=torch.uint8, device=dev)
        w32 = lambda: _C.call('evk_conv2d_wgrad', ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, wsp.data_ptr(), wsb, st)
        wx3 = lambda: _C.call('evk_conv2d_wgrad_x3', ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw3.data_ptr(), None, wsp.data_ptr(), wsb, st)
        gf = 2.0 * n * ho * wo * cout * cin * k * k / 1e9
        ta, tb, tc, td = timeit(f32), timeit(fx3), timeit(d32), timeit(dx3f)
        te, tf = timeit(w32), timeit(wx3)
        t32 += ta + tc + te
        t3 += tb + td + tf
        ew = ['-', '-']
        if check:
            # weight gradient truth in fp64 on the GPU-computed operands of a sub-batch is not separable (sum over
            # the batch), so compare on the full batch through torch's CPU fp64 conv2d_weight
            wr = torch.nn.grad.conv2d_weight(x.cpu().double().permute(0, 3, 1, 2), (cout, cin, k, k),
                                             dy.cpu().double().permute(0, 3, 1, 2), s, p).permute(0, 2, 3, 1)
            relw = lambda a: float((a.cpu().double() - wr).abs().max() / wr.abs().max())
            ew = [f'{relw(dw):.1e}', f'{relw(dw3):.1e}']
        e = ['-'] * 4
        if check:
            # fp64 truth on a sub-batch (CPU)
            nb = 1
            xc = x[:nb].cpu().double().permute(0, 3, 1, 2)
            wc = wt.cpu().double().permute(0, 3, 1, 2)
            yr = torch.nn.functional.conv2d(xc, wc, None, s, p).permute(0, 2, 3, 1)
            dxr = torch.nn.grad.conv2d_input(xc.shape, wc, dy[:nb].cpu().double().permute(0, 3, 1, 2), s, p).permute(0, 2, 3, 1)
            rel = lambda a, b: float((a.cpu().double() - b).abs().max() / b.abs().max())
            e = [f'{rel(y[:nb], yr):.1e}', f'{rel(y3[:nb], yr):.1e}', f'{rel(dx[:nb], dxr):.1e}', f'{rel(dx3[:nb], dxr):.1e}']
        print(f'{name:20s} {gf:7.1f} | {gf/ta/1e3:7.1f} {gf/tb/1e3:7.1f}  {e[0]:>8s} {e[1]:>8s} | {gf/tc/1e3:7.1f} {gf/td/1e3:7.1f}  {e[2]:>8s} {e[3]:>8s}'
              f' | {gf/te/1e3:7.1f} {gf/tf/1e3:7.1f}  {ew[0]:>8s} {ew[1]:>8s}')
    print(f'total fwd+dgrad+wgrad: f32 {t32*1e3:.2f} ms, x3 {t3*1e3:.2f} ms')


if __name__ == '__main__':
    main()
