"""dev tool (GPU box): the f16x2 convolution arithmetic against the bf16x3 one through the C-ABI — accuracy vs an fp64
evaluation (forward and data gradient on a sub-batch, weight gradient on the full batch) and speed, per layer shape."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd import _C  # noqa: E402

B = int(os.environ.get('BATCH', 16))
SHAPES = [
    ('l1.1x1.64-256', B, 128, 128, 64, 256, 1, 1, 0),
    ('l1.1x1.256-64', B, 128, 128, 256, 64, 1, 1, 0),
    ('l1.3x3.64', B, 128, 128, 64, 64, 3, 1, 1),
    ('l2.3x3.128.s2', B, 128, 128, 128, 128, 3, 2, 1),
    ('l2.3x3.128', B, 64, 64, 128, 128, 3, 1, 1),
    ('l2.ds.256-512.s2', B, 128, 128, 256, 512, 1, 2, 0),
    ('l3.3x3.256', B, 32, 32, 256, 256, 3, 1, 1),
    ('l3.1x1.1024-256', B, 32, 32, 1024, 256, 1, 1, 0),
    ('l4.3x3.512', B, 16, 16, 512, 512, 3, 1, 1),
    ('l4.1x1.512-2048', B, 16, 16, 512, 2048, 1, 1, 0),
    ('fpn.3x3.256@128', B, 128, 128, 256, 256, 3, 1, 1),
    ('fpn.1x1.256@128', B, 128, 128, 256, 256, 1, 1, 0),
    ('odd.3x3.72-40', 3, 11, 13, 72, 40, 3, 1, 1),
]


def timeit(fn, iters=10):
    for _ in range(3):
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
    aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=dev)
    check = os.environ.get('CHECK', '1') == '1'
    scale_x = float(os.environ.get('XSCALE', 1.0))      # e.g. 1e-6: gradient-sized operands
    print(f'{"layer":18s} {"GF":>7s} | fwd x3 TF  h2 TF  err_x3   err_h2  | dgrad x3 TF  h2 TF  err_x3   err_h2 | wgrad x3 TF  h2 TF  err_x3   err_h2')
    t3 = th = 0.0
    for name, n, h, w, cin, cout, k, s, p in SHAPES:
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        d = _C.ConvDesc(n, h, w, cin, ho, wo, cout, k, k, s, s, p, p, 1, 1)
        g = torch.Generator(device='cpu').manual_seed(1)
        x = ((torch.randn(n, h, w, cin, generator=g) + 0.5) * scale_x).to(dev)
        wt = (torch.randn(cout, k, k, cin, generator=g) * 0.05).to(dev)
        dy = (torch.randn(n, ho, wo, cout, generator=g) * scale_x).to(dev)
        y3, yh = torch.empty(n, ho, wo, cout, device=dev), torch.empty(n, ho, wo, cout, device=dev)
        dx3, dxh = torch.empty_like(x), torch.empty_like(x)
        pf3 = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
        pd3 = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 1), dtype=torch.uint8, device=dev)
        pfh, pdh = torch.empty_like(pf3), torch.empty_like(pd3)
        nw = int(lib.evk_absmax_words())
        bx, bw, bdy = (torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(3))   # slot 0 = the maximum
        _C.call('evk_absmax', x.data_ptr(), x.numel(), bx.data_ptr(), aws.data_ptr(), st)
        _C.call('evk_absmax', wt.data_ptr(), wt.numel(), bw.data_ptr(), aws.data_ptr(), st)
        _C.call('evk_absmax', dy.data_ptr(), dy.numel(), bdy.data_ptr(), aws.data_ptr(), st)
        torch.cuda.synchronize()
        ref_bits = [int(t.abs().max().view(torch.int32)) for t in (x, wt, dy)]
        got = [int(b.view(64, nw // 64)[:, 0].max()) for b in (bx, bw, bdy)]
        assert got == ref_bits, (got, ref_bits)
        _C.call('evk_conv2d_split_weight', ctypes.byref(d), wt.data_ptr(), 0, pf3.data_ptr(), st)
        _C.call('evk_conv2d_split_weight', ctypes.byref(d), wt.data_ptr(), 1, pd3.data_ptr(), st)
        _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 0, pfh.data_ptr(), bw.data_ptr(), st)
        _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 1, pdh.data_ptr(), bw.data_ptr(), st)
        zero = ctypes.c_int32(0)
        f3 = lambda: _C.call('evk_conv2d_fwd_x3', ctypes.byref(d), x.data_ptr(), pf3.data_ptr(), None, y3.data_ptr(), 0, st)
        fh = lambda: _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), x.data_ptr(), bx.data_ptr(), pfh.data_ptr(), bw.data_ptr(),
                             None, None, yh.data_ptr(), 0, None, 0, ctypes.byref(zero), None, st)
        g3 = lambda: _C.call('evk_conv2d_dgrad_x3', ctypes.byref(d), dy.data_ptr(), pd3.data_ptr(), None, dx3.data_ptr(), st)
        gh = lambda: _C.call('evk_conv2d_dgrad_f16x2', ctypes.byref(d), dy.data_ptr(), bdy.data_ptr(), pdh.data_ptr(), bw.data_ptr(),
                             None, dxh.data_ptr(), None, st)
        dw3, dwh = torch.empty_like(wt), torch.empty_like(wt)
        wsb = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d))
        wsp = torch.empty(wsb, dtype=torch.uint8, device=dev)
        w3 = lambda: _C.call('evk_conv2d_wgrad_x3', ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw3.data_ptr(), None, wsp.data_ptr(), wsb, st)
        wh = lambda: _C.call('evk_conv2d_wgrad_f16x2', ctypes.byref(d), x.data_ptr(), bx.data_ptr(), dy.data_ptr(), bdy.data_ptr(),
                             dwh.data_ptr(), None, wsp.data_ptr(), wsb, st)
        am = lambda: _C.call('evk_absmax', x.data_ptr(), x.numel(), bx.data_ptr(), aws.data_ptr(), st)
        gf = 2.0 * n * ho * wo * cout * cin * k * k / 1e9
        ta, tb, tc, td, te, tf, tm = timeit(f3), timeit(fh), timeit(g3), timeit(gh), timeit(w3), timeit(wh), timeit(am)
        t3 += ta + tc + te
        th += tb + td + tf
        e = ['-'] * 6
        if check:
            nb = 1
            xc = x[:nb].cpu().double().permute(0, 3, 1, 2)
            wc = wt.cpu().double().permute(0, 3, 1, 2)
            yr = torch.nn.functional.conv2d(xc, wc, None, s, p).permute(0, 2, 3, 1)
            dxr = torch.nn.grad.conv2d_input(xc.shape, wc, dy[:nb].cpu().double().permute(0, 3, 1, 2), s, p).permute(0, 2, 3, 1)
            wr = torch.nn.grad.conv2d_weight(x.cpu().double().permute(0, 3, 1, 2), (cout, cin, k, k),
                                             dy.cpu().double().permute(0, 3, 1, 2), s, p).permute(0, 2, 3, 1)
            rel = lambda a, b: float((a.cpu().double() - b).abs().max() / b.abs().max())
            e = [f'{rel(y3[:nb], yr):.1e}', f'{rel(yh[:nb], yr):.1e}', f'{rel(dx3[:nb], dxr):.1e}', f'{rel(dxh[:nb], dxr):.1e}',
                 f'{rel(dw3, wr):.1e}', f'{rel(dwh, wr):.1e}']
        print(f'{name:18s} {gf:7.1f} | {gf/ta/1e3:7.1f} {gf/tb/1e3:7.1f}  {e[0]:>8s} {e[1]:>8s} | {gf/tc/1e3:7.1f} {gf/td/1e3:7.1f}  {e[2]:>8s} {e[3]:>8s}'
              f' | {gf/te/1e3:7.1f} {gf/tf/1e3:7.1f}  {e[4]:>8s} {e[5]:>8s} | absmax(x) {tm*1e6:6.1f} us {x.numel()*4/tm/1e12:5.2f} TB/s')
    print(f'total fwd+dgrad+wgrad: x3 {t3*1e3:.2f} ms, f16x2 {th*1e3:.2f} ms')


if __name__ == '__main__':
    main()
