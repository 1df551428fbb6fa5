"""dev tool (GPU box): the 3x3 / stride-1 layers of the step through the f16x2 entry points — time and error against an fp64
convolution — under whatever EVK_WINO says (0 = direct halo kernel, 1 = default dispatch, 2 = Winograd wherever it applies).
Run it once per setting inside ONE gpurun call: boxes differ by +-3 %."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd import _C  # noqa: E402

B = int(os.environ.get('BATCH', 16))
SHAPES = [
    ('fpn.256-256@128', B, 128, 128, 256, 256),
    ('dec.256-128@128', B, 128, 128, 256, 128),
    ('l1.64-64@128', B, 128, 128, 64, 64),
    ('fpn.256-256@64', B, 64, 64, 256, 256),
    ('dec.256-128@64', B, 64, 64, 256, 128),
    ('l2.128-128@64', B, 64, 64, 128, 128),
    ('l3.256-256@32', B, 32, 32, 256, 256),
    ('dec.256-128@32', B, 32, 32, 256, 128),
    ('l4.512-512@16', B, 16, 16, 512, 512),
    ('odd.72-136@40x24', 3, 40, 24, 72, 136),
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
    only = os.environ.get('ONLY')
    print(f'EVK_WINO={os.environ.get("EVK_WINO", "(default)")}')
    print(f'{"layer":20s} {"GF":>7s} |  fwd us   TF/s   err    |  dgrad us  TF/s   err')
    for name, n, h, w, cin, cout in SHAPES:
        if only and only not in name:
            continue
        d = _C.ConvDesc(n, h, w, cin, h, w, cout, 3, 3, 1, 1, 1, 1, 1, 1)
        g = torch.Generator(device='cpu').manual_seed(1)
        x = (torch.randn(n, h, w, cin, generator=g) + 0.5).to(dev)
        wt = (torch.randn(cout, 3, 3, cin, generator=g) * 0.05).to(dev)
        dy = torch.randn(n, h, w, cout, generator=g).to(dev)
        yh, dxh = torch.empty(n, h, w, cout, device=dev), torch.empty_like(x)
        pfh = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
        pdh = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 1), dtype=torch.uint8, device=dev)
        nw = int(lib.evk_absmax_words())
        bx, bw, bdy = (torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(3))
        _C.call('evk_absmax', x.data_ptr(), x.numel(), bx.data_ptr(), aws.data_ptr(), st)
        _C.call('evk_absmax', wt.data_ptr(), wt.numel(), bw.data_ptr(), aws.data_ptr(), st)
        _C.call('evk_absmax', dy.data_ptr(), dy.numel(), bdy.data_ptr(), aws.data_ptr(), st)
        _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 0, pfh.data_ptr(), bw.data_ptr(), st)
        _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 1, pdh.data_ptr(), bw.data_ptr(), st)
        zero = ctypes.c_int32(0)
        fh = lambda: _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), x.data_ptr(), bx.data_ptr(), pfh.data_ptr(), bw.data_ptr(),
                             None, None, yh.data_ptr(), 0, None, 0, ctypes.byref(zero), None, st)
        gh = lambda: _C.call('evk_conv2d_dgrad_f16x2', ctypes.byref(d), dy.data_ptr(), bdy.data_ptr(), pdh.data_ptr(), bw.data_ptr(),
                             None, dxh.data_ptr(), None, st)
        gf = 2.0 * n * h * w * cout * cin * 9 / 1e9
        tf, tg = timeit(fh), timeit(gh)
        nb = 1
        xc = x[:nb].cpu().double().permute(0, 3, 1, 2)
        wc = wt.cpu().double().permute(0, 3, 1, 2)
        yr = torch.nn.functional.conv2d(xc, wc, None, 1, 1).permute(0, 2, 3, 1)
        dxr = torch.nn.grad.conv2d_input(xc.shape, wc, dy[:nb].cpu().double().permute(0, 3, 1, 2), 1, 1).permute(0, 2, 3, 1)
        rel = lambda a, b: float((a.cpu().double() - b).abs().max() / b.abs().max())
        print(f'{name:20s} {gf:7.1f} | {tf*1e6:7.1f} {gf/tf/1e3:6.1f} {rel(yh[:nb], yr):.1e} | {tg*1e6:7.1f} {gf/tg/1e3:6.1f} {rel(dxh[:nb], dxr):.1e}',
              flush=True)


if __name__ == '__main__':
    main()
