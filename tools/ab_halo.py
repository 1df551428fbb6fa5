"""dev tool (GPU box): the 3x3 halo kernel on the FarSeg shapes with its ablations (EVK_TUNE=1, EVK_HALO_DBG: 1 = staging
waves idle, 2 = matrix waves idle).  usage: python tools/ab_halo.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('EVK_TUNE', '1')
from ever_amd import _C
dev = torch.device('cuda:0'); lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
B = 16
nw = int(lib.evk_absmax_words())
aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=dev)


def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


if len(sys.argv) > 1 and sys.argv[1] == 'variants':
    # every tile form on the narrow / small-map shapes (bottleneck conv2 layers), with and without the statistics epilogue
    for (h, c) in [(128, 64), (64, 128), (32, 256), (16, 512)]:
        for stats in (0, 1):
            d = _C.ConvDesc(B, h, h, c, h, h, c, 3, 3, 1, 1, 1, 1, 1, 1)
            g = torch.Generator().manual_seed(1)
            x = (torch.randn(B, h, h, c, generator=g) + 0.5).to(dev)
            wt = (torch.randn(c, 3, 3, c, generator=g) * 0.05).to(dev)
            bits = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(2)]
            _C.call('evk_absmax', x.data_ptr(), x.numel(), bits[0].data_ptr(), aws.data_ptr(), st)
            _C.call('evk_absmax', wt.data_ptr(), wt.numel(), bits[1].data_ptr(), aws.data_ptr(), st)
            planes = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
            _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 0, planes.data_ptr(), bits[1].data_ptr(), st)
            src = torch.empty_like(x)
            _C.call('evk_pack_f16x2', x.data_ptr(), x.numel(), bits[0].data_ptr(), src.data_ptr(), st)
            out = torch.empty(B, h, h, c, device=dev)
            cap = int(lib.evk_conv2d_stats_max_parts(ctypes.byref(d))) if stats else 0
            parts = torch.empty(max(cap, 1) * 3 * c, device=dev)
            z = ctypes.c_int32(0)
            fn = lambda: _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), src.data_ptr(), bits[0].data_ptr(), planes.data_ptr(),
                                 bits[1].data_ptr(), None, None, out.data_ptr(), 2, parts.data_ptr() if cap else None, cap,
                                 ctypes.byref(z), None, st)
            os.environ['EVK_X3_HALO_FORCE'] = ''
            fn(); ref = out.clone()
            row = []
            for force in ('', 'h64x8', 'm64x8', 'h64x16', 'm64x16', 'h128x8', 'm128x8', 'h128x16', 'm128x16'):
                if force.endswith('x16') and h * h * B // 256 < 64:
                    continue
                os.environ['EVK_X3_HALO_FORCE'] = force
                t = timeit(fn)
                err = float((out - ref).abs().max() / ref.abs().max())
                row.append(f'{force or "default"}={t:.1f}' + ('' if err < 1e-5 else f'(err {err:.1e})'))
            os.environ['EVK_X3_HALO_FORCE'] = ''
            print(f'3x3x{c} @{h}^2 stats={stats}: ' + ' '.join(row), flush=True)
    sys.exit(0)
for (h, c, packed) in [(128, 256, 0), (128, 256, 1), (64, 256, 0), (64, 256, 1), (32, 256, 1), (16, 512, 1), (128, 64, 1), (64, 128, 1)]:
    d = _C.ConvDesc(B, h, h, c, h, h, c, 3, 3, 1, 1, 1, 1, 1, 1)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(B, h, h, c, generator=g) + 0.5).to(dev)
    wt = (torch.randn(c, 3, 3, c, generator=g) * 0.05).to(dev)
    bits = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(2)]
    _C.call('evk_absmax', x.data_ptr(), x.numel(), bits[0].data_ptr(), aws.data_ptr(), st)
    _C.call('evk_absmax', wt.data_ptr(), wt.numel(), bits[1].data_ptr(), aws.data_ptr(), st)
    planes = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
    _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 0, planes.data_ptr(), bits[1].data_ptr(), st)
    src, flags = x, 0
    if packed:
        src = torch.empty_like(x)
        _C.call('evk_pack_f16x2', x.data_ptr(), x.numel(), bits[0].data_ptr(), src.data_ptr(), st)
        flags = 2
    out = torch.empty(B, h, h, c, device=dev)
    z = ctypes.c_int32(0)
    fn = lambda: _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), src.data_ptr(), bits[0].data_ptr(), planes.data_ptr(),
                         bits[1].data_ptr(), None, None, out.data_ptr(), flags, None, 0, ctypes.byref(z), None, st)
    row = []
    for force in ('', 'h128x8'):
        os.environ['EVK_X3_HALO_FORCE'] = force
        for dbg in (0, 1, 2):
            os.environ['EVK_HALO_DBG'] = str(dbg)
            row.append(f'{force or "default"}/dbg{dbg}={timeit(fn):.0f}')
    os.environ['EVK_HALO_DBG'] = '0'
    os.environ['EVK_X3_HALO_FORCE'] = ''
    gf = 2.0 * B * h * h * c * c * 9 / 1e9
    print(f'3x3x{c} @{h}^2 packed={packed} {gf:6.1f} GF (mfma at 833 TF: {gf / 833.3 * 1e3:.0f} us): ' + ' '.join(row), flush=True)
