"""dev tool (GPU box): the LDS-DMA one-tap kernel (csrc/conv1x1_dma.hip) against the dispatch default on the FarSeg 1x1
shapes, fp32 and packed activation operand, plus its ablations (-DEVK_C1_DMA_ABL, tools/build_variant.sh: 1 no activation DMA, 2 no weight DMA,
4 no compute, 8 no stores).  usage: EVK_TUNE=1 python tools/ab_c1dma.py [ablate]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('EVK_TUNE', '1')
from ever_amd import _C
dev = torch.device('cuda:0'); lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
B = 16
nw = int(lib.evk_absmax_words())
aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=dev)


def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


def problem(h, cin, cout, packed, stats):
    d = _C.ConvDesc(B, h, h, cin, h, h, cout, 1, 1, 1, 1, 0, 0, 1, 1)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(B, h, h, cin, generator=g) + 0.5).to(dev)
    wt = (torch.randn(cout, 1, 1, cin, generator=g) * 0.05).to(dev)
    bits = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(2)]
    _C.call('evk_absmax', x.data_ptr(), x.numel(), bits[0].data_ptr(), aws.data_ptr(), st)
    _C.call('evk_absmax', wt.data_ptr(), wt.numel(), bits[1].data_ptr(), aws.data_ptr(), st)
    planes = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
    _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 0, planes.data_ptr(), bits[1].data_ptr(), st)
    src = x
    flags = 0
    if packed:
        src = torch.empty_like(x)
        _C.call('evk_pack_f16x2', x.data_ptr(), x.numel(), bits[0].data_ptr(), src.data_ptr(), st)
        flags = 2   # EVK_CONV_X_PACKED
    out = torch.empty(B, h, h, cout, device=dev)
    cap = int(lib.evk_conv2d_stats_max_parts(ctypes.byref(d))) if stats else 0
    parts = torch.empty(max(cap, 1) * 3 * cout, device=dev)
    npart = ctypes.c_int32(0)
    fn = lambda: _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), src.data_ptr(), bits[0].data_ptr(), planes.data_ptr(),
                         bits[1].data_ptr(), None, None, out.data_ptr(), flags, parts.data_ptr() if cap else None, cap,
                         ctypes.byref(npart), None, st)
    return fn, out, (x, wt, src, planes, parts, bits)


SHAPES = [(128, 64, 256), (128, 256, 64), (128, 256, 256), (128, 256, 128), (64, 128, 512), (64, 512, 128), (64, 512, 256),
          (64, 256, 256), (32, 256, 1024), (32, 1024, 256), (32, 1024, 512), (16, 512, 2048), (16, 2048, 512)]


def main():
    # (the kernel's ablations are compile-time since round 6: tools/build_variant.sh NAME conv1x1_dma.hip -DEVK_C1_DMA_ABL=n, then
    # EVK_LIB=<that build> python tools/ab_c1dma.py)
    tot = {}
    for (h, ci, co) in SHAPES:
        for packed in (0, 1):
            for stats in (0, 1):
                fn, out, keep = problem(h, ci, co, packed, stats)
                os.environ['EVK_X3_FORCE'] = ''
                timeit(fn, 5)
                base = timeit(fn)
                ref = out.clone()
                res = {}
                for cfg in ('d256', 'd128', 'd64', 'e128', 'e64'):
                    os.environ['EVK_X3_FORCE'] = cfg
                    t = timeit(fn)
                    err = float((out - ref).abs().max() / ref.abs().max())
                    res[cfg] = (t, err)
                os.environ['EVK_X3_FORCE'] = ''
                gf = 2.0 * B * h * h * ci * co / 1e9
                mb = B * h * h * (ci + co) * 4 / 1e6
                best = min(res, key=lambda k: res[k][0])
                key = (packed, stats)
                a, b = tot.get(key, (0.0, 0.0))
                tot[key] = (a + base, b + min(base, res[best][0]))
                print(f'{ci:4d}->{co:4d} @{h:3d}^2 pk={packed} st={stats} {gf:5.1f} GF {mb:6.1f} MB  default {base:6.1f} us ({mb / base:4.2f} TB/s) | '
                      + ' '.join(f'{k}={v[0]:.0f}' + ('' if v[1] < 1e-5 else f'(err {v[1]:.1e})') for k, v in res.items()), flush=True)
    print('sums (default, best-of):', {k: (round(v[0]), round(v[1])) for k, v in tot.items()})


main()
