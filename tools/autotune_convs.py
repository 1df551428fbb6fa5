"""dev tool (GPU box): every distinct forward / data-gradient convolution problem of one FarSeg-R50 training step, timed
under each tile shape the split kernels are instantiated for (EVK_TUNE=1 + EVK_X3_FORCE / EVK_X3_HALO_FORCE), next to the
shape the dispatch heuristics pick.  Prints per problem: launches per step, default time, best forced time and which.

usage: EVK_TUNE=1 python tools/autotune_convs.py            (halo layers: halo shapes; others: generic shapes)
       EVK_TUNE=1 EVK_X3_HALO=0 python tools/autotune_convs.py   (3x3 layers on the generic kernels)
"""
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('EVK_TUNE', '1')
import ever_amd as er  # noqa: E402
from ever_amd import _C  # noqa: E402

FIELDS = [f[0] for f in _C.ConvDesc._fields_]
GENERIC = ['c128x128', 'c64x128', 'c128x64', 'c64x64', 'w256', 'w128', 'w64', 'd256', 'd128', 'd64', 'e128', 'e64', 'q128', 's128', 's64', 't128']
HALO = ['h64x8', 'm64x8', 'm64x16', 'h128x8', 'h128x16', 'm128x8', 'm128x16']
B = int(os.environ.get('BATCH', 16))


def record_problems():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    m = er.module.FarSeg(dict()).to(dev).train()
    x = torch.randn(B, 3, 512, 512, device=dev)
    y = (torch.rand(B, 512, 512, device=dev) > 0.5).long()
    probs = collections.Counter()
    orig = _C.call

    def spy(name, *args):
        if name in ('evk_conv2d_fwd_f16x2', 'evk_conv2d_dgrad_f16x2', 'evk_conv2d_dgrad_f16x2_ex', 'evk_conv2d_dgrad_f16x2_masked'):
            d = args[0]._obj
            key = tuple(getattr(d, f) for f in FIELDS)
            if name == 'evk_conv2d_fwd_f16x2':
                kind = 'fwd_stats' if args[9] else 'fwd'
            else:
                kind = 'dgrad_accum' if args[5] else 'dgrad'
            probs[(kind, key)] += 1
        return orig(name, *args)
    _C.call = spy
    try:
        sum(m(x, y).values()).backward()
        torch.cuda.synchronize()
    finally:
        _C.call = orig
    del m, x, y
    torch.cuda.empty_cache()
    return probs


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    dev = torch.device('cuda:0')
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    probs = record_problems()
    rows = []
    for (kind, key), count in sorted(probs.items(), key=lambda kv: -kv[1]):
        d = _C.ConvDesc(*key)
        n, h, w, cin, ho, wo, cout, kh, kw = key[:9]
        if os.environ.get('AUTOTUNE_ONLY') == 'k3' and kh != 3:
            continue
        if os.environ.get('AUTOTUNE_ONLY') == 'k1' and kh != 1:
            continue
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(n, h, w, cin, generator=g) + 0.5).to(dev)
        wt = (torch.randn(cout, kh, kw, cin, generator=g) * 0.05).to(dev)
        dy = torch.randn(n, ho, wo, cout, generator=g).to(dev)
        for_dgrad = 1 if kind.startswith('dgrad') else 0
        planes = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), for_dgrad), dtype=torch.uint8, device=dev)
        nw = int(lib.evk_absmax_words()); bits = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(2)]
        aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=dev)
        src = dy if for_dgrad else x
        _C.call('evk_absmax', src.data_ptr(), src.numel(), bits[0].data_ptr(), aws.data_ptr(), st)
        _C.call('evk_absmax', wt.data_ptr(), wt.numel(), bits[1].data_ptr(), aws.data_ptr(), st)
        _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), for_dgrad, planes.data_ptr(), bits[1].data_ptr(), st)
        npart = ctypes.c_int32(0)
        if kind in ('fwd', 'fwd_stats'):
            out = torch.empty(n, ho, wo, cout, device=dev)
            cap = int(lib.evk_conv2d_stats_max_parts(ctypes.byref(d))) if kind == 'fwd_stats' else 0
            parts = torch.empty(max(cap, 1) * 3 * cout, device=dev)
            fn = lambda: _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), x.data_ptr(), bits[0].data_ptr(), planes.data_ptr(),
                                 bits[1].data_ptr(), None, None, out.data_ptr(), 0, parts.data_ptr() if cap else None, cap,
                                 ctypes.byref(npart), None, st)
        else:
            out = torch.empty(n, h, w, cin, device=dev)
            acc = torch.randn(n, h, w, cin, device=dev) if kind == 'dgrad_accum' else None
            fn = lambda: _C.call('evk_conv2d_dgrad_f16x2', ctypes.byref(d), dy.data_ptr(), bits[0].data_ptr(), planes.data_ptr(),
                                 bits[1].data_ptr(), acc.data_ptr() if acc is not None else None, out.data_ptr(), None, st)
        gf = 2.0 * n * ho * wo * cout * cin * kh * kw / 1e9
        iters = 20 if gf < 50 else 8
        os.environ['EVK_X3_FORCE'] = ''
        os.environ['EVK_X3_HALO_FORCE'] = ''
        timeit(fn, iters)
        base = timeit(fn, iters)
        ref = out.clone()
        res = {}
        halo_layer = kh == 3 and os.environ.get('EVK_X3_HALO', '1') != '0' and key[9] == 1 and key[13] == 1
        for cfg in (HALO if halo_layer else GENERIC):
            os.environ['EVK_X3_HALO_FORCE' if halo_layer else 'EVK_X3_FORCE'] = cfg
            try:
                t = timeit(fn, iters)
                err = float((out - ref).abs().max() / (ref.abs().max() + 1e-30))
                res[cfg] = t if err < 1e-4 else float('inf')
            except Exception:   # a shape this tile cannot run
                res[cfg] = float('inf')
        os.environ['EVK_X3_FORCE'] = ''
        os.environ['EVK_X3_HALO_FORCE'] = ''
        base = min(base, timeit(fn, iters))     # the first timing of a problem runs on cold clocks / pages
        best = min(res, key=res.get)
        rows.append((count * base, count, kind, key, gf, base, best, res[best], res))
        del x, wt, dy, out, planes
    rows.sort(key=lambda r: -r[0])
    tot_base = sum(r[0] for r in rows)
    tot_best = sum(r[1] * min(r[5], r[7]) for r in rows)
    print(f'{len(rows)} distinct problems, {sum(r[1] for r in rows)} launches; default {tot_base/1e3:.2f} ms, best-of {tot_best/1e3:.2f} ms')
    for tb, count, kind, key, gf, base, best, tbest, res in rows:
        n, h, w, cin, ho, wo, cout, kh, kw, sh = key[:10]
        gain = (base - tbest) * count
        alls = ' '.join(f'{k}={v:.0f}' for k, v in res.items() if v != float('inf'))
        print(f'{kind:11s} x{count:2d} {cin:4d}->{cout:4d} k{kh} s{sh} {h:3d}x{w:<3d} {gf:7.1f} GF  default {base:7.1f} us ({gf/base*1e3:6.1f} TF)'
              f'  best {best:8s} {tbest:7.1f} us  gain/step {gain:7.1f} us | {alls}')


if __name__ == '__main__':
    main()
