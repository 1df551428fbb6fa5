"""dev tool (GPU box): every distinct weight-gradient problem of one FarSeg-R50 training step under the planner's
knobs (EVK_TUNE=1: EVK_WG_WS, EVK_WG_ROUNDS, EVK_WG_MINCHUNK re-read per call), next to the default plan."""
import collections
import ctypes
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('EVK_TUNE', '1')
import ever_amd as er  # noqa: E402
from ever_amd import _C  # noqa: E402

FIELDS = [f[0] for f in _C.ConvDesc._fields_]
B = int(os.environ.get('BATCH', 16))
MODEL = os.environ.get('MODEL', 'farseg')     # 'freenet': the C5 scene (batch 1, 200 bands, 616 x 344)
KNOB_NAMES = ('EVK_WG_WS', 'EVK_WG_ROUNDS', 'EVK_WG_MINCHUNK', 'EVK_WG_WS_MINCOUT')
KNOBS = [dict(zip(KNOB_NAMES, v)) for v in itertools.product(('1', '0'), ('1', '2'), ('256', '1024'), ('128', '64'))
         if not (v[0] == '0' and v[3] == '64')]


def record():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    if MODEL == 'freenet':
        m = er.module.FreeNet(dict()).to(dev).train()
        x = torch.randn(1, 200, 616, 344, device=dev)
        y = torch.randint(0, 17, (1, 616, 344), device=dev)
    else:
        m = er.module.FarSeg(dict()).to(dev).train()
        x = torch.randn(B, 3, 512, 512, device=dev)
        y = (torch.rand(B, 512, 512, device=dev) > 0.5).long()
    probs = collections.Counter()
    orig = _C.call

    def spy(name, *args):
        if name in ('evk_conv2d_wgrad_x3', 'evk_conv2d_wgrad_f16x2', 'evk_conv2d_wgrad_f16x2_ex'):
            d = args[0]._obj
            probs[tuple(getattr(d, f) for f in FIELDS)] += 1
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
    return s.elapsed_time(e) / iters * 1e3


def setk(k):
    for n in KNOB_NAMES:
        os.environ[n] = k.get(n, '') if k else ''


def main():
    dev = torch.device('cuda:0')
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for key, count in record().items():
        d = _C.ConvDesc(*key)
        n, h, w, cin, ho, wo, cout, kh, kw = key[:9]
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(n, h, w, cin, generator=g) + 0.5).to(dev)
        dy = torch.randn(n, ho, wo, cout, generator=g).to(dev)
        dw = torch.empty(cout, kh, kw, cin, device=dev)
        gf = 2.0 * n * ho * wo * cout * cin * kh * kw / 1e9
        iters = 20 if gf < 50 else 8
        nw = int(lib.evk_absmax_words()); bits = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(2)]
        aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=dev)
        _C.call('evk_absmax', x.data_ptr(), x.numel(), bits[0].data_ptr(), aws.data_ptr(), st)
        _C.call('evk_absmax', dy.data_ptr(), dy.numel(), bits[1].data_ptr(), aws.data_ptr(), st)

        def run():
            wsb = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d))
            wsp = torch.empty(wsb, dtype=torch.uint8, device=dev)
            fn = lambda: _C.call('evk_conv2d_wgrad_f16x2', ctypes.byref(d), x.data_ptr(), bits[0].data_ptr(), dy.data_ptr(),
                                 bits[1].data_ptr(), dw.data_ptr(), None, wsp.data_ptr(), wsb, st)
            return timeit(fn, iters)
        setk(None)
        run()
        base = run()
        ref = dw.clone()
        res = {}
        for k in KNOBS:
            setk(k)
            try:
                t = run()
                err = float((dw - ref).abs().max() / (ref.abs().max() + 1e-30))
                res[tuple(k[n] for n in KNOB_NAMES)] = t if err < 1e-4 else float('inf')
            except Exception:
                pass
        setk(None)
        base = min(base, run())
        best = min(res, key=res.get)
        rows.append((count * base, count, key, gf, base, best, res[best], res))
        del x, dy, dw
    rows.sort(key=lambda r: -r[0])
    print(f'{len(rows)} problems; default {sum(r[0] for r in rows)/1e3:.2f} ms, best-of {sum(r[1]*min(r[4], r[6]) for r in rows)/1e3:.2f} ms')
    for tb, count, key, gf, base, best, tbest, res in rows:
        n, h, w, cin, ho, wo, cout, kh, kw, sh = key[:10]
        top = sorted(res.items(), key=lambda kv: kv[1])[:4]
        print(f'x{count:2d} {cin:4d}->{cout:4d} k{kh} s{sh} {h:3d}x{w:<3d} {gf:7.1f} GF default {base:7.1f} us ({gf/base*1e3:6.1f} TF) '
              f'best ws/rounds/minchunk/mincout {best} {tbest:7.1f} us gain/step {(base-tbest)*count:7.1f} | ' +
              ' '.join(f'{k}={v:.0f}' for k, v in top))


if __name__ == '__main__':
    main()
