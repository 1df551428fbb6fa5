"""dev probe (GPU box): evk_bn_bwd timings at the small-map shapes (EVK_BN_FUSED=0/1 A/B)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ever_amd import _C
from ever_amd.hip.functional import workspace
dev = torch.device('cuda:0'); lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
out = []
for rows, c in ((4096, 512), (16384, 256), (4096, 2048), (65536, 128), (16384, 512), (1024, 512), (16384, 1024)):
    x = torch.randn(rows, c, device=dev); dy = torch.randn(rows, c, device=dev); dx = torch.empty_like(x)
    g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev); mean = x.mean(0); invstd = 1 / x.std(0)
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    wsb = lib.evk_bn_workspace_bytes(rows, c); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    bits = torch.zeros(int(lib.evk_absmax_words()), dtype=torch.int32, device=dev)
    for flags in (1, 3):
        def fn():
            if flags & 2: bits.zero_()
            _C.call('evk_bn_bwd', dy.data_ptr(), x.data_ptr(), None, g.data_ptr(), b.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                    dx.data_ptr(), None, dg.data_ptr(), db.data_ptr(), rows, c, flags, 1, ws.data_ptr(), wsb, bits.data_ptr(), st)
        t = timeit(fn)
        out.append(f'{rows}x{c} {"pk" if flags & 2 else "f32"} {t:6.1f} us {rows * c * 4 * 5 / t / 1e6:5.2f} TB/s(5u)')
print('EVK_BN_FUSED=' + os.environ.get('EVK_BN_FUSED', '1'), ' | '.join(out))
