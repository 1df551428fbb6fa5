"""dev probe (GPU box): the stem's weight gradient (space-to-depth 4x4x16 -> 64 on 16 x 259 x 259) with fp32 / packed operands and
under the planner's knobs — it runs at the very end of the backward pass with nothing beside it."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('EVK_TUNE', '1')
from ever_amd import _C
dev = torch.device('cuda:0'); lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
n, h, w, cout = 16, 512, 512, 64
d = _C.ConvDesc(n, h // 2 + 3, w // 2 + 3, 16, h // 2, w // 2, cout, 4, 4, 1, 1, 0, 0, 1, 1)
g = torch.Generator().manual_seed(0)
xs = torch.randn(n, h // 2 + 3, w // 2 + 3, 16, generator=g).to(dev)
dy = torch.randn(n, h // 2, w // 2, cout, generator=g).to(dev)
nw = int(lib.evk_absmax_words()); aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=dev)
bx, bd = (torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(2))
_C.call('evk_absmax', xs.data_ptr(), xs.numel(), bx.data_ptr(), aws.data_ptr(), st)
_C.call('evk_absmax', dy.data_ptr(), dy.numel(), bd.data_ptr(), aws.data_ptr(), st)
xp, dp = torch.empty_like(xs), torch.empty_like(dy)
_C.call('evk_pack_f16x2', xs.data_ptr(), xs.numel(), bx.data_ptr(), xp.data_ptr(), st)
_C.call('evk_pack_f16x2', dy.data_ptr(), dy.numel(), bd.data_ptr(), dp.data_ptr(), st)
wsb = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
dw = torch.empty(cout, 4, 4, 16, device=dev)
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
ref = None
for knobs in ({}, {'EVK_WG_MINCHUNK': '1024'}, {}, {'EVK_WG_MINCHUNK': '1024'}, {'EVK_WG_MINCHUNK': '4096'}, {}):
    for k in ('EVK_WG_ROUNDS', 'EVK_WG_MINCHUNK', 'EVK_WG_WS_MINCOUT'):
        if k in knobs: os.environ[k] = knobs[k]
        else: os.environ.pop(k, None)
    for name, (xa, da, fl) in {'fp32 / fp32': (xs, dy, 0), 'fp32 / packed dy': (xs, dp, 4), 'packed / packed': (xp, dp, 6)}.items():
        fn = lambda: _C.call('evk_conv2d_wgrad_f16x2_ex', ctypes.byref(d), xa.data_ptr(), bx.data_ptr(), da.data_ptr(), bd.data_ptr(),
                             dw.data_ptr(), None, ws.data_ptr(), wsb, fl, st)
        t = timeit(fn)
        if ref is None: ref = dw.clone()
        err = float((dw - ref).abs().max() / ref.abs().max())
        print(f'{str(knobs):32s} {name:18s} {t:7.1f} us  (vs first {err:.1e})', flush=True)
