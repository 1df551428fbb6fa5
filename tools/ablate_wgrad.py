"""dev tool (GPU box): the wave-specialised weight gradient under EVK_WG_DBG ablation bits (1 no global loads, 2 no
split / LDS writes, 4 no fragment reads / MFMAs, 8 no stores) for two 3x3 shapes, f16x2 and bf16x3.  Results of the
ablated runs are wrong by construction: timing only.  usage: EVK_WG_DBG=<bits> python tools/ablate_wgrad.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd import _C
dev = torch.device('cuda:0'); lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=dev)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
out = []
for (h, c) in ((128, 256), (64, 256), (32, 256)):
    d = _C.ConvDesc(16, h, h, c, h, h, c, 3, 3, 1, 1, 1, 1, 1, 1)
    x = torch.randn(16, h, h, c, device=dev); dy = torch.randn(16, h, h, c, device=dev)
    dw = torch.empty(c, 3, 3, c, device=dev)
    nw = int(lib.evk_absmax_words()); bits = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(2)]
    _C.call('evk_absmax', x.data_ptr(), x.numel(), bits[0].data_ptr(), aws.data_ptr(), st)
    _C.call('evk_absmax', dy.data_ptr(), dy.numel(), bits[1].data_ptr(), aws.data_ptr(), st)
    wsb = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d)); wsp = torch.empty(wsb, dtype=torch.uint8, device=dev)
    th = timeit(lambda: _C.call('evk_conv2d_wgrad_f16x2', ctypes.byref(d), x.data_ptr(), bits[0].data_ptr(), dy.data_ptr(),
                                bits[1].data_ptr(), dw.data_ptr(), None, wsp.data_ptr(), wsb, st))
    t3 = timeit(lambda: _C.call('evk_conv2d_wgrad_x3', ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None,
                                wsp.data_ptr(), wsb, st))
    # both operands packed (evk_pack_f16x2): EVK_WG_BIG=1 routes these to the 256x256 kernel
    xp, dyp = torch.empty_like(x), torch.empty_like(dy)
    _C.call('evk_pack_f16x2', x.data_ptr(), x.numel(), bits[0].data_ptr(), xp.data_ptr(), st)
    _C.call('evk_pack_f16x2', dy.data_ptr(), dy.numel(), bits[1].data_ptr(), dyp.data_ptr(), st)
    ref = dw.clone()
    _C.call('evk_conv2d_wgrad_f16x2', ctypes.byref(d), x.data_ptr(), bits[0].data_ptr(), dy.data_ptr(), bits[1].data_ptr(),
            ref.data_ptr(), None, wsp.data_ptr(), wsb, st)
    tp = timeit(lambda: _C.call('evk_conv2d_wgrad_f16x2_ex', ctypes.byref(d), xp.data_ptr(), bits[0].data_ptr(), dyp.data_ptr(),
                                bits[1].data_ptr(), dw.data_ptr(), None, wsp.data_ptr(), wsb, 6, st))
    torch.cuda.synchronize()
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    out.append(f'{c}@{h}: f16x2 {th:7.1f} us  packed {tp:7.1f} us (rel diff {err:.1e})  bf16x3 {t3:7.1f} us')
print('EVK_WG_DBG=' + os.environ.get('EVK_WG_DBG', '0'), ' | '.join(out))
