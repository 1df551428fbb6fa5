"""dev tool (GPU box): the planar-operand weight gradient (csrc/conv_wgrad_tr.hip: DMA + transposing LDS reads) against
the packed-operand wave-specialised kernel on the layer shapes of the FarSeg-R50 step: time, and the difference of the
results (same operands, same scales, same split-K plan where both take the 128 x 256 tile).
usage: [EVK_LIB=<variant built with -DEVK_WG_ABL=bits>] python tools/ab_wgrad_tr.py [quick]"""
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
# (h, cin, cout, k, stride)
SHAPES = [(128, 256, 256, 3, 1), (64, 256, 256, 3, 1), (32, 256, 256, 3, 1), (16, 512, 512, 3, 1),
          (128, 64, 256, 1, 1), (128, 256, 64, 1, 1), (64, 128, 512, 1, 1), (64, 512, 128, 1, 1), (32, 256, 1024, 1, 1),
          (32, 1024, 256, 1, 1), (16, 2048, 512, 1, 1), (16, 512, 2048, 1, 1), (128, 256, 256, 1, 1), (64, 128, 128, 3, 1),
          (128, 64, 64, 3, 1), (64, 256, 512, 1, 2), (128, 128, 128, 3, 2)]
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    SHAPES = SHAPES[:3] + [(24, 64, 128, 3, 1), (20, 128, 192, 3, 1), (32, 128, 192, 1, 2)]
tot_p = tot_t = 0.0
for (h, cin, cout, k, s) in SHAPES:
    pad = k // 2
    ho = (h + 2 * pad - k) // s + 1
    d = _C.ConvDesc(16, h, h, cin, ho, ho, cout, k, k, s, s, pad, pad, 1, 1)
    g = torch.Generator(device=dev).manual_seed(h * 1000 + cin)
    x = torch.randn(16, h, h, cin, device=dev, generator=g); dy = torch.randn(16, ho, ho, cout, device=dev, generator=g)
    dw = torch.empty(cout, k, k, cin, device=dev); ref = torch.empty_like(dw)
    nw = int(lib.evk_absmax_words()); bits = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(2)]
    _C.call('evk_absmax', x.data_ptr(), x.numel(), bits[0].data_ptr(), aws.data_ptr(), st)
    _C.call('evk_absmax', dy.data_ptr(), dy.numel(), bits[1].data_ptr(), aws.data_ptr(), st)
    wsb = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d)); wsp = torch.empty(wsb, dtype=torch.uint8, device=dev)
    xp, dyp = torch.empty_like(x), torch.empty_like(dy)
    _C.call('evk_pack_f16x2', x.data_ptr(), x.numel(), bits[0].data_ptr(), xp.data_ptr(), st)
    _C.call('evk_pack_f16x2', dy.data_ptr(), dy.numel(), bits[1].data_ptr(), dyp.data_ptr(), st)
    xq, dyq = torch.empty_like(x), torch.empty_like(dy)
    _C.call('evk_pack_planar_f16x2', x.data_ptr(), x.numel(), bits[0].data_ptr(), xq.data_ptr(), st)
    _C.call('evk_pack_planar_f16x2', dy.data_ptr(), dy.numel(), bits[1].data_ptr(), dyq.data_ptr(), st)
    back = torch.empty_like(x)
    _C.call('evk_unpack_planar_f16x2', xq.data_ptr(), x.numel(), bits[0].data_ptr(), back.data_ptr(), st)
    rt = ((back - x).abs().max() / x.abs().max()).item()
    fp = lambda: _C.call('evk_conv2d_wgrad_f16x2_ex', ctypes.byref(d), xp.data_ptr(), bits[0].data_ptr(), dyp.data_ptr(),
                         bits[1].data_ptr(), ref.data_ptr(), None, wsp.data_ptr(), wsb, 6, st)
    ft = lambda: _C.call('evk_conv2d_wgrad_f16x2_ex', ctypes.byref(d), xq.data_ptr(), bits[0].data_ptr(), dyq.data_ptr(),
                         bits[1].data_ptr(), dw.data_ptr(), None, wsp.data_ptr(), wsb, 24, st)
    tp = timeit(fp); tt = timeit(ft)
    torch.cuda.synchronize()
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    # fp64 check of a slice (one output channel block) on small problems only
    gf = 2.0 * 16 * ho * ho * cout * cin * k * k / 1e9
    tot_p += tp; tot_t += tt
    print(f'{k}x{k} s{s} {cin:4d}->{cout:4d} @{h:3d}: packed ws {tp:7.1f} us ({gf / tp * 1e3:6.1f} TF)   planar tr {tt:7.1f} us ({gf / tt * 1e3:6.1f} TF)'
          f'   x{tp / tt:4.2f}   max rel diff {err:.1e}   planar round trip {rt:.1e}', flush=True)
print(f'EVK_LIB={os.environ.get("EVK_LIB", "(default build)")}  sum: packed {tot_p:.0f} us, planar {tot_t:.0f} us')
