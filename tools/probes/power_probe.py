"""dev probe (GPU box): is the big 3x3 halo layer clock / power bound?  The same launch (3x3x256 @128^2 x16, f16x2) on random
operands, on an all-zero activation tensor and on all-zero weights: zeros toggle no matrix-pipe or LDS data lines, the chip
holds a higher clock under its power cap (MI355X_MICROARCH.md, DVFS give-back)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
for (h, cin, cout, k) in ((128, 256, 256, 3), (64, 256, 256, 3), (32, 1024, 256, 1), (128, 256, 256, 1)):
    d = _C.ConvDesc(B, h, h, cin, h, h, cout, k, k, 1, 1, k // 2, k // 2, 1, 1)
    g = torch.Generator().manual_seed(1)
    row = []
    for xmode, wmode in (('rand', 'rand'), ('zero', 'rand'), ('rand', 'zero'), ('const', 'const')):
        x = torch.randn(B, h, h, cin, generator=g).to(dev)
        wt = (torch.randn(cout, k, k, cin, generator=g) * 0.05).to(dev)
        bits = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(2)]
        # scales from the RANDOM tensors, then the data is replaced: the kernels run the same code on cheaper data
        _C.call('evk_absmax', x.data_ptr(), x.numel(), bits[0].data_ptr(), aws.data_ptr(), st)
        _C.call('evk_absmax', wt.data_ptr(), wt.numel(), bits[1].data_ptr(), aws.data_ptr(), st)
        if xmode == 'zero': x.zero_()
        if xmode == 'const': x.fill_(1.0)
        if wmode == 'zero': wt.zero_()
        if wmode == 'const': wt.fill_(0.03125)
        planes = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
        _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 0, planes.data_ptr(), bits[1].data_ptr(), st)
        out = torch.empty(B, h, h, cout, device=dev)
        npart = ctypes.c_int32(0)
        fn = lambda: _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), x.data_ptr(), bits[0].data_ptr(), planes.data_ptr(), bits[1].data_ptr(),
                             None, None, out.data_ptr(), 0, None, 0, ctypes.byref(npart), None, st)
        row.append(f'x={xmode} w={wmode}: {timeit(fn):.1f}')
    print(f'{k}x{k} {cin}->{cout} @{h}^2: ' + ' | '.join(row), flush=True)
