"""dev tool (GPU box): launch ONE one-tap convolution form a few times (for rocprofv3 --pmc runs).
usage: EVK_TUNE=1 EVK_X3_FORCE=q128 [EVK_LIB=<variant built with -DEVK_C1_DMA_ABL=n>] python tools/one_c1.py h cin cout packed stats [iters]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('EVK_TUNE', '1')
from ever_amd import _C
h, cin, cout, packed, stats = map(int, sys.argv[1:6]); iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device('cuda:0'); lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
B = 16
nw = int(lib.evk_absmax_words())
aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=dev)
d = _C.ConvDesc(B, h, h, cin, h, h, cout, 1, 1, 1, 1, 0, 0, 1, 1)
g = torch.Generator().manual_seed(1)
x = (torch.randn(B, h, h, cin, generator=g) + 0.5).to(dev)
wt = (torch.randn(cout, 1, 1, cin, generator=g) * 0.05).to(dev)
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
out = torch.empty(B, h, h, cout, device=dev)
cap = int(lib.evk_conv2d_stats_max_parts(ctypes.byref(d))) if stats else 0
parts = torch.empty(max(cap, 1) * 3 * cout, device=dev)
npart = ctypes.c_int32(0)
for _ in range(iters):
    _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), src.data_ptr(), bits[0].data_ptr(), planes.data_ptr(), bits[1].data_ptr(),
            None, None, out.data_ptr(), flags, parts.data_ptr() if cap else None, cap, ctypes.byref(npart), None, st)
torch.cuda.synchronize()
