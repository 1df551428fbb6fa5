"""dev probe (GPU box): evk_bn_bwd / evk_bn_fwd on the big maps, to be run under rocprofv3 --kernel-trace --stats"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ever_amd import _C
dev = torch.device('cuda:0'); lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
for rows, c in ((262144, 256), (262144, 64), (65536, 512), (65536, 128), (16384, 1024), (16384, 256)):
    x = torch.randn(rows, c, device=dev); dy = torch.randn(rows, c, device=dev); dx = torch.empty_like(x)
    g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev); mean = x.mean(0); invstd = 1 / x.std(0)
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    wsb = lib.evk_bn_workspace_bytes(rows, c); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    bits = torch.zeros(int(lib.evk_absmax_words()), dtype=torch.int32, device=dev)
    for flags in (1, 3):
        for _ in range(6):
            if flags & 2: bits.zero_()
            _C.call('evk_bn_bwd', dy.data_ptr(), x.data_ptr(), None, g.data_ptr(), b.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                    dx.data_ptr(), None, dg.data_ptr(), db.data_ptr(), rows, c, flags, 1, ws.data_ptr(), wsb, bits.data_ptr(), st)
    torch.cuda.synchronize()
