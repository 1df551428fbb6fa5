"""dev tool: launch one conv kernel a few times (for rocprofv3 --pmc runs). args: which h w cin cout k s p [iters]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd import _C
from ever_amd.hip.workspace import workspace
which = sys.argv[1]; h, w, cin, cout, k, s, p = map(int, sys.argv[2:9]); iters = int(sys.argv[9]) if len(sys.argv) > 9 else 3
B = 16; dev = torch.device('cuda:0'); lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
ho, wo = (h+2*p-k)//s+1, (w+2*p-k)//s+1
d = _C.ConvDesc(B,h,w,cin,ho,wo,cout,k,k,s,s,p,p,1,1)
x = torch.randn(B,h,w,cin,device=dev); dy = torch.randn(B,ho,wo,cout,device=dev)
wt = torch.randn(cout,k,k,cin,device=dev)*0.05; dw = torch.empty(cout,k,k,cin,device=dev)
y = torch.empty(B,ho,wo,cout,device=dev); dx = torch.empty_like(x); wtt = torch.empty(cin,k,k,cout,device=dev)
ws_b = max(lib.evk_conv2d_wgrad_workspace_bytes(ctypes.byref(d)), lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d))); ws = workspace(dev, ws_b)
pf = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
pd = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 1), dtype=torch.uint8, device=dev)
_C.call('evk_conv2d_split_weight', ctypes.byref(d), wt.data_ptr(), 0, pf.data_ptr(), st)
_C.call('evk_conv2d_split_weight', ctypes.byref(d), wt.data_ptr(), 1, pd.data_ptr(), st)
for _ in range(iters):
    if which == 'fwd': _C.call('evk_conv2d_fwd', ctypes.byref(d), x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), 0, st)
    elif which == 'fwd3': _C.call('evk_conv2d_fwd_x3', ctypes.byref(d), x.data_ptr(), pf.data_ptr(), None, y.data_ptr(), 0, st)
    elif which == 'dgrad3': _C.call('evk_conv2d_dgrad_x3', ctypes.byref(d), dy.data_ptr(), pd.data_ptr(), None, dx.data_ptr(), st)
    elif which == 'wgrad3': _C.call('evk_conv2d_wgrad_x3', ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), ws_b, st)
    elif which == 'wgrad': _C.call('evk_conv2d_wgrad', ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), ws_b, st)
    else:
        _C.call('evk_conv2d_pack_dgrad_weight', ctypes.byref(d), wt.data_ptr(), wtt.data_ptr(), st)
        _C.call('evk_conv2d_dgrad', ctypes.byref(d), dy.data_ptr(), wtt.data_ptr(), None, dx.data_ptr(), st)
torch.cuda.synchronize()
