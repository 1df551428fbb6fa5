"""dev tool: wgrad-only timing across the model's layer shapes (with multiplicity) -> ms per step."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd import _C
from ever_amd.hip.workspace import workspace
B = 16
# (count, H, W, Cin, Cout, k, s, p)  FarSeg-R50 @512, all 85 convs
L = [(1,512,512,4,64,7,2,3),
 (1,128,128,64,64,1,1,0),(3,128,128,64,64,3,1,1),(4,128,128,64,256,1,1,0),(2,128,128,256,64,1,1,0),
 (1,128,128,256,128,1,1,0),(1,128,128,128,128,3,2,1),(4,64,64,128,512,1,1,0),(1,128,128,256,512,1,2,0),(3,64,64,512,128,1,1,0),(3,64,64,128,128,3,1,1),
 (1,64,64,512,256,1,1,0),(1,64,64,256,256,3,2,1),(6,32,32,256,1024,1,1,0),(1,64,64,512,1024,1,2,0),(5,32,32,1024,256,1,1,0),(5,32,32,256,256,3,1,1),
 (1,32,32,1024,512,1,1,0),(1,32,32,512,512,3,2,1),(3,16,16,512,2048,1,1,0),(1,32,32,1024,2048,1,2,0),(2,16,16,2048,512,1,1,0),(2,16,16,512,512,3,1,1),
 (1,128,128,256,256,1,1,0),(1,64,64,512,256,1,1,0),(1,32,32,1024,256,1,1,0),(1,16,16,2048,256,1,1,0),
 (2,128,128,256,256,3,1,1),(4,64,64,256,256,3,1,1),(3,32,32,256,256,3,1,1),(2,16,16,256,256,3,1,1),
 (2,128,128,256,256,1,1,0),(2,64,64,256,256,1,1,0),(2,32,32,256,256,1,1,0),(2,16,16,256,256,1,1,0),
 (4,1,1,2048,256,1,1,0),(4,1,1,256,256,1,1,0),(1,128,128,256,4,1,1,0)]
dev = torch.device('cuda:0'); lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
tot_t = tot_gf = 0; which = sys.argv[1] if len(sys.argv) > 1 else 'wgrad'
rows = []
for cnt, h, w, cin, cout, k, s, p in L:
    ho, wo = (h+2*p-k)//s+1, (w+2*p-k)//s+1
    d = _C.ConvDesc(B,h,w,cin,ho,wo,cout,k,k,s,s,p,p,1,1)
    x = torch.randn(B,h,w,cin,device=dev); dy = torch.randn(B,ho,wo,cout,device=dev)
    wt = torch.randn(cout,k,k,cin,device=dev)*0.05; dw = torch.empty(cout,k,k,cin,device=dev)
    y = torch.empty(B,ho,wo,cout,device=dev); dx = torch.empty_like(x); wtt = torch.empty(cin,k,k,cout,device=dev)
    ws_b = lib.evk_conv2d_wgrad_workspace_bytes(ctypes.byref(d)); ws = workspace(dev, ws_b)
    gf = 2.0*B*ho*wo*cout*cin*k*k/1e9
    if which == 'wgrad_x3':
        ws_b = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d)); ws = workspace(dev, ws_b)
        if cin % 8 or cout % 8: continue
        t = timeit(lambda: _C.call('evk_conv2d_wgrad_x3', ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), ws_b, st))
    elif which == 'wgrad':
        t = timeit(lambda: _C.call('evk_conv2d_wgrad', ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), ws_b, st))
    elif which == 'fwd':
        t = timeit(lambda: _C.call('evk_conv2d_fwd', ctypes.byref(d), x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), 0, st))
    else:
        _C.call('evk_conv2d_pack_dgrad_weight', ctypes.byref(d), wt.data_ptr(), wtt.data_ptr(), st)
        t = timeit(lambda: _C.call('evk_conv2d_dgrad', ctypes.byref(d), dy.data_ptr(), wtt.data_ptr(), None, dx.data_ptr(), st))
    rows.append((cnt*t, f'{cnt}x {h}x{w} {cin}->{cout} k{k}s{s}: {t*1e3:.3f} ms {gf/t/1e3:.1f} TF  (x{cnt} = {cnt*t*1e3:.2f} ms)'))
    tot_t += cnt*t; tot_gf += cnt*gf
for _, r in sorted(rows, reverse=True)[:int(os.environ.get('TOP', 12))]: print(r)
print(f'TOTAL {which}: {tot_t*1e3:.2f} ms/step, {tot_gf/tot_t/1e3:.1f} TF/s over {tot_gf:.0f} GF')
