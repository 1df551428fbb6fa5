"""dev probe (GPU box): board power (rocm-smi) while ONE kernel family runs in a loop — how far below the cap do the
HBM-bound phases of the step run?  usage: python tools/probes/power_phase.py"""
import ctypes, os, subprocess, sys, threading, time, torch, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ever_amd import _C
dev = torch.device('cuda:0'); lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
B = 16
nw = int(lib.evk_absmax_words())
aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=dev)

def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=10).stdout
        return out.strip().replace('\n', ' ')[:600]
    except Exception as e:
        return 'ERR ' + str(e)

def conv(h, cin, cout, k):
    d = _C.ConvDesc(B, h, h, cin, h, h, cout, k, k, 1, 1, k // 2, k // 2, 1, 1)
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(B, h, h, cin, generator=g)).to(dev)
    wt = (torch.randn(cout, k, k, cin, generator=g) * 0.05).to(dev)
    bits = [torch.zeros(nw, dtype=torch.int32, device=dev) for _ in range(2)]
    _C.call('evk_absmax', x.data_ptr(), x.numel(), bits[0].data_ptr(), aws.data_ptr(), st)
    _C.call('evk_absmax', wt.data_ptr(), wt.numel(), bits[1].data_ptr(), aws.data_ptr(), st)
    planes = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
    _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 0, planes.data_ptr(), bits[1].data_ptr(), st)
    out = torch.empty(B, h, h, cout, device=dev)
    npart = ctypes.c_int32(0)
    keep = (x, wt, bits, planes, out)
    return lambda: _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), x.data_ptr(), bits[0].data_ptr(), planes.data_ptr(), bits[1].data_ptr(),
                           None, None, out.data_ptr(), 0, None, 0, ctypes.byref(npart), None, st), keep

def bn(rows, c):
    x = torch.randn(rows, c, device=dev); dy = torch.randn(rows, c, device=dev); dx = torch.empty_like(x)
    g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev); mean = x.mean(0); invstd = 1 / x.std(0)
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    wsb = lib.evk_bn_workspace_bytes(rows, c); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    bits = torch.zeros(nw, dtype=torch.int32, device=dev)
    keep = (x, dy, dx, g, b, mean, invstd, dg, db, ws, bits)
    return lambda: _C.call('evk_bn_bwd', dy.data_ptr(), x.data_ptr(), None, g.data_ptr(), b.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                           dx.data_ptr(), None, dg.data_ptr(), db.data_ptr(), rows, c, 1, 1, ws.data_ptr(), wsb, bits.data_ptr(), st), keep

def copy():
    a = torch.randn(64 << 20, device=dev); b_ = torch.empty_like(a)
    return lambda: b_.copy_(a), (a, b_)

print('idle:', smi(), flush=True)
for name, mk in (('3x3x256 @128^2 halo conv (matrix-bound)', lambda: conv(128, 256, 256, 3)),
                 ('1x1 256->256 @128^2 (ps2, HBM-bound conv)', lambda: conv(128, 256, 256, 1)),
                 ('1x1 1024->256 @32^2', lambda: conv(32, 1024, 256, 1)),
                 ('BatchNorm backward 262144 x 256 (HBM-bound)', lambda: bn(262144, 256)),
                 ('device copy 268 MB', copy)):
    fn, keep = mk()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    samples = []
    stop = [False]
    def sampler():
        time.sleep(0.7)
        while not stop[0]:
            samples.append(smi()); time.sleep(0.5)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 3.5:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    stop[0] = True; th.join()
    print(f'== {name}: {(time.time() - t0) / n * 1e6:.1f} us per call', flush=True)
    for s in samples[:4]: print('   ', s, flush=True)
    del keep
