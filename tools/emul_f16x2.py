"""Dev probe (CPU): accuracy of the f16x2 convolution arithmetic — per-tensor power-of-two scale, 2-term fp16 split of both
operands, the three retained partial products accumulated in fp32 — at the LOGITS of the whole FarSeg network against an
fp64 evaluation of the oracle, next to the fp32 oracle itself and the bf16x3 split (tools/emul_split_bf16.py).
usage: python tools/emul_f16x2.py <tile> <batch> <resnet18|resnet50>      (R50 128 2: 1.33e-4 / fp32 1.54e-4 / bf16x3 0.94e-4)"""
import os, sys, math
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import farseg_ref, portable
torch.set_num_threads(16)
_conv = F.conv2d
MODE = {'m': None}

def split3(t):
    i = t.contiguous().view(torch.int32)
    h = (i & -65536).view(torch.float32); r = t - h
    m = (r.view(torch.int32) & -65536).view(torch.float32); l = r - m
    l = (l.view(torch.int32) & -65536).view(torch.float32)
    return h, m, l

def split_h(t, rtz=False):
    amax = float(t.abs().max())
    if amax == 0: return (t, t), 1.0
    s = 2.0 ** (math.floor(math.log2(amax)) - 13)
    xs = t / s
    h = xs.half().float()
    l = (xs - h).half().float()
    return (h, l), s

def conv_emul(x, w, b=None, *a, **k):
    m = MODE['m']
    if m is None or x.dtype != torch.float32:
        return _conv(x, w, b, *a, **k)
    if m == 'bf16x3':
        xs, ws = split3(x), split3(w)
        pairs = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]; sc = 1.0
    else:
        (xs, sx), (ws, sw) = split_h(x), split_h(w)
        pairs = [(1, 0), (0, 1), (0, 0)] + ([(1, 1)] if m == 'fp16x2+ll' else [])
        pairs = pairs[::-1] if False else pairs
        sc = sx * sw
    y = None
    for i, j in pairs:
        t = _conv(xs[i], ws[j], None, *a, **k)
        y = t if y is None else y + t
    y = y * sc
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y

def run(hw, n, dt, mode, arch):
    MODE['m'] = mode
    ora = farseg_ref.FarSegRef(arch, 3, 1)
    farseg_ref.load_portable_weights(ora, portable.fill_state_dict(ora.state_dict()))
    ora = ora.to(dt).train()
    x, y = portable.synthetic_batch('oracle256', n, 3, hw, hw, 1)
    with torch.no_grad():
        lg = ora.logits(torch.from_numpy(x).to(dt))
    return lg.double().numpy()

if __name__ == '__main__':
    hw, n, arch = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import torch.nn.modules.conv as C
    F.conv2d = conv_emul; torch.nn.functional.conv2d = conv_emul; C.F.conv2d = conv_emul
    ref64 = run(hw, n, torch.float64, None, arch)
    rel = lambda a: np.abs(a - ref64).max() / np.abs(ref64).max()
    print('fp32 oracle vs fp64:', rel(run(hw, n, torch.float32, None, arch)))
    for m in ('bf16x3', 'fp16x2', 'fp16x2+ll'):
        print(f'{m:10s} vs fp64:', rel(run(hw, n, torch.float32, m, arch)))
