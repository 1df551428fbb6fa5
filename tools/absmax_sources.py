"""dev tool (GPU box): where the f16x2 operand scales of one FarSeg-R50 training step come from — produced by the
kernel that wrote the tensor (fused), reused from an earlier consumer (hits), or a stand-alone evk_absmax pass."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
from ever_amd.hip import functional as F
from ever_amd import _C
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = er.module.FarSeg(dict()).to(dev).train()
x = torch.randn(16, 3, 512, 512, device=dev)
y = (torch.rand(16, 512, 512, device=dev) > 0.5).long()
for it in range(2):
    for k in F.absmax_stats: F.absmax_stats[k] = 0
    sizes = []
    orig = _C.call
    import traceback
    def spy(name, *a):
        if name == 'evk_absmax':
            fr = [f.name for f in traceback.extract_stack()[:-1] if 'ever_amd' in f.filename]
            sizes.append((round(a[1] * 4 / 1e6), '>'.join(fr[-6:-1])))
        return orig(name, *a)
    _C.call = spy
    sum(m(x, y).values()).backward()
    torch.cuda.synchronize()
    _C.call = orig
    print('step', it, dict(F.absmax_stats))
    for mb, who in sorted(sizes, reverse=True)[:40]:
        print('   ', mb, 'MB', who)
