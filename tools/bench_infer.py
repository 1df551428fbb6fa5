"""dev tool: eval-mode (inference) throughput of FarSeg-R50 at batch 16 x 512^2, BatchNorm as a pass vs folded."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
from ever_amd.module.fold import fold_batchnorm
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = er.module.FarSeg(dict()).to(dev).eval()
x = torch.randn(16, 3, 512, 512, device=dev)
def run(tag):
    with torch.no_grad():
        for _ in range(3): m(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): m(x)
        e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f'{tag}: {ms:.2f} ms / batch 16  = {16e3/ms:.0f} tiles/s')
run('BatchNorm passes')
fold_batchnorm(m)
run('folded          ')
