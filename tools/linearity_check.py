"""dev tool / test helper (GPU box): batch-linearity of the backward pass and per-sample independence of the forward at the
BASELINE sizes, meant to be run with the kernel plan PINNED (EVK_X3_HALO_MIN_WG=0: the LDS-halo 3x3 kernel wherever the
shape allows, whatever the batch), so that a batch and its halves accumulate every convolution output in the same order:
what remains is the split-K rounding of the weight gradients.  usage: python tools/linearity_check.py c2|c3  -> JSON line"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
from ever_amd.hip import functional as HF
from ever_amd.module import loss as L
cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu')
if cfg == 'c2':
    torch.manual_seed(2333)
    m = er.module.FarSeg(dict()).to(dev).eval()
    g.manual_seed(99)
    n, c, hw = 16, 3, 512
else:
    torch.manual_seed(7)
    m = er.module.FarSegPP(dict(encoder=dict(in_channels=4))).to(dev).eval()
    g.manual_seed(123)
    n, c, hw = 8, 4, 1024
x = torch.randn(n, c, hw, hw, generator=g).to(dev)
y = (torch.rand(n, hw, hw, generator=g) < 0.3).long()
y[:, :8, :8] = 255
y = y.to(dev)
h = n // 2
assert int((y[:h] != 255).sum()) == int((y[h:] != 255).sum())


def run(xs, ys):
    m.zero_grad(set_to_none=True)
    lg = m.head(m.en(HF.as_nhwc(xs)))
    L.binary_cross_entropy_with_logits(lg, ys).backward()
    return lg.detach(), [p.grad.double().clone() for p in m.parameters()]


lg_f, full = run(x, y)
lg_a, ga = run(x[:h], y[:h])
lg_b, gb = run(x[h:], y[h:])
torch.cuda.synchronize()
num = sum(float(((a + b) / 2 - f).square().sum()) for a, b, f in zip(ga, gb, full))
den = sum(float(f.square().sum()) for f in full)
worst = max(float(((a + b) / 2 - f).norm() / f.norm()) for a, b, f in zip(ga, gb, full) if float(f.norm()) > 1e-8)
fwd = float((torch.cat([lg_a, lg_b]) - lg_f).abs().max() / lg_f.abs().max())
print(json.dumps(dict(config=cfg, conv_math=HF.get_conv_math(), global_rel_l2=(num / den) ** 0.5, worst_tensor=worst,
                      forward_max_rel=fwd, halo_min_wg=os.environ.get('EVK_X3_HALO_MIN_WG', '256'))))
