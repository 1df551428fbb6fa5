"""dev tool: per-parameter gradient comparison HIP vs oracle (CPU) for a golden config."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import farseg_ref, portable
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_e2e_gpu import _hip_model
name = sys.argv[1]
dt = torch.float64 if len(sys.argv) > 2 and sys.argv[2] == 'f64' else torch.float32
meta = json.load(open(f'tests/golden/e2e_{name}.json'))
dev = torch.device('cuda:0')
m = _hip_model(meta, dev).train()
ora = farseg_ref.FarSegRef(meta['resnet_type'], meta['in_channels'], meta['num_classes'], meta['decoder_channels'], meta['classifier_kernel'])
farseg_ref.load_portable_weights(ora, portable.fill_state_dict(ora.state_dict()))
ora = ora.to(dt).train()
x, y = portable.synthetic_batch(name, meta['n'], meta['in_channels'], meta['hw'], meta['hw'], meta['num_classes'])
xt, yt = torch.from_numpy(x), torch.from_numpy(y)
lg_o = ora.logits(xt.to(dt)); lg_o.retain_grad()
lo = ora.loss_from_logits(lg_o, yt); sum(lo.values()).backward()
lg = m.head(m.en(xt.to(dev))); lg.retain_grad()
lh = m.loss(lg, yt.to(dev)); sum(lh.values()).backward()
def rel(a, b):
    a = a.detach().cpu().contiguous().double().numpy(); b = b.detach().double().numpy()
    return np.abs(a-b).max() / max(np.abs(b).max(), 1e-30), np.linalg.norm(a)/max(np.linalg.norm(b),1e-30)
print('logits', rel(lg, lg_o), 'dlogits', rel(lg.grad, lg_o.grad))
for (k, p), (k2, q) in list(zip(m.named_parameters(), ora.named_parameters()))[::-1]:
    e, r = rel(p.grad, q.grad)
    print(f'{k:60s} maxrel {e:.2e} normratio {r:.5f} |g|={float(q.grad.norm()):.3e}')
