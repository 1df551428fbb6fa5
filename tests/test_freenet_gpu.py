"""FreeNet-style 200-band model on the HIP kernels vs its stock-torch restatement (oracle/freenet_ref.py,
"parity unpinned": the reference tree holds no definition of this model), and the C5 configuration
[1, 200, 610, 340] end to end (divisible padding, odd spatial sizes through every kernel)."""
import numpy as np
import pytest
import torch

from oracle import freenet_ref, portable

pytestmark = pytest.mark.gpu


def _models(cuda, **kw):
    from ever_amd.module import FreeNet
    m = FreeNet(dict(**kw))
    filled = portable.fill_state_dict(m.state_dict())
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in filled.items()}, strict=True)
    ora = freenet_ref.FreeNetRef(**{k: v for k, v in m.config.items() if k in ('in_channels', 'num_classes', 'num_blocks',
                                                                             'reduction_ratio')})
    assert list(ora.state_dict().keys()) == list(m.state_dict().keys())
    ora.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in filled.items()}, strict=True)
    return m.to(cuda).train(), ora.train()


def test_freenet_matches_torch_restatement(cuda, conv_math):
    m, ora = _models(cuda, in_channels=200, num_classes=16)
    x = torch.from_numpy(portable.normalish('freenet/x', (1, 200, 72, 48)))
    y = torch.from_numpy(portable.integers('freenet/y', (1, 72, 48), 17).astype(np.int64))   # 0 = unlabelled
    w = torch.from_numpy((portable.uniform01('freenet/w', 72 * 48) < 0.5).astype(np.float32).reshape(1, 72, 48))
    lg_o = ora.logits(x)
    lo = ora.loss(lg_o, y, w)
    lo.backward()
    out = m(x.to(cuda), y.to(cuda), w.to(cuda))
    out['cls_loss'].backward()
    m.eval()
    with torch.no_grad():
        lg = m(x.to(cuda)).cpu().contiguous().numpy().astype(np.float64)
    ref = lg_o.detach().numpy().astype(np.float64)
    assert np.abs(lg - ref).max() / np.abs(ref).max() < 1e-3
    assert (lg.argmax(1) == ref.argmax(1)).mean() > 0.999
    assert abs(out['cls_loss'].item() - lo.item()) <= 1e-4 * abs(lo.item())
    dot = na = nb = 0.0
    for (k, p), (_, q) in zip(m.named_parameters(), ora.named_parameters()):
        if q.grad is None:          # the published model builds four fuse convolutions and uses three
            assert p.grad is None, k
            continue
        g, r = p.grad.cpu().contiguous().numpy().astype(np.float64), q.grad.numpy().astype(np.float64)
        dot, na, nb = dot + float((g * r).sum()), na + float((g * g).sum()), nb + float((r * r).sum())
    cos = dot / np.sqrt(na * nb)
    assert cos > 0.9999 and abs(np.sqrt(na / nb) - 1) < 2e-3, (cos, np.sqrt(na / nb))


def test_freenet_c5_global_forward_backward(cuda):
    """BASELINE configs[4]: one 200-band 610 x 340 scene, padded to a multiple of the model stride (8)."""
    from ever_amd.module.freenet import divisible_pad
    m, _ = _models(cuda)
    g = torch.Generator().manual_seed(3)
    x = divisible_pad(torch.randn(1, 200, 610, 340, generator=g), 8)
    y = divisible_pad(torch.randint(0, 17, (1, 610, 340), generator=g).float(), 8).long()
    assert x.shape[-2:] == (616, 344) and y.shape[-2:] == (616, 344)
    out = m(x.to(cuda), y.to(cuda))
    out['cls_loss'].backward()
    assert torch.isfinite(out['cls_loss']) and 0.5 < out['cls_loss'].item() < 50.0   # hash-initialised weights
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    m.eval()
    with torch.no_grad():
        lg = m(x.to(cuda))
    assert lg.shape == (1, 16, 616, 344)
