"""The consumer side of a trained checkpoint and the small reference modules around upsampling (a9):
api.infer_tool (reference ever/api/infer_tool.py:16-74), magic.bigimage.sliding_window_inference (on top of
sliding_window.py:8-33) and module.ops.ConvUpsampling (ops.py:169-181), through the HIP path."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_conv_upsampling_matches_torch(cuda):
    import ever_amd as er
    torch.manual_seed(4)
    m = er.module.ConvUpsampling(16, 8, scale_factor=2, kernel_size=3, padding=1).to(cuda)
    ref = torch.nn.Sequential(torch.nn.Conv2d(16, 8, 3, 1, 1), torch.nn.UpsamplingBilinear2d(scale_factor=2)).double()
    sd = m.state_dict()
    assert set(sd) == {'0.weight', '0.bias'}      # same keys as the reference Sequential(conv, Bf16compatible(up))
    ref.load_state_dict({k: v.cpu().double() for k, v in sd.items()})
    x = torch.randn(2, 16, 9, 7)
    xr = x.double().requires_grad_()
    yr = ref(xr)
    g = torch.randn_like(yr)
    yr.backward(g)
    xg = x.to(cuda).requires_grad_()
    y = m(xg)
    y.backward(g.float().to(cuda))
    assert tuple(y.shape) == (2, 8, 18, 14)
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 1e-5 * float(yr.abs().max()) + 1e-6
    assert float((xg.grad.cpu().double() - xr.grad).abs().max()) < 1e-4 * float(xr.grad.abs().max())
    assert float((m[0].weight.grad.cpu().double() - ref[0].weight.grad).abs().max()) < 1e-4 * float(ref[0].weight.grad.abs().max())


def test_infer_tool_round_trip_and_sliding_window_inference(cuda, tmp_path):
    """Train-side checkpoint layout (core/checkpoint.py: checkpoint-<step>.pth with model / opt / global_step) ->
    api.infer_tool.build_from_model_dir -> the same eval-mode logits; then tiled inference over an image larger than
    the window equals whole-image inference wherever a pixel's receptive field lies inside its window (checked on the
    centre of a single-window case and against the averaging formula on an overlapping one)."""
    import ever_amd as er
    from ever_amd.api import infer_tool
    from ever_amd.magic.bigimage import sliding_window, sliding_window_inference
    cfg_py = tmp_path / 'config.py'
    cfg_py.write_text(
        "config = dict(model=dict(type='FarSeg', params=dict(\n"
        "    encoder=dict(resnet_type='resnet18', in_channels=3),\n"
        "    head=dict(fpn=dict(in_channels_list=(64, 128, 256, 512), out_channels=256),\n"
        "              fs_relation=dict(scene_embedding_channels=512)))))\n")
    torch.manual_seed(5)
    m = er.builder.make_model(er.config.import_config(str(cfg_py))['model'])
    m.eval()
    torch.save({'model': m.state_dict(), 'opt': {}, 'global_step': 7}, tmp_path / 'checkpoint-7.pth')
    loaded, step = infer_tool.build_from_model_dir(str(tmp_path))
    assert step == 7 and type(loaded).__name__ == 'FarSeg'
    for (ka, va), (kb, vb) in zip(m.state_dict().items(), loaded.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    m, loaded = m.to(cuda), loaded.to(cuda)
    x = torch.randn(1, 3, 128, 128, device=cuda)
    with torch.no_grad():
        a, b = m(x), loaded(x)
    assert torch.equal(a, b)
    # one window that covers the image: tiled inference IS whole-image inference
    s1 = sliding_window_inference(loaded, x, kernel_size=128, stride=64, batch_size=2)
    assert torch.allclose(s1, a, atol=1e-6)
    # overlapping windows on a larger image: every pixel = the mean of the scores of the windows that cover it
    big = torch.randn(1, 3, 192, 160, device=cuda)
    out = sliding_window_inference(loaded, big, kernel_size=128, stride=64, batch_size=3)
    boxes = np.unique(sliding_window((192, 160), 128, 64), axis=0)
    acc = torch.zeros_like(out)
    cnt = torch.zeros((1, 1, 192, 160), device=cuda)
    with torch.no_grad():
        for x0, y0, x1, y1 in boxes:
            acc[:, :, y0:y1, x0:x1] += loaded(big[:, :, y0:y1, x0:x1].contiguous())
            cnt[:, :, y0:y1, x0:x1] += 1
    assert float(cnt.min()) >= 1
    assert torch.allclose(out, acc / cnt, atol=1e-5)
