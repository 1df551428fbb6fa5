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


def _r18(cuda, seed=0):
    import ever_amd as er
    torch.manual_seed(seed)
    widths = (64, 128, 256, 512)
    m = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet18', in_channels=4),
                              head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                                        fs_relation=dict(scene_embedding_channels=512))))
    # (random running statistics: an eval-mode BatchNorm with the initial 0 / 1 would hide a wrong fold)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.2)
            mod.running_var.uniform_(0.5, 1.5)
    return m.to(cuda).eval()


@pytest.mark.parametrize('fold', [False, True], ids=['bn-layers', 'bn-folded'])
def test_export_model_torchscript_round_trip_equals_eager(cuda, tmp_path, fold):
    """reference api/infer_tool.py:70-74 (export_model = torch.jit.trace + torch.jit.save).  The no-grad forward of every HIP
    layer kind is a registered `ever_amd::` operator (hip/oplib.py), so the trace holds operator calls instead of opaque Python:
    the traced module, and the module loaded back from the file, give the eager output BIT FOR BIT (same kernels, same
    operands), also on a second input of the traced shape."""
    from ever_amd.api import infer_tool
    from ever_amd.module.fold import fold_batchnorm, unfold_batchnorm
    m = _r18(cuda)
    x = torch.randn(2, 4, 64, 64, device=cuda)
    x2 = torch.randn(2, 4, 64, 64, device=cuda)
    traced = infer_tool.trace_model(m, torch.ones(2, 4, 64, 64, device=cuda), fold=fold)
    kinds = {n.kind() for n in traced.graph.nodes()} | {n.kind() for n in traced.inlined_graph.nodes()}
    assert any(k.startswith('ever_amd::') for k in kinds), kinds
    assert not any('PythonOp' in k for k in kinds), kinds
    assert ('ever_amd::conv2d_folded' in kinds) == fold
    if not fold:
        unfold_batchnorm(m)
    with torch.no_grad():
        want, want2 = m(x), m(x2)
        got, got2 = traced(x), traced(x2)
    assert torch.equal(got, want) and torch.equal(got2, want2)
    path = str(tmp_path / 'farseg_r18.pt')
    torch.jit.save(traced, path)
    back = torch.jit.load(path)
    with torch.no_grad():
        assert torch.equal(back(x2), want2)


def test_export_model_from_config_and_checkpoint(cuda, tmp_path):
    """the reference's call: export_model(config_path, checkpoint_path, input_shape, output_path)"""
    import ever_amd as er
    from ever_amd.api import infer_tool
    cfg = tmp_path / 'cfg.py'
    cfg.write_text("config = dict(model=dict(type='FarSeg', params=dict(encoder=dict(resnet_type='resnet18', in_channels=4), "
                   "head=dict(fpn=dict(in_channels_list=(64, 128, 256, 512), out_channels=256), "
                   "fs_relation=dict(scene_embedding_channels=512)))))\n")
    m = _r18(cuda, seed=3)
    torch.save({'model': m.state_dict(), 'opt': {}, 'global_step': 7}, tmp_path / 'checkpoint-7.pth')
    out = str(tmp_path / 'exported.pt')
    infer_tool.export_model(str(cfg), str(tmp_path / 'checkpoint-7.pth'), (1, 4, 64, 64), out)
    back = torch.jit.load(out)
    x = torch.randn(1, 4, 64, 64, device=cuda)
    from ever_amd.module.fold import fold_batchnorm
    with torch.no_grad():
        unfolded = m(x)
        folded = fold_batchnorm(m)(x)       # export_model folds BatchNorm into the convolutions before tracing
        got = back(x)
    assert torch.equal(got, folded)
    assert float((got - unfolded).abs().max()) < 1e-5


def test_torch_compile_hook_trains_like_eager(cuda):
    """reference trainer.py:241-243: `config.train.torch_compile` -> torch.compile(model, **kwargs).  The HIP entry points are
    opaque to the compiler (hip.compiler_opaque); two SGD steps through the compiled model equal the eager ones bit for bit."""
    import ever_amd as er
    from ever_amd.trainer.trainer import Trainer

    class _T(Trainer):
        def __init__(self):
            self._cfg = er.AttrDict.from_dict(dict(train=dict(torch_compile=dict(dynamic=False))))
    a, b = _r18(cuda, seed=5).train(), _r18(cuda, seed=5).train()
    b.load_state_dict(a.state_dict())
    cb = _T().torch_compile(b)
    assert cb is not b
    oa = er.opt.FusedSGD(a.parameters(), lr=0.01, momentum=0.9)
    ob = er.opt.FusedSGD(b.parameters(), lr=0.01, momentum=0.9)
    x = torch.randn(2, 4, 64, 64, device=cuda)
    y = (torch.rand(2, 64, 64, device=cuda) < 0.3).long()
    for _ in range(2):
        la = a(x, y)
        sum(la.values()).backward()
        oa.step(); oa.zero_grad(set_to_none=True)
        lb = cb(x, y)
        sum(lb.values()).backward()
        ob.step(); ob.zero_grad(set_to_none=True)
        for k in la:
            assert torch.equal(la[k], lb[k]), k
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q), k


def test_library_operators_differentiate(cuda):
    """SURVEY 8 b4 / VERDICT r4 weak 12: the `ever_amd::*` operators carry autograd.  A direct call on tensors that require grad
    gives the gradients of the package's own autograd.Function path — the operator's backward re-runs the entry point under
    autograd (hip/oplib.py), the same kernels in the same order: bit for bit."""
    from ever_amd.hip import functional as HF
    ops = torch.ops.ever_amd
    g = torch.Generator().manual_seed(11)
    cl = torch.channels_last

    def leaf(*shape, scale=1.0):
        t = (torch.randn(*shape, generator=g) * scale).to(cuda)
        return (t.contiguous(memory_format=cl) if t.dim() == 4 else t).requires_grad_()

    def same(a, b):
        assert len(a) == len(b)
        for u, v in zip(a, b):
            assert (u is None) == (v is None)
            if u is not None:
                assert torch.equal(u, v), float((u - v).abs().max())

    # convolution (3x3, bias, fused ReLU): x, weight and bias gradients
    x, w, b = leaf(2, 32, 24, 20), leaf(48, 32, 3, 3, scale=0.1), leaf(48)
    dy = torch.randn(2, 48, 24, 20, generator=g).to(cuda).contiguous(memory_format=cl)
    y1 = ops.conv2d(x, w, b, [1, 1], [1, 1], [1, 1], True)
    g1 = torch.autograd.grad(y1, (x, w, b), dy)
    y2 = HF._conv2d_plain(x, w, b, (1, 1), (1, 1), (1, 1), relu=True)
    g2 = torch.autograd.grad(y2, (x, w, b), dy)
    assert torch.equal(y1, y2)
    same(g1, g2)
    # one-tap convolution without bias, only the input needs a gradient
    w1 = (torch.randn(64, 32, 1, 1, generator=g) * 0.1).to(cuda).contiguous(memory_format=cl)
    y1 = ops.conv2d(x, w1, None, [1, 1], [0, 0], [1, 1], False)
    y2 = HF._conv2d_plain(x, w1, None, (1, 1), (0, 0), (1, 1), relu=False)
    d1 = torch.randn_like(y1)
    same(torch.autograd.grad(y1, (x,), d1), torch.autograd.grad(y2, (x,), d1))
    # element-wise / pooling / resampling operators
    a, c = leaf(2, 16, 12, 12), leaf(2, 16, 12, 12)
    for op, plain, args in ((ops.relu, HF._relu_plain, (a,)), (ops.add, HF._add_plain, (a, c)),
                            (ops.max_pool3x3s2, HF._max_pool_plain, (a,)), (ops.global_avg_pool, HF._gap_plain, (a,))):
        o1, o2 = op(*args), plain(*args)
        d = torch.randn_like(o1)
        assert torch.equal(o1, o2)
        same(torch.autograd.grad(o1, args, d), torch.autograd.grad(o2, args, d))
    o1, o2 = ops.upsample_bilinear(a, 2.0, 2.0), HF._bilinear_plain(a, (2.0, 2.0))
    d = torch.randn_like(o1)
    same(torch.autograd.grad(o1, (a,), d), torch.autograd.grad(o2, (a,), d))
    top, lat = leaf(2, 16, 6, 6), leaf(2, 16, 12, 12)
    o1, o2 = ops.upsample_nearest2x_add(top, lat), HF._nearest_add_plain(top, lat)
    d = torch.randn_like(o1)
    same(torch.autograd.grad(o1, (top, lat), d), torch.autograd.grad(o2, (top, lat), d))
    # eval-mode BatchNorm (+ residual, ReLU): input, affine parameters and residual
    gam, bet, res = leaf(16), leaf(16), leaf(2, 16, 12, 12)
    rm, rv = torch.randn(16, generator=g).to(cuda), (torch.rand(16, generator=g) + 0.5).to(cuda)
    o1 = ops.batch_norm_eval(a, gam, bet, rm, rv, 1e-5, res, True)
    o2 = HF._bn_act_plain(a, gam, bet, rm, rv, False, 0.1, 1e-5, residual=res, relu=True)
    d = torch.randn_like(o1)
    assert torch.equal(o1, o2)
    same(torch.autograd.grad(o1, (a, gam, bet, res), d), torch.autograd.grad(o2, (a, gam, bet, res), d))
