"""GPU: the training-side pieces around the kernels — fused SGD + clipping vs torch.optim.SGD,
the Launcher driving the HIP FarSeg for a few steps, encoder options (dilation / frozen stages /
frozen BN / activation checkpointing) against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('decoupled', [False, True])
def test_fused_adam_and_adamw_match_torch(cuda, decoupled):
    """opt/optimizer.py:8-9 registers 'adam' / 'adamw': one-launch HIP step vs torch.optim.Adam / AdamW over 4 steps
    (clipping folded in; a parameter that skips a step keeps its own step count), identical state-dict schema, and a
    resume from the torch optimizer's state dict (NCHW-dense moments re-laid to the channels_last parameters)."""
    import ever_amd as er
    torch.manual_seed(1)
    shapes = [(64, 4, 7, 7), (64,), (256, 64, 1, 1), (32, 32, 3, 3), (1,), (7, 3), (1031,)]
    ps_a = [torch.randn(s, device=cuda) for s in shapes]
    ps_a = [(p.contiguous(memory_format=torch.channels_last) if p.dim() == 4 else p).requires_grad_() for p in ps_a]
    ps_b = [p.detach().clone().contiguous().requires_grad_() for p in ps_a]
    kw = dict(lr=0.01, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.02)
    fused, stock = (er.opt.FusedAdamW, torch.optim.AdamW) if decoupled else (er.opt.FusedAdam, torch.optim.Adam)
    assert er.registry.OPT['adamw' if decoupled else 'adam'] is fused
    oa, ob = fused(ps_a, **kw), stock(ps_b, **kw)
    for step in range(4):
        for k, (p, q) in enumerate(zip(ps_a, ps_b)):
            if step == 1 and k == 2:
                p.grad = q.grad = None        # no gradient this step: its step counter falls behind the others
                continue
            g = torch.randn_like(q)
            p.grad, q.grad = g.clone(), g.clone()
        oa.fused_clip(max_norm=5.0)
        torch.nn.utils.clip_grad_norm_([q for q in ps_b if q.grad is not None], max_norm=5.0)
        oa.step()
        ob.step()
        for p, q in zip(ps_a, ps_b):
            assert torch.allclose(p, q, rtol=2e-5, atol=2e-6), (step, p.shape, float((p - q).abs().max()))
    sa, sb = oa.state_dict(), ob.state_dict()
    assert set(sa['state'][0]) == set(sb['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
    assert float(sa['state'][2]['step']) == float(sb['state'][2]['step']) == 3.0
    # resume from the stock optimizer's checkpoint
    oc = fused([p.detach().clone().contiguous(memory_format=torch.channels_last).requires_grad_() if p.dim() == 4
                else p.detach().clone().requires_grad_() for p in ps_b], **kw)
    import copy
    oc.load_state_dict(copy.deepcopy(sb))      # (torch shares same-device state tensors with the dict it loads)
    pc = [p for g in oc.param_groups for p in g['params']]
    for p, q in zip(pc, ps_b):
        g = torch.randn_like(q)
        p.grad, q.grad = g.clone(), g.clone()
    oc.step()
    ob.step()
    for p, q in zip(pc, ps_b):
        assert torch.allclose(p, q, rtol=2e-5, atol=2e-6)
        assert oc.state[p]['exp_avg'].stride() == p.stride()


def test_grad_clip_reaches_groups_that_take_torchs_own_step(cuda):
    """fused_clip only leaves a coefficient for the fused kernels; a group the fused step cannot take
    (amsgrad=True here) must still train on clipped gradients, also when it sits beside a fused group, and the table cache
    must not grow when parameters of one group are at different step counts."""
    import ever_amd as er
    torch.manual_seed(3)
    mk = lambda: [torch.randn(s, device=cuda) for s in ((32, 8, 3, 3), (32,), (17, 5))]
    base_a, base_b = mk(), mk()
    a1 = [p.clone().requires_grad_() for p in base_a]; a2 = [p.clone().requires_grad_() for p in base_b]
    b1 = [p.clone().requires_grad_() for p in base_a]; b2 = [p.clone().requires_grad_() for p in base_b]
    groups = lambda x, y: [dict(params=x, amsgrad=True), dict(params=y)]
    oa = er.opt.FusedAdam(groups(a1, a2), lr=0.01, betas=(0.9, 0.99))
    ob = torch.optim.Adam(groups(b1, b2), lr=0.01, betas=(0.9, 0.99))
    for step in range(6):
        for k, (p, q) in enumerate(zip(a1 + a2, b1 + b2)):
            if step == 2 and k == 4:
                p.grad = q.grad = None          # mixed step counts inside the fused group from here on
                continue
            g = 10.0 * torch.randn_like(q)
            p.grad, q.grad = g.clone(), g.clone()
        oa.fused_clip(max_norm=1.0)
        torch.nn.utils.clip_grad_norm_([q for q in b1 + b2 if q.grad is not None], max_norm=1.0)
        oa.step()
        ob.step()
        for p, q in zip(a1 + a2, b1 + b2):
            assert torch.allclose(p, q, rtol=2e-5, atol=2e-6), (step, tuple(p.shape), float((p - q).abs().max()))
        if step == 3:       # two step-count buckets in the fused group since this step
            n_tabs = len(oa._tabs)
    assert len(oa._tabs) == n_tabs, 'optimizer pointer tables keep growing with mixed step counts'


def test_fused_sgd_matches_torch_sgd(cuda):
    import ever_amd as er
    torch.manual_seed(0)
    shapes = [(64, 4, 7, 7), (64,), (256, 64, 1, 1), (128, 128, 3, 3), (1, 256, 1, 1), (1,), (7, 3), (1031,)]
    ps_a = [torch.randn(s, device=cuda).requires_grad_() for s in shapes]
    ps_a = [p.detach().contiguous(memory_format=torch.channels_last).requires_grad_() if p.dim() == 4 else p for p in ps_a]
    ps_b = [p.detach().clone().requires_grad_() for p in ps_a]
    kw = dict(lr=0.05, momentum=0.9, weight_decay=1e-4, nesterov=False)
    oa = er.opt.FusedSGD(ps_a, **kw)
    ob = torch.optim.SGD(ps_b, **kw)
    oa.er_config = dict(grad_clip=dict(max_norm=0.5, norm_type=2))
    for step in range(3):
        gs = [torch.randn_like(p) for p in ps_a]
        for k, (p, q, g) in enumerate(zip(ps_a, ps_b, gs)):
            p.grad = g.clone()
            q.grad = g.clone()
            if k == 7:   # a gradient at a 4-byte (not 16-byte) aligned address, as a view into a DDP bucket can be
                flat = torch.zeros(g.numel() + 1, device=cuda)
                flat[1:].copy_(g.reshape(-1))
                p.grad = flat[1:].view_as(g)
                assert p.grad.data_ptr() % 16 != 0
        oa.fused_clip(max_norm=0.5)
        ref_norm = torch.nn.utils.clip_grad_norm_(ps_b, max_norm=0.5)
        assert abs(oa.last_grad_norm.item() - ref_norm.item()) <= 1e-5 * ref_norm.item()
        oa.step()
        ob.step()
        for p, q in zip(ps_a, ps_b):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), (step, p.shape)
    sd = oa.state_dict()   # same schema as torch.optim.SGD (checkpoint compatible)
    assert 'momentum_buffer' in next(iter(sd['state'].values()))


def test_launcher_trains_hip_farseg(cuda, tmp_path):
    """3 Launcher iterations on the GPU (R18, 4-band 64x64, batch 2) + the same 3 iterations of the oracle
    model on the CPU with torch SGD: per-step losses must agree (fp32, 2e-3), the clipped-gradient norm to 1e-2."""
    import ever_amd as er
    from oracle import farseg_ref, portable
    from tests import plumbing_common as pc
    widths = (64, 128, 256, 512)
    model = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet18', in_channels=4),
                                  head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                                            fs_relation=dict(scene_embedding_channels=512))))
    ora = pc.OracleFarSeg(dict())
    model.load_state_dict(ora.state_dict(), strict=True)
    model = model.to(cuda)
    loader = torch.utils.data.DataLoader(pc.ToyTiles(), batch_size=2, shuffle=False)
    recs = {}
    for name, m in (('hip', model), ('cpu', ora)):
        sched = er.builder.make_learningrate(dict(type='poly', params=dict(base_lr=0.01, power=0.9, max_iters=3)))
        opt = er.builder.make_optimizer(er.AttrDict.from_dict(dict(type='sgd', params=dict(momentum=0.9, weight_decay=1e-4, lr=0.01),
                                                                   grad_clip=dict(max_norm=35, norm_type=2))),
                                        params=m.custom_param_groups())
        tl = er.Launcher(str(tmp_path / name), m, opt, sched)
        if name == 'cpu':
            tl._device = torch.device('cpu')
        rec = []
        orig = tl._logger.train_log

        def spy(_rec=rec, _orig=orig, **kw):
            _rec.append({k: float(v) for k, v in kw['loss_dict'].items()})
            return _orig(**kw)

        tl._logger.train_log = spy
        tl.train_by_config(loader, config=er.AttrDict.from_dict(dict(num_iters=3, save_ckpt_interval_epoch=1000)))
        recs[name] = rec
    print('per-step records:', recs)
    for a, b in zip(recs['hip'], recs['cpu']):
        for k in ('bce_loss', 'dice_loss'):
            assert a[k] == pytest.approx(b[k], rel=2e-3), (k, a, b)
        # the gradient norm is discontinuous in the rounding: at 64x64 the coarse maps hold 2x2..4x4 pixels, and ONE
        # ReLU decision on a pre-activation within 1e-5 of zero (tools/cmp_bn_epilogue.py found blocks.2.0: 8192
        # elements, forward 2e-5 apart between two statistic orders) moves the norm by 3e-3 — both sides are "right"
        assert a['grad_norm'] == pytest.approx(b['grad_norm'], rel=1e-2), (a, b)


@pytest.mark.parametrize('opts', [dict(output_stride=16), dict(output_stride=8), dict(freeze_at=2, batchnorm_trainable=False),
                                  dict(with_cp=(True, True, False, False))])
def test_encoder_options_match_oracle(cuda, opts):
    import ever_amd as er
    from oracle import farseg_ref, portable
    torch.manual_seed(0)
    enc = er.module.ResNetEncoder(dict(resnet_type='resnet50', in_channels=3, **opts))
    ora = farseg_ref.ResNetEncoderRef('resnet50', 3, opts.get('output_stride', 32))
    filled = portable.fill_state_dict(ora.state_dict())
    farseg_ref.load_portable_weights(ora, filled)
    enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in filled.items()}, strict=True)
    enc = enc.to(cuda).train()
    ora.train()
    if not opts.get('batchnorm_trainable', True):
        for m in ora.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
    x = torch.from_numpy(portable.normalish('encopt', (2, 3, 128, 128)))
    xg = x.to(cuda).requires_grad_()
    xo = x.clone().requires_grad_()
    fo, fg = ora(xo), enc(xg)
    ws = [torch.from_numpy(portable.normalish(f'encw{i}', tuple(f.shape))) for i, f in enumerate(fo)]
    sum((f * w).sum() for f, w in zip(fo, ws)).backward()
    sum((f * w.to(cuda)).sum() for f, w in zip(fg, ws)).backward()
    for i, (a, b) in enumerate(zip(fg, fo)):
        assert a.shape == b.shape
        err = (a.detach().cpu() - b.detach()).abs().max() / b.detach().abs().max()
        assert err < 1e-3, (i, float(err))
    g_a, g_b = xg.grad.cpu().double(), xo.grad.double()
    assert (g_a - g_b).norm() / g_b.norm() < 3e-2   # see test_e2e_gpu.py on gradient conditioning
    if opts.get('freeze_at'):
        assert enc.resnet.conv1.weight.grad is None and enc.resnet.layer1[0].conv1.weight.grad is None
        assert enc.resnet.layer2[0].conv1.weight.grad is not None
        assert enc.resnet.layer2[0].bn1.weight.grad is None     # frozen BN


def test_gradient_slots_equal_autograd_sum(cuda):
    """The encoder's stage outputs feed the next stage and the head; with gradient slots the head's gradient is added
    inside the next stage's first data-gradient launch (no autograd add pass).  Gradients must equal the plain
    autograd-sum path (EVK_GRAD_SLOTS=0) to rounding, and no ATen add may remain for the c2..c4 fan-out."""
    import os
    import ever_amd as er
    torch.manual_seed(21)
    m = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet50'))).to(cuda).train()
    x = torch.randn(2, 3, 128, 128, device=cuda)
    y = (torch.rand(2, 128, 128, device=cuda) < 0.3).long()

    def grads():
        m.zero_grad(set_to_none=True)
        sum(m(x, y).values()).backward()
        return [p.grad.double().clone() for p in m.parameters()]
    with_slots = grads()
    os.environ['EVK_GRAD_SLOTS'] = '0'
    try:
        plain = grads()
    finally:
        del os.environ['EVK_GRAD_SLOTS']
    num = sum(float((a - b).square().sum()) for a, b in zip(with_slots, plain))
    den = sum(float(b.square().sum()) for b in plain)
    assert (num / den) ** 0.5 < 1e-5, (num / den) ** 0.5     # same sums in another order (BN statistics moved once more)
    from torch.profiler import profile, ProfilerActivity

    def count_adds():
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            grads()
        return sum(e.count for e in prof.key_averages() if e.key in ('aten::add', 'aten::add_'))
    n_slots = count_adds()
    os.environ['EVK_GRAD_SLOTS'] = '0'
    try:
        n_plain = count_adds()
    finally:
        del os.environ['EVK_GRAD_SLOTS']
    print(f'aten::add calls per step: {n_plain} with autograd sums, {n_slots} with gradient slots')
    # c2, c3, c4 (encoder outputs: next stage + FPN lateral) and, since round 4, the FPN's inner maps of levels 3..5 (output
    # convolution + the top-down path of the next finer level): one whole-map add pass each, gone
    assert n_slots == n_plain - 6, (n_plain, n_slots)


def test_no_activation_outlives_the_step(cuda):
    """Reference cycles through a convolution's ctx (y -> grad_fn -> ctx -> state -> y) once kept the whole upstream graph
    — 2.8 GB of activations per FarSeg-R50 step — allocated until the cyclic collector ran, and the caching allocator
    growing (hipMalloc inside the step) whenever it was late.  With the collector OFF, memory after a step must be what
    it was before it, and no tensor may sit in a cycle."""
    import gc
    import ever_amd as er
    torch.manual_seed(3)
    model = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet18'),
                                  head=dict(fpn=dict(in_channels_list=(64, 128, 256, 512), out_channels=256),
                                            fs_relation=dict(scene_embedding_channels=512)))).to(cuda).train()
    opt = er.opt.FusedSGD(model.parameters(), lr=0.01, momentum=0.9)
    x = torch.randn(2, 3, 128, 128, device=cuda)
    y = dict(cls=torch.randint(0, 2, (2, 128, 128), device=cuda))

    def step():
        out = model(x, y)
        sum(v for k, v in out.items() if k.endswith('loss')).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    try:
        before = torch.cuda.memory_allocated()
        step()
        torch.cuda.synchronize()
        after = torch.cuda.memory_allocated()
        gc.set_debug(gc.DEBUG_SAVEALL)
        gc.collect()
        cyclic = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
    finally:
        gc.set_debug(0)
        gc.garbage.clear()
        gc.enable()
    assert not cyclic, [tuple(t.shape) for t in cyclic]
    assert after <= before + (1 << 20), (before, after)


def test_launcher_fp16_mode_runs_the_gradscaler_protocol(cuda, tmp_path):
    """`mixed_precision='fp16'` (reference launcher.py:46-80, interface/module.py:63-94) on a HIP model: same kernels as the
    default (fp16 MFMAs on scaled, split operands), plus GradScaler scale / unscale_ / step / update.  The scale is a power of
    two and every backward kernel is linear in the incoming gradient, so the logged losses equal the fp32 launcher's."""
    import ever_amd as er
    from tests import plumbing_common as pc
    widths = (64, 128, 256, 512)
    loader = torch.utils.data.DataLoader(pc.ToyTiles(n=8), batch_size=2, shuffle=False)
    recs = {}
    for name in ('fp32', 'fp16'):
        torch.manual_seed(7)
        m = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet18', in_channels=4),
                                  head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                                            fs_relation=dict(scene_embedding_channels=512)))).to(cuda)
        sched = er.builder.make_learningrate(dict(type='poly', params=dict(base_lr=0.01, power=0.9, max_iters=4)))
        opt = er.builder.make_optimizer(er.AttrDict.from_dict(dict(type='sgd', params=dict(momentum=0.9, weight_decay=1e-4, lr=0.01),
                                                                   grad_clip=dict(max_norm=35, norm_type=2))),
                                        params=m.custom_param_groups())
        tl = er.Launcher(str(tmp_path / name), m, opt, sched, mixed_precision=name)
        rec = []
        orig = tl._logger.train_log

        def spy(_rec=rec, _orig=orig, **kw):
            _rec.append({k: float(v) for k, v in kw['loss_dict'].items()})
            return _orig(**kw)
        tl._logger.train_log = spy
        tl.train_by_config(loader, config=er.AttrDict.from_dict(dict(num_iters=4, save_ckpt_interval_epoch=1000)))
        recs[name] = rec
        if name == 'fp16':
            assert tl.scaler is not None and tl.scaler.get_scale() > 1.0
    assert len(recs['fp16']) == 4
    for a, b in zip(recs['fp32'], recs['fp16']):
        for k in ('bce_loss', 'dice_loss'):
            assert a[k] == pytest.approx(b[k], rel=1e-5), (k, a, b)
