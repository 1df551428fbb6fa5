"""The training step as one captured hipGraph (ever_amd/core/graph.py): after three eager steps the whole
step — forward, BCE + dice, backward, gradient clipping, fused SGD with the learning rate read from a device word, the
weight-plane refresh — is captured and replayed.  Everything in it is deterministic, so parameters, momentum buffers and
BatchNorm statistics after N steps must be BIT-identical to N eager steps, also while the learning rate changes every step
and the input changes every step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(cuda, graphed, steps=7, short_at=None):
    import ever_amd as er
    from ever_amd.core.graph import GraphedTrainStep
    torch.manual_seed(11)
    widths = (64, 128, 256, 512)
    m = er.module.FarSeg(dict(
        encoder=dict(resnet_type='resnet18', in_channels=4),
        head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                  fs_relation=dict(scene_embedding_channels=512)))).to(cuda).train()
    opt = er.opt.FusedSGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt.er_config = dict(grad_clip=dict(max_norm=35, norm_type=2))

    def step_fn(x, y):
        out = m(x, y)
        sum(v for k, v in out.items() if k.endswith('loss')).backward()
        opt.fused_clip(max_norm=35)
        opt.step()
        opt.zero_grad(set_to_none=True)
        return out
    step = GraphedTrainStep(step_fn, opt, modules=(m,)) if graphed else step_fn
    g = torch.Generator().manual_seed(5)
    losses = []
    for i in range(steps):
        nb = 1 if i == short_at else 2          # (an epoch's short last batch)
        x = torch.randn(nb, 4, 128, 128, generator=g).to(cuda)
        y = (torch.rand(nb, 128, 128, generator=g) < 0.3).long().to(cuda)
        for grp in opt.param_groups:
            grp['lr'] = 0.01 * (1.0 - i / 10.0)          # a schedule: the captured launch must follow it
        out = step(x, y)
        losses.append({k: float(v.detach()) for k, v in out.items()})
    torch.cuda.synchronize()
    if graphed:
        assert step.replays == steps - 3 - (short_at is not None), step.replays
        assert step.eager_fallbacks == (short_at is not None)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    bufs = [opt.state[p]['momentum_buffer'].clone() for p in m.parameters()]
    return losses, state, bufs


def test_graphed_step_is_bit_identical_to_eager(cuda):
    from ever_amd.hip import functional as F
    prev = F.set_conv_math('f16x2')
    try:
        le, se, be = _run(cuda, False)
        lg, sg, bg = _run(cuda, True)
    finally:
        F.set_conv_math(prev)
    assert le == lg, (le, lg)
    bad = [k for k in se if not torch.equal(se[k], sg[k])]
    assert not bad, bad[:5]
    assert all(torch.equal(a, b) for a, b in zip(be, bg))
    assert int(sg['en.resnet.bn1.num_batches_tracked']) == 7


def test_other_shapes_fall_back_to_one_eager_step(cuda):
    """a call with another batch size than the captured one runs eagerly (device learning rate in sync, the replays before
    and after it untouched): still bit-identical to the all-eager run"""
    le, se, be = _run(cuda, False, steps=7, short_at=5)
    lg, sg, bg = _run(cuda, True, steps=7, short_at=5)
    assert le == lg, (le, lg)
    assert not [k for k in se if not torch.equal(se[k], sg[k])]
    assert all(torch.equal(a, b) for a, b in zip(be, bg))


def test_graph_step_refuses_optimizers_with_host_side_schedules(cuda):
    import ever_amd as er
    from ever_amd.core.graph import GraphedTrainStep
    p = torch.nn.Parameter(torch.zeros(4, device=cuda))
    with pytest.raises(TypeError):
        GraphedTrainStep(lambda: {}, er.opt.FusedAdam([p], lr=1e-3))


def test_launcher_uses_the_graph_under_env_switch(cuda, tmp_path, monkeypatch):
    """EVK_GRAPH=1: the Launcher's loop (poly schedule, gradient clipping, logging) runs 7 iterations — 3 eager, 1 capture +
    replay, 3 replays — and lands on the very weights and logged losses of the eager loop."""
    import ever_amd as er
    from tests import plumbing_common as pc
    widths = (64, 128, 256, 512)
    loader = torch.utils.data.DataLoader(pc.ToyTiles(n=14), batch_size=2, shuffle=False)
    results = {}
    for name, env in (('eager', '0'), ('graph', '1')):
        monkeypatch.setenv('EVK_GRAPH', env)
        torch.manual_seed(3)
        m = er.module.FarSeg(dict(encoder=dict(resnet_type='resnet18', in_channels=4),
                                  head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                                            fs_relation=dict(scene_embedding_channels=512)))).to(cuda)
        sched = er.builder.make_learningrate(dict(type='poly', params=dict(base_lr=0.01, power=0.9, max_iters=7)))
        opt = er.builder.make_optimizer(er.AttrDict.from_dict(dict(type='sgd', params=dict(momentum=0.9, weight_decay=1e-4, lr=0.01),
                                                                   grad_clip=dict(max_norm=35, norm_type=2))),
                                        params=m.custom_param_groups())
        tl = er.Launcher(str(tmp_path / name), m, opt, sched)
        rec = []
        orig = tl._logger.train_log

        def spy(_rec=rec, _orig=orig, **kw):
            _rec.append({k: float(v) for k, v in kw['loss_dict'].items()})
            return _orig(**kw)
        tl._logger.train_log = spy
        tl.train_by_config(loader, config=er.AttrDict.from_dict(dict(num_iters=7, save_ckpt_interval_epoch=1000)))
        torch.cuda.synchronize()
        results[name] = (rec, {k: v.detach().clone() for k, v in m.state_dict().items()}, tl)
    assert results['graph'][2]._graph_step.replays == 4
    assert getattr(results['eager'][2], '_graph_step', None) is None
    assert len(results['eager'][0]) == 7 and results['eager'][0] == results['graph'][0], results
    bad = [k for k, v in results['eager'][1].items() if not torch.equal(v, results['graph'][1][k])]
    assert not bad, bad[:5]
