"""The head's pyramid levels on two streams (hip/streams.py: HeadBranches, EVK_HEAD_BRANCH=1 — measured level to
-0.9 % on the step and therefore off by default; reference fs_relation.py:56-73,
fpn.py:183-189 — the per-level relation and decoder branches are independent between the FPN and the decoder's mean).
It must be invisible: losses, first-step gradients and trained weights bit for bit those of the plain order, for the FarSeg
head (FSRelation, commuted classifier), the FarSeg++ head (FSRelationV2 + projection) and the decoder's `features` path;
really on a second stream (launch counts); off under observers."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(cuda, pp=False):
    import ever_amd as er
    torch.manual_seed(13)
    widths = (64, 128, 256, 512)
    cls = er.module.FarSegPP if pp else er.module.FarSeg
    return cls(dict(encoder=dict(resnet_type='resnet18', in_channels=4),
                    head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                              fs_relation=dict(scene_embedding_channels=512)))).to(cuda).train()


def _train(cuda, on, pp=False, steps=3):
    import ever_amd as er
    from ever_amd.hip import functional as HF
    prev = HF.set_head_branch(on)
    try:
        m = _model(cuda, pp)
        opt = er.opt.FusedSGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        g = torch.Generator().manual_seed(2)
        grads, losses = None, []
        for i in range(steps):
            x = torch.randn(2, 4, 128, 128, generator=g).to(cuda)
            y = (torch.rand(2, 128, 128, generator=g) < 0.3).long().to(cuda)
            out = m(x, y)
            losses.append({k: v.detach().clone() for k, v in out.items()})
            sum(out.values()).backward()
            if i == 0:
                torch.cuda.synchronize()
                grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
            opt.fused_clip(max_norm=35)
            opt.step()
            opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        return losses, grads, {k: v.detach().clone() for k, v in m.state_dict().items()}
    finally:
        HF.set_head_branch(prev)


@pytest.mark.parametrize('pp', [False, True], ids=['farseg', 'farsegpp'])
def test_training_is_bit_identical_with_and_without_the_branch_stream(cuda, pp):
    l0, g0, s0 = _train(cuda, False, pp)
    l1, g1, s1 = _train(cuda, True, pp)
    for a, b in zip(l0, l1):
        assert all(torch.equal(a[k], b[k]) for k in a), (a, b)
    assert all(torch.equal(g1[k], g0[k]) for k in g0), [k for k in g0 if not torch.equal(g1[k], g0[k])][:5]
    assert all(torch.equal(s1[k], s0[k]) for k in s0), [k for k in s0 if not torch.equal(s1[k], s0[k])][:5]


def test_levels_really_run_on_the_branch_stream(cuda):
    """the session hands out a second stream, levels 1.. switch torch's current stream to it and level 0 does not"""
    from ever_amd.hip import functional as HF
    prev = HF.set_head_branch(True)
    try:
        t = torch.zeros(4, device=cuda)
        br = HF.head_branches(t)
        if br is None:
            pytest.skip('no second hardware queue on this box (the head then runs in plain order)')
        main = torch.cuda.current_stream().cuda_stream
        with br.level(0):
            assert torch.cuda.current_stream().cuda_stream == main and not br.forked
        with br.level(2):
            assert torch.cuda.current_stream().cuda_stream == br.side.cuda_stream != main and br.forked
            a = torch.ones(1 << 20, device=cuda) * 3.0
        assert torch.cuda.current_stream().cuda_stream == main
        br.join()
        assert not br.forked
        assert float((a + 1.0).sum().item()) == 4.0 * (1 << 20)     # ordered behind the branch stream by the join
    finally:
        HF.set_head_branch(prev)


def test_features_path_and_eval_forward_equal_plain_order(cuda):
    from ever_amd.hip import functional as HF
    x = torch.randn(2, 4, 128, 128, device=cuda)
    outs = []
    for on in (False, True):
        m = _model(cuda)          # (same seed: same weights and running statistics in both rounds)
        prev = HF.set_head_branch(on)
        try:
            with torch.no_grad():
                feats = m.en(x)
                outs.append((m.head.features(feats).clone(), m.head(feats).clone()))
            m.eval()
            with torch.no_grad():
                outs[-1] += (m(x).clone(),)
            m.train()
        finally:
            HF.set_head_branch(prev)
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_off_while_a_global_forward_hook_is_installed(cuda):
    from ever_amd.hip import functional as HF
    prev = HF.set_head_branch(True)
    h = torch.nn.modules.module.register_module_forward_hook(lambda mod, i, o: None)
    try:
        assert HF.head_branches(torch.zeros(4, device=cuda)) is None
    finally:
        h.remove()
        HF.set_head_branch(prev)
