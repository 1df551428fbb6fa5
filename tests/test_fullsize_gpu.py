"""Size-independent properties at the BASELINE workload size (FarSeg-R50, 3x512x512, batch 16), where the CPU
oracle would take minutes per evaluation:
  * per-sample independence in eval mode (BatchNorm frozen): a tile's logits do not depend on its batch mates;
  * linearity of the backward pass in the batch: with frozen statistics and a mean-reduced BCE over equally
    sized halves, grad(batch) = (grad(half A) + grad(half B)) / 2  — exercises every conv forward / data-gradient /
    weight-gradient launch (tile shapes, split-K plans, wave-specialised kernels) at the sizes bench.py times;
  * a training step is finite and its BatchNorm statistics match a two-pass fp64 evaluation of the stem output."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(cuda, **loss):
    import ever_amd as er
    torch.manual_seed(2333)
    cfg = dict(loss=loss) if loss else dict()
    return er.module.FarSeg(cfg).to(cuda)


def _batch(cuda, n=16):
    g = torch.Generator(device='cpu').manual_seed(99)
    x = torch.randn(n, 3, 512, 512, generator=g).to(cuda)
    y = (torch.rand(n, 512, 512, generator=g) < 0.3).long()
    y[:, :8, :8] = 255
    return x, y.to(cuda)


def test_eval_logits_are_per_sample_independent_at_full_size(cuda, conv_math):
    m = _model(cuda).eval()
    x, _ = _batch(cuda)
    with torch.no_grad():
        full = m(x)                       # sigmoid probabilities [16, 1, 512, 512]
        alone = m(x[5:6])
        pair = m(x[4:6])
    ref = full[5:6]
    scale = float(ref.abs().max())
    # different batch sizes select different kernels (tile shapes, the LDS-halo 3x3 kernel above a grid size):
    # the accumulation order changes, the values agree to rounding — two orders below the 1e-3 contract
    e1, e2 = float((alone - ref).abs().max()) / scale, float((pair[1:2] - ref).abs().max()) / scale
    print(f'per-sample independence: rel diff {e1:.2e} (alone), {e2:.2e} (pair)')
    # exact-fp32 kernels all accumulate in the same order => identical values.  In the split arithmetic the
    # LDS-halo 3x3 kernel (used above a grid size, i.e. for the big batch only) sums channel-chunk-major instead of
    # tap-major: the two evaluations differ by per-layer rounding (~2e-6) times this random-init network's
    # conditioning (measured 2..6e-4 at the probabilities); the contract is 1e-3.
    tol = 1e-6 if conv_math == 'f32' else 1e-3
    assert e1 <= tol and e2 <= tol, (e1, e2)


def test_backward_is_linear_in_the_batch_with_frozen_statistics(cuda, conv_math):
    m = _model(cuda)
    m.eval()                              # BatchNorm uses running statistics: every sample is independent
    for p in m.parameters():
        p.requires_grad_(True)
    x, y = _batch(cuda)

    def grads(xs, ys):
        m.zero_grad(set_to_none=True)
        from ever_amd.hip import functional as HF
        from ever_amd.module import loss as L
        logits = m.head(m.en(HF.as_nhwc(xs)))
        L.binary_cross_entropy_with_logits(logits, ys).backward()
        return [p.grad.detach().clone() for p in m.parameters()]

    g_all = grads(x, y)
    g_a = grads(x[:8], y[:8])
    g_b = grads(x[8:], y[8:])
    # equal numbers of valid pixels in both halves => the batch mean is the mean of the halves' means
    assert int((y[:8] != 255).sum()) == int((y[8:] != 255).sum())
    num = den = 0.0
    worst = 0.0
    for ga, gb, gf in zip(g_a, g_b, g_all):
        comb = 0.5 * (ga.double() + gb.double())
        d = float((comb - gf.double()).norm())
        s = float(gf.double().norm())
        num += d * d
        den += s * s
        if s > 1e-8:
            worst = max(worst, d / s)
    # f32: same kernels' accumulation order for every batch size => linear to rounding.  bf16x3: the halves run the
    # tap-major kernels where the full batch runs the LDS-halo kernel; ReLU decisions that flip on a rounding
    # difference make the gradients of this random-init network agree only at its conditioning (the fp32 CPU
    # oracle is 2-3 % from its own fp64 evaluation, test_e2e_gpu.py), a bug would show as O(1).
    glob = (num / den) ** 0.5
    print(f'batch linearity: global rel L2 {glob:.2e}, worst tensor {worst:.2e}')
    if conv_math == 'f32':
        assert glob < 2e-5 and worst < 5e-4, (glob, worst)
    else:
        assert glob < 5e-2, glob


def test_training_step_at_full_size_is_finite_and_stem_statistics_match_fp64(cuda):
    import ever_amd as er
    m = _model(cuda).train()
    x, y = _batch(cuda)
    losses = m(x, y)
    sum(losses.values()).backward()
    assert all(torch.isfinite(v) for v in losses.values())
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    # BatchNorm running statistics of the stem after one step = 0.9*init + 0.1*batch statistics of conv1(x)
    stem_conv, stem_bn = m.en.resnet.conv1, m.en.resnet.bn1
    with torch.no_grad():
        from ever_amd.hip import functional as HF
        z = stem_conv(HF.as_nhwc(x)).double()
        mean = z.mean((0, 2, 3))
        var = z.var((0, 2, 3), unbiased=True)
    sd = m.state_dict()
    rm = sd['en.resnet.bn1.running_mean'].double()
    rv = sd['en.resnet.bn1.running_var'].double()
    assert float((rm - 0.1 * mean).abs().max()) <= 1e-5 * float(mean.abs().max() + 1e-3)
    assert float((rv - (0.9 + 0.1 * var)).abs().max()) <= 1e-5 * float(var.abs().max() + 1.0)


def test_config_c3_farsegpp_training_step_and_backward_linearity(cuda):
    """BASELINE configuration C3 on one GPU as stated: FarSeg++ (ResNet-50 + FPN + FSRelationV2 + decoder), 4-band
    1024x1024 tiles, batch 8 (maps of 256^2 .. 32^2: twice the GEMM rows of the bench workload; the 4-band stem through
    the channel-padded fp32 path; GroupNorm scene MLPs, channel concat, 512->256 projection, Dropout2d in the relation
    module).  A training step is finite, the stem statistics match fp64, and with frozen statistics (and Dropout2d off)
    the backward is linear in the batch."""
    import ever_amd as er
    from ever_amd.module.fs_relation import FSRelationV2
    torch.manual_seed(7)
    m = er.module.FarSegPP(dict(encoder=dict(in_channels=4))).to(cuda).train()
    assert isinstance(m.head.fs_relation, FSRelationV2)
    g = torch.Generator(device='cpu').manual_seed(123)
    x = torch.randn(8, 4, 1024, 1024, generator=g).to(cuda)
    y = (torch.rand(8, 1024, 1024, generator=g) < 0.3).long()
    y[:, :8, :8] = 255
    y = y.to(cuda)
    losses = m(x, y)
    sum(losses.values()).backward()
    assert all(torch.isfinite(v) for v in losses.values())
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    with torch.no_grad():
        from ever_amd.hip import functional as HF
        z = m.en.resnet.conv1(HF.as_nhwc(x)).double()
        mean, var = z.mean((0, 2, 3)), z.var((0, 2, 3), unbiased=True)
        del z
    sd = m.state_dict()
    assert float((sd['en.resnet.bn1.running_mean'].double() - 0.1 * mean).abs().max()) <= 1e-5 * float(mean.abs().max() + 1e-3)
    assert float((sd['en.resnet.bn1.running_var'].double() - (0.9 + 0.1 * var)).abs().max()) <= 1e-5 * float(var.abs().max() + 1.0)
    # frozen statistics (eval-mode BatchNorm, gradients still flow): grad(batch) = mean of the halves' gradients
    m.zero_grad(set_to_none=True)
    m.eval()
    from ever_amd.module import loss as L

    def grads(xs, ys):
        m.zero_grad(set_to_none=True)
        lg = m.head(m.en(xs))
        L.binary_cross_entropy_with_logits(lg, ys).backward()
        return [p.grad.double().clone() for p in m.parameters()]
    full = grads(x, y)
    ga, gb = grads(x[:4], y[:4]), grads(x[4:], y[4:])
    num = sum(float(((a + b) / 2 - f).square().sum()) for a, b, f in zip(ga, gb, full))
    den = sum(float(f.square().sum()) for f in full)
    rel = (num / den) ** 0.5
    print(f'C3-size backward linearity: global relative L2 {rel:.2e}')
    assert rel < 5e-2, rel
