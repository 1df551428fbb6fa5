"""GPU: the HIP modules under torch DistributedDataParallel on the RCCL ("nccl") backend.
The box has one GPU, so the process group has world_size 1: this still exercises the full DDP path
(parameter broadcast, reducer hooks on the custom autograd Functions, bucket views over
channels_last gradients, the fused optimizer reading the bucket-view gradients)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_farseg_under_ddp_rccl_world1(cuda):
    import torch.distributed as dist
    import ever_amd as er
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29617')
    created = False
    if not dist.is_initialized():
        dist.init_process_group(backend='nccl', init_method='env://', rank=0, world_size=1)
        created = True
    try:
        torch.manual_seed(0)
        widths = (64, 128, 256, 512)
        cfg = dict(encoder=dict(resnet_type='resnet18', in_channels=4),
                   head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                             fs_relation=dict(scene_embedding_channels=512)))
        ref = er.module.FarSeg(cfg).to(cuda).train()
        ddp_model = er.module.FarSeg(cfg).to(cuda).train()
        ddp_model.load_state_dict(ref.state_dict())
        ddp = torch.nn.parallel.DistributedDataParallel(ddp_model, device_ids=[0], output_device=0, bucket_cap_mb=64,
                                                        gradient_as_bucket_view=True)
        x = torch.randn(2, 4, 128, 128, device=cuda)
        y = (torch.rand(2, 128, 128, device=cuda) < 0.3).long()
        opt_a = er.opt.FusedSGD(ref.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        opt_b = er.opt.FusedSGD(ddp_model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        for _ in range(2):
            la = ref(x, y)
            sum(la.values()).backward()
            opt_a.step()
            opt_a.zero_grad()
            lb = ddp(x, y)
            sum(lb.values()).backward()
            opt_b.step()
            opt_b.zero_grad()
            for k in la:
                assert torch.equal(la[k], lb[k]), k        # same kernels, same order: bit-identical
        for (k, p), (_, q) in zip(ref.named_parameters(), ddp_model.named_parameters()):
            assert torch.equal(p, q), k
    finally:
        if created:
            dist.destroy_process_group()


def test_sync_batchnorm_model_under_rccl_world1(cuda):
    """convert_sync_batchnorm over the whole FarSeg model, through the real collectives (world_size 1 on this
    box): with one rank SyncBatchNorm must reproduce BatchNorm (same statistics, fp64-merged), including the
    fused residual / ReLU call sites of the ResNet blocks and the running statistics."""
    import torch.distributed as dist
    import ever_amd as er
    from ever_amd.module.sync_bn import SyncBatchNorm, convert_sync_batchnorm
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29618')
    created = False
    if not dist.is_initialized():
        dist.init_process_group(backend='nccl', init_method='env://', rank=0, world_size=1)
        created = True
    try:
        torch.manual_seed(0)
        widths = (64, 128, 256, 512)
        cfg = dict(encoder=dict(resnet_type='resnet18', in_channels=4),
                   head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                             fs_relation=dict(scene_embedding_channels=512)))
        plain = er.module.FarSeg(cfg).to(cuda).train()
        synced = er.module.FarSeg(cfg)
        synced.load_state_dict(plain.state_dict())
        synced = convert_sync_batchnorm(synced).to(cuda).train()
        assert sum(isinstance(m, SyncBatchNorm) for m in synced.modules()) == sum(
            isinstance(m, torch.nn.BatchNorm2d) for m in plain.modules())
        assert list(synced.state_dict().keys()) == list(plain.state_dict().keys())
        x = torch.randn(2, 4, 128, 128, device=cuda)
        y = (torch.rand(2, 128, 128, device=cuda) < 0.3).long()
        la, lb = plain(x, y), synced(x, y)
        sum(la.values()).backward()
        sum(lb.values()).backward()
        for k in la:
            assert abs(la[k].item() - lb[k].item()) <= 1e-4 * abs(la[k].item()), (k, la[k].item(), lb[k].item())
        sa, sb = plain.state_dict(), synced.state_dict()
        for k in sa:
            if 'running_' in k:
                assert torch.allclose(sa[k], sb[k], rtol=1e-4, atol=1e-6), k
            if 'num_batches_tracked' in k:
                assert int(sa[k]) == int(sb[k]) == 1, k
        num = den = 0.0
        for (k, p), (_, q) in zip(plain.named_parameters(), synced.named_parameters()):
            num += float((p.grad.double() - q.grad.double()).pow(2).sum())
            den += float(p.grad.double().pow(2).sum())
        assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5   # fp32 vs fp64-merged statistics, ill-conditioned net
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize('side_stream', [False, True], ids=['one-stream', 'wgrad-side-stream'])
def test_flat_grad_ddp_world1_matches_plain_training(cuda, side_stream):
    """FlatGradDDP on the GPU (HIP pack kernel, bucket views as gradients, fused optimizer reading them; RCCL group of
    one rank): two training steps are bit-identical to the unwrapped model.  With the weight gradients on their side stream
    (hip/functional.py; the bucket pack then follows that stream) the wrapped model's BatchNorm backward takes the
    three-launch form while the unwrapped one — no reducer that knows the stream, so one stream — takes the one-launch form:
    equal to fp32 rounding instead of bit for bit."""
    import torch.distributed as dist
    import ever_amd as er
    from ever_amd.hip import functional as HF
    from ever_amd.trainer.grad_reducer import FlatGradDDP
    prev_stream = HF.set_wgrad_stream(side_stream)
    prev_shared = HF.set_wgrad_shared_split(False)   # (bit-for-bit comparison of the mechanism: launches split as if alone)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29619')
    created = False
    if not dist.is_initialized():
        dist.init_process_group(backend='nccl', init_method='env://', rank=0, world_size=1)
        created = True
    try:
        torch.manual_seed(0)
        widths = (64, 128, 256, 512)
        cfg = dict(encoder=dict(resnet_type='resnet18', in_channels=4),
                   head=dict(fpn=dict(in_channels_list=widths, out_channels=256),
                             fs_relation=dict(scene_embedding_channels=512)))
        ref = er.module.FarSeg(cfg).to(cuda).train()
        wrapped = er.module.FarSeg(cfg).to(cuda).train()
        wrapped.load_state_dict(ref.state_dict())
        ddp = FlatGradDDP(wrapped, bucket_cap_mb=16)
        assert len(ddp.buckets) >= 3
        x = torch.randn(2, 4, 128, 128, device=cuda)
        y = (torch.rand(2, 128, 128, device=cuda) < 0.3).long()
        opt_a = er.opt.FusedSGD(ref.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        opt_b = er.opt.FusedSGD(wrapped.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        strict = True
        for _ in range(2):
            la = ref(x, y)
            sum(la.values()).backward()
            opt_a.step()
            opt_a.zero_grad()
            lb = ddp(x, y)
            sum(lb.values()).backward()
            for p in wrapped.parameters():   # gradients are views into the flat buckets
                assert p.grad is not None and p.grad.stride() == p.stride()
            opt_b.step()
            opt_b.zero_grad()
            for k in la:
                assert torch.equal(la[k], lb[k]) or (not strict and float((la[k] - lb[k]).abs()) < 1e-5 * float(la[k].abs())), k

        strict = True     # (since ABI 20 there is ONE BatchNorm backward form: with or without the side stream, bit for bit)

        def same(a, b):
            if strict or not a.dtype.is_floating_point:
                return torch.equal(a, b)
            # (BatchNorm biases start at zero: after two steps they are ~1e-4 and the two BatchNorm backward forms leave them
            # 1e-7 apart; the strict child-process run below pins the form and compares bit for bit)
            return float((a.double() - b.double()).abs().max()) <= 1e-3 * float(b.double().abs().max()) + 2e-6
        for (k, p), (_, q) in zip(ref.named_parameters(), wrapped.named_parameters()):
            assert same(p, q), (k, side_stream, float((p.double() - q.double()).abs().max()), float(q.double().abs().max()),
                                dict(HF.wgrad_stream_stats))
        sa, sb = ref.state_dict(), wrapped.state_dict()
        for k in sa:
            assert same(sa[k], sb[k]), k       # running statistics live in the flat buffer tensor now
    finally:
        HF.set_wgrad_stream(prev_stream)
        HF.set_wgrad_shared_split(prev_shared)
        if created:
            dist.destroy_process_group()


def test_flat_grad_ddp_with_the_side_stream_is_bit_identical_when_batchnorm_is_pinned(cuda):
    """the same comparison in a child process with EVK_BN_FUSED=0 (read once per process): the unwrapped model on one stream
    and the wrapped one with its weight gradients on the side stream and the bucket packs following it — bit for bit"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_ddp_gpu.py'), '-x', '-q', '-k',
                          'test_flat_grad_ddp_world1_matches_plain_training and wgrad-side-stream'],
                         env=dict(os.environ, EVK_BN_FUSED='0', MASTER_PORT='29623'), capture_output=True, text=True, timeout=900,
                         cwd=root)
    assert out.returncode == 0 and '1 passed' in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]
