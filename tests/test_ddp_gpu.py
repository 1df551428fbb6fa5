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


def test_flat_grad_ddp_world1_matches_plain_training(cuda):
    """FlatGradDDP on the GPU (HIP pack kernel, bucket views as gradients, fused optimizer reading them; RCCL group of
    one rank): two training steps are bit-identical to the unwrapped model."""
    import torch.distributed as dist
    import ever_amd as er
    from ever_amd.trainer.grad_reducer import FlatGradDDP
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
                assert torch.equal(la[k], lb[k]), k
        for (k, p), (_, q) in zip(ref.named_parameters(), wrapped.named_parameters()):
            assert torch.equal(p, q), k
        sa, sb = ref.state_dict(), wrapped.state_dict()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k       # running statistics live in the flat buffer tensor now
    finally:
        if created:
            dist.destroy_process_group()
