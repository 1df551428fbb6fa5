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
