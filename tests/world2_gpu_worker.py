"""One of TWO ranks sharing cuda:0: the HIP exchange path at a real world size of 2.

RCCL refuses two ranks on one device, so the process group is gloo over DEVICE tensors (gloo stages them through the host;
everything on this package's side of the collective — evk_pack_multi, the communication stream, the weight-gradient side
stream's hand-off, SyncBatchNorm's staged kernels, the dice statistics buffers — is the product path).  Two processes
sharing the device was also exactly the co-residency hazard of the one-launch BatchNorm backward (removed in ABI 20).

Reference: ever/trainer/th_ddp_trainer.py:13-30 (env:// group, DDP wrap), ever/module/loss.py:20-23,46-48 (dice statistics
all-reduce), ever/core/launcher.py:196,317-321 (forward_times without no_sync).

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/world2_gpu_worker.py OUT_DIR [case ...]
writes OUT_DIR/rank{r}.json = {case: 'ok' | 'FAIL: ...'}."""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

R18 = dict(encoder=dict(resnet_type='resnet18', in_channels=4),
           head=dict(fpn=dict(in_channels_list=(64, 128, 256, 512), out_channels=256),
                     fs_relation=dict(scene_embedding_channels=512)))


def _data(rank, dev, n=2, hw=128, seed=100):
    g = torch.Generator().manual_seed(seed + rank)
    x = torch.randn(n, 4, hw, hw, generator=g)
    y = (torch.rand(n, hw, hw, generator=g) < 0.3).long()
    y[:, :4, :4] = 255
    return x.to(dev), y.to(dev)


def _same_on_all_ranks(t, what):
    """bit-equality of a device tensor across the two ranks"""
    mine = t.detach().float().contiguous().view(-1).clone()
    other = mine.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(mine, other), f'{what}: the ranks differ (max {float((mine - other).abs().max()):.3e})'


def case_flat_equals_torch_ddp(rank, dev):
    """FarSeg-R18, three SGD steps, different data per rank: FlatGradDDP (HIP pack launch per bucket, communication stream,
    weight gradients on the side stream) ends bit-identical to torch DistributedDataParallel on the same group, and the two
    ranks end bit-identical to each other."""
    import ever_amd as er
    from ever_amd.hip import functional as HF
    from ever_amd.trainer.grad_reducer import FlatGradDDP
    assert HF.wgrad_stream_enabled()
    torch.manual_seed(7)             # same initial weights on both ranks and both wrappers
    a = er.module.FarSeg(R18).to(dev).train()
    b = er.module.FarSeg(R18).to(dev).train()
    b.load_state_dict(a.state_dict())
    flat = FlatGradDDP(a, bucket_cap_mb=16)
    assert len(flat.buckets) >= 3 and flat.world == 2
    tddp = torch.nn.parallel.DistributedDataParallel(b, device_ids=[0], output_device=0, bucket_cap_mb=16,
                                                     gradient_as_bucket_view=True)
    oa = er.opt.FusedSGD(a.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    ob = er.opt.FusedSGD(b.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    x, y = _data(rank, dev)
    HF.wgrad_stream_stats['side'] = HF.wgrad_stream_stats['main'] = 0
    for step in range(3):
        la = flat(x, y)
        sum(la.values()).backward()
        oa.step()
        oa.zero_grad(set_to_none=True)
        side_after_flat = HF.wgrad_stream_stats['side']
        lb = tddp(x, y)
        sum(lb.values()).backward()
        ob.step()
        ob.zero_grad(set_to_none=True)
        assert HF.wgrad_stream_stats['side'] == side_after_flat, 'torch DDP step used the side stream'
        for k in la:
            assert torch.equal(la[k], lb[k]), (step, k, float(la[k]), float(lb[k]))
    assert HF.wgrad_stream_stats['side'] > 0, 'FlatGradDDP steps never used the weight-gradient side stream'
    torch.cuda.synchronize()
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q), f'{k}: FlatGradDDP != torch DDP (max {float((p - q).abs().max()):.3e})'
    for (k, p), (_, q) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.equal(p, q), f'buffer {k}'
    _same_on_all_ranks(torch.cat([p.detach().reshape(-1) for p in a.parameters()]), 'parameters after 3 steps')
    # the two ranks saw different data: the averaged gradient is not either rank's own
    return f"side-stream weight gradients: {HF.wgrad_stream_stats['side']}"


def case_forward_times_2(rank, dev):
    """Gradient accumulation without no_sync (reference launcher.py:196,317-321): two micro-batches, each backward
    all-reduces; the second accumulates into the bucket view of the first and packs p.grad onto itself.  Result =
    avg_ranks(g1) + avg_ranks(g2), checked against gradients computed without any wrapper and averaged by hand."""
    import ever_amd as er
    from ever_amd.trainer.grad_reducer import FlatGradDDP
    torch.manual_seed(11)
    a = er.module.FarSeg(R18).to(dev).train()
    b = er.module.FarSeg(R18).to(dev).train()
    b.load_state_dict(a.state_dict())
    flat = FlatGradDDP(a, bucket_cap_mb=16, broadcast_buffers=False)
    mbs = [_data(rank, dev, seed=300), _data(rank, dev, seed=400)]
    for x, y in mbs:
        out = flat(x, y)
        (sum(out.values()) / 2).backward()
    # by hand: the unwrapped twin with the same BatchNorm buffers history
    want = None
    for x, y in mbs:
        out = b(x, y)
        (sum(out.values()) / 2).backward()
        gs = torch.cat([p.grad.detach().reshape(-1) for p in b.parameters()])
        for p in b.parameters():
            p.grad = None
        dist.all_reduce(gs)
        gs /= 2
        want = gs if want is None else want + gs
    got = torch.cat([p.grad.detach().reshape(-1) for p in a.parameters()])
    torch.cuda.synchronize()
    err = float((got - want).norm() / want.norm())
    assert err < 2e-6, f'accumulated gradient off by {err:.2e}'
    _same_on_all_ranks(got, 'accumulated gradients')
    return f'rel L2 {err:.1e}'


def case_sync_bn_two_ranks(rank, dev):
    """SyncBatchNorm with two REAL ranks (all_gather of (mean, M2, count) forward, all_reduce of (sum g, sum g xhat)
    backward) == BatchNorm over the concatenated batch in one process: output, input gradient, running statistics; weight /
    bias gradients are the rank-local sums (DDP averages them)."""
    from ever_amd.module import layers
    from ever_amd.module.sync_bn import SyncBatchNorm
    g = torch.Generator().manual_seed(21)
    c = 64
    full = (torch.randn(6, c, 24, 20, generator=g) * 2 + 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    up = torch.randn(6, c, 24, 20, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    sl = slice(0, 2) if rank == 0 else slice(2, 6)     # unequal shares
    bn = layers.BatchNorm2d(c).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=None)
        bn.bias.uniform_(-0.5, 0.5)
    dist.broadcast(bn.weight.data, 0)
    dist.broadcast(bn.bias.data, 0)
    sbn = SyncBatchNorm(c).to(dev).train()
    sbn.load_state_dict(bn.state_dict())
    xf = full.clone().requires_grad_()
    yf = bn(xf, relu=True)
    yf.backward(up)
    xs = full[sl].clone().contiguous(memory_format=torch.channels_last).requires_grad_()
    ys = sbn(xs, relu=True)
    ys.backward(up[sl].contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()

    def close(a, b, tol, what):
        e = float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))
        assert e < tol, f'{what}: {e:.2e}'
    close(ys, yf[sl], 1e-5, 'output')
    close(xs.grad, xf.grad[sl], 1e-4, 'input gradient')
    close(sbn.running_mean, bn.running_mean, 1e-5, 'running_mean')
    close(sbn.running_var, bn.running_var, 1e-5, 'running_var')
    # parameter gradients: local sums; their SUM over ranks is the full-batch gradient
    gw, gb = sbn.weight.grad.clone(), sbn.bias.grad.clone()
    dist.all_reduce(gw)
    dist.all_reduce(gb)
    close(gw, bn.weight.grad, 1e-4, 'weight gradient (summed over ranks)')
    close(gb, bn.bias.grad, 1e-4, 'bias gradient (summed over ranks)')
    return 'ok'


def case_dice_two_ranks(rank, dev):
    """dice statistics all-reduced across two real ranks before the ratio (reference loss.py:20-23,46-48): each rank reports
    the dice loss of the UNION of the pixels and d loss / d logits of its own pixels times the world size."""
    from ever_amd.hip import functional as HF
    from oracle import farseg_ref
    out = []
    for c in (1, 5):
        g = torch.Generator().manual_seed(5 + c)
        z = torch.randn(5, c, 12, 10, generator=g) * 2
        y = torch.randint(0, max(c, 2), (5, 12, 10), generator=g)
        y[0, :3, :4] = 255
        sl = slice(0, 2) if rank == 0 else slice(2, 5)
        zr = z.double().requires_grad_()
        ref = farseg_ref.dice_ref(zr, y)
        ref.backward()
        zi = z[sl].to(dev).requires_grad_()
        li = HF.dice_loss_with_logits(zi, y[sl].to(dev))
        li.backward()
        assert abs(li.item() - ref.item()) <= 1e-6 * abs(ref.item()) + 1e-7, (c, li.item(), ref.item())
        ga, gb = zi.grad.cpu().contiguous().double().numpy(), 2.0 * zr.grad[sl].numpy()
        assert np.abs(ga - gb).max() <= 1e-4 * np.abs(gb).max(), (c, np.abs(ga - gb).max(), np.abs(gb).max())
        out.append(round(li.item(), 6))
    return f'dice {out}'


def case_trainer_three_steps(rank, dev):
    """THDDPTrainer -> FlatGradDDP -> Launcher, FarSeg-R18 through the registry, three iterations with forward_times=2 on a
    toy loader sharded by StepDistributedSampler: runs (no hang with two processes on one device), logs finite losses, the
    replicas end bit-identical."""
    import tempfile
    import ever_amd as er
    from ever_amd.hip import functional as HF
    from tests.plumbing_common import ToyTilesLoader  # noqa: F401  (registers the loader)
    work = tempfile.mkdtemp(prefix=f'w2_rank{rank}_')
    cfg_path = os.path.join(work, 'cfg.py')
    with open(cfg_path, 'w') as f:
        f.write('''
config = dict(
    model=dict(type='FarSeg', params=dict(encoder=dict(resnet_type='resnet18', in_channels=4),
               head=dict(fpn=dict(in_channels_list=(64, 128, 256, 512), out_channels=256),
                         fs_relation=dict(scene_embedding_channels=512)))),
    data=dict(train=dict(type='ToyTilesLoader', params=dict(n=16, c=4, hw=64, batch_size=2, distributed=True))),
    optimizer=dict(type='sgd', params=dict(momentum=0.9, weight_decay=1e-4), grad_clip=dict(max_norm=35, norm_type=2)),
    learning_rate=dict(type='poly', params=dict(base_lr=0.01, power=0.9, max_iters=3)),
    train=dict(forward_times=2, num_iters=3, distributed=True, log_interval_step=1, save_ckpt_interval_epoch=1000),
    test=dict(),
)
''')
    os.environ['LOCAL_RANK'] = '0'          # both ranks on cuda:0
    torch.manual_seed(3)
    trainer = er.trainer.get_trainer('th_ddp', argv=['--config_path', cfg_path, '--model_dir', os.path.join(work, 'run')])
    assert type(trainer).__name__ == 'THDDPTrainer'
    out = trainer.run()
    tl = out['launcher']
    assert tl.global_step == 3
    flat = torch.cat([p.detach().reshape(-1) for p in tl.unwrapped_model.parameters()])
    assert bool(torch.isfinite(flat).all())
    _same_on_all_ranks(flat, 'trainer replicas')
    return 'ok'


CASES = dict(flat_equals_torch_ddp=case_flat_equals_torch_ddp, forward_times_2=case_forward_times_2,
             sync_bn_two_ranks=case_sync_bn_two_ranks, dice_two_ranks=case_dice_two_ranks,
             trainer_three_steps=case_trainer_three_steps)


def main():
    out_dir = sys.argv[1]
    names = sys.argv[2:] or list(CASES)
    rank = int(os.environ['RANK'])
    torch.cuda.set_device(0)
    dev = torch.device('cuda:0')
    import datetime
    dist.init_process_group(backend='gloo', init_method='env://', rank=rank, world_size=2,
                            timeout=datetime.timedelta(seconds=240))   # a rank that died must not hang its peer
    res = {}
    for n in names:
        try:
            dist.barrier()
            res[n] = 'ok: ' + str(CASES[n](rank, dev))
        except Exception:
            res[n] = 'FAIL: ' + traceback.format_exc()[-1800:]
        torch.cuda.synchronize()
        with open(os.path.join(out_dir, f'rank{rank}.json'), 'w') as f:
            json.dump(res, f, indent=1)
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        pass


if __name__ == '__main__':
    main()
