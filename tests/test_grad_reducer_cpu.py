"""FlatGradDDP (ever_amd/trainer/grad_reducer.py) against torch DistributedDataParallel, world_size 2 on gloo:
same initial broadcast, same averaged gradients (bit-identical here: the same two addends per element), same
parameters after a few SGD steps, across several buckets, with a parameter that receives no gradient."""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(3, 8, 3, padding=1)
        self.bn = nn.BatchNorm2d(8)
        self.b = nn.Conv2d(8, 8, 3, padding=1)
        self.c = nn.Conv2d(8, 2, 1)
        self.unused = nn.Linear(4, 4)   # never receives a gradient

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.bn(self.a(x)))))).mean((2, 3))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ever_amd.trainer.grad_reducer import FlatGradDDP
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(100 + rank)            # replicas start different: the constructor must broadcast rank 0
    flat_net, ref_net = _Net(), _Net()
    ref_net.load_state_dict(flat_net.state_dict())
    flat = FlatGradDDP(flat_net, bucket_cap_mb=0.001)   # ~262 floats per bucket => many buckets
    ref = nn.parallel.DistributedDataParallel(ref_net, find_unused_parameters=True)
    assert len(flat.buckets) >= 3
    opt_f = torch.optim.SGD(flat_net.parameters(), lr=0.1, momentum=0.9)
    opt_r = torch.optim.SGD(ref_net.parameters(), lr=0.1, momentum=0.9)
    g = torch.Generator().manual_seed(7 + rank)   # every rank its own data shard
    for step in range(3):
        x = torch.randn(4, 3, 8, 8, generator=g)
        t = torch.randn(4, 2, generator=g)
        for net, opt in ((flat, opt_f), (ref, opt_r)):
            opt.zero_grad(set_to_none=True)
            ((net(x) - t) ** 2).mean().backward()
        for (k, p), (_, r) in zip(flat_net.named_parameters(), ref_net.named_parameters()):
            if r.grad is None:
                assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k
            else:
                assert torch.equal(p.grad, r.grad), (step, k)
        opt_f.step()
        opt_r.step()
    same = all(torch.equal(p, r) for p, r in zip(flat_net.state_dict().values(), ref_net.state_dict().values()))
    # parameters only: BatchNorm running statistics are rank 0's at the START of a forward (DDP semantics) and
    # then absorb the rank's own batch
    digest = float(sum(p.double().sum() for p in flat_net.parameters()))
    q.put((rank, same, digest))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_ddp_equals_torch_ddp_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], 'FlatGradDDP and torch DDP diverged'
    assert res[0][2] == res[1][2], 'replicas diverged across ranks'


def test_the_bucket_that_closes_last_is_small():
    """Bucket layout (no process group: world 1): cut from the front of the registration order, reduced back to front; the
    last-closing bucket — the first-registered parameters, whose all-reduce starts when the backward pass ends and nothing
    hides — stays under tail_caps_mb[0], the one before it under tail_caps_mb[1], the others under the cap; every
    parameter sits in exactly one bucket and buckets follow the readiness order.  Sizes of a FarSeg-R50-like profile."""
    sys.path.insert(0, ROOT)
    from ever_amd.trainer.grad_reducer import FlatGradDDP

    class _Prof(nn.Module):
        def __init__(self):
            super().__init__()
            # floats per "stage" in registration order (stem, layer1..4, head), as 1-D parameters
            for i, n in enumerate([9_408, 215_808, 1_219_584, 7_098_368, 14_964_736, 8_000_000]):
                for j in range(4):
                    self.register_parameter(f's{i}_{j}', nn.Parameter(torch.zeros(n // 4)))

    m = _Prof()
    flat = FlatGradDDP(m, bucket_cap_mb=64)
    sizes_mb = [sum(p.numel() for p in b.params) * 4 / 2 ** 20 for b in flat.buckets]
    assert len(flat.buckets) >= 3, sizes_mb
    assert sizes_mb[-1] <= 4.0 + 1e-6, sizes_mb            # closes last: exposed
    assert sizes_mb[-2] <= 32.0 + 1e-6, sizes_mb
    assert all(s <= 64.0 + 1e-6 for s in sizes_mb), sizes_mb
    params = list(m.parameters())
    order = [p for b in flat.buckets for p in b.params]
    assert len(order) == len(params) and {id(p) for p in order} == {id(p) for p in params}
    assert [id(p) for p in order] == [id(p) for p in reversed(params)]      # readiness order, bucket after bucket
    # a uniform cap (the layout until round 4) for comparison: its last bucket held everything below layer 4
    uniform = FlatGradDDP(_Prof(), bucket_cap_mb=64, tail_caps_mb=())
    last_mb = sum(p.numel() for p in uniform.buckets[-1].params) * 4 / 2 ** 20
    assert last_mb > 8 * sizes_mb[-1], (last_mb, sizes_mb)
