"""diagnostic (GPU box, 2 ranks on cuda:0 over gloo): first-step gradients of FlatGradDDP and torch DDP against the hand-averaged ones"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.world2_gpu_worker import R18, _data
rank = int(os.environ['RANK']); torch.cuda.set_device(0); dev = torch.device('cuda:0')
dist.init_process_group('gloo', init_method='env://', rank=rank, world_size=2)
import ever_amd as er
from ever_amd.trainer.grad_reducer import FlatGradDDP
torch.manual_seed(7)
a = er.module.FarSeg(R18).to(dev).train(); b = er.module.FarSeg(R18).to(dev).train(); c = er.module.FarSeg(R18).to(dev).train()
b.load_state_dict(a.state_dict()); c.load_state_dict(a.state_dict())
flat = FlatGradDDP(a, bucket_cap_mb=16)
tddp = torch.nn.parallel.DistributedDataParallel(b, device_ids=[0], output_device=0, bucket_cap_mb=16, gradient_as_bucket_view=True)
x, y = _data(rank, dev)
for it in range(3):
    for m in (a, b, c):
        for p in m.parameters(): p.grad = None
    sum(flat(x, y).values()).backward()
    sum(tddp(x, y).values()).backward()
    sum(c(x, y).values()).backward()
    torch.cuda.synchronize()
    worst = {}
    for (k, pa), (_, pb), (_, pc) in zip(a.named_parameters(), b.named_parameters(), c.named_parameters()):
        g = pc.grad.detach().clone(); dist.all_reduce(g); g /= 2
        ea = float((pa.grad - g).abs().max() / g.abs().max().clamp_min(1e-30)); eb = float((pb.grad - g).abs().max() / g.abs().max().clamp_min(1e-30))
        if ea > 1e-4 or eb > 1e-4: worst[k] = (round(ea, 5), round(eb, 5))
    print(f'rank {rank} iter {it}: params off (flat, torch): {dict(list(worst.items())[:6])}', flush=True)
dist.barrier(); dist.destroy_process_group()
