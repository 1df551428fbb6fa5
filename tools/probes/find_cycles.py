"""dev probe (GPU box): what the cyclic GC finds after one training step (reference cycles keep activations alive until a
generation-2 collection: the caching allocator then has to grow — hipMalloc inside the step)."""
import collections, gc, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ever_amd as er
from ever_amd import _C
import bench
dev = torch.device('cuda:0'); torch.cuda.set_device(dev); _C.load()
torch.manual_seed(2333)
model = er.module.FarSeg(dict()).to(dev).train()
opt = er.opt.FusedSGD(model.parameters(), lr=0.007, momentum=0.9, weight_decay=1e-4)
x, y = bench.make_batch(dev, 16, 0)
def step():
    out = model(x, y)
    sum(v for k, v in out.items() if k.endswith('loss')).backward()
    opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(2): step()
torch.cuda.synchronize(); gc.collect()
a0 = torch.cuda.memory_allocated()
gc.disable()
step(); torch.cuda.synchronize()
a1 = torch.cuda.memory_allocated()
print(f'allocated before / after one step without GC: {a0 / 2**30:.2f} / {a1 / 2**30:.2f} GB')
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
print('unreachable objects:', n)
hist = collections.Counter(type(o).__name__ for o in gc.garbage)
print(hist.most_common(15))
tens = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
print('tensors in garbage:', len(tens), 'bytes', sum(t.numel() * t.element_size() for t in tens) / 2**30, 'GB')
for t in tens[:6]:
    refs = [type(r).__name__ for r in gc.get_referrers(t) if r is not tens and r is not gc.garbage][:6]
    print(tuple(t.shape), t.dtype, 'grad_fn', type(t.grad_fn).__name__ if t.grad_fn is not None else None,
          'attrs', [k for k in vars(t)] if hasattr(t, '__dict__') else None, 'referrers', refs)
others = [o for o in gc.garbage if not isinstance(o, torch.Tensor)]
for o in others[:12]:
    s = repr(o)
    print(type(o).__name__, s[:160].replace('\n', ' '))
