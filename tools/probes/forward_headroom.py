"""probe (GPU box): what does matrix-bound work cost BESIDE the forward pass, and beside the backward pass?  DESIGN 8 names one
lever that removes no work: the head's weight gradients (~3 ms, matrix-bound, no consumer inside the step) issued beside the NEXT
step's encoder forward, where the board is below its power cap.  This probe does not build that; it prices it: X = four
3x3x256 @128^2 weight gradients on dummy operands (the bf16x3 entry point: no scale words needed) on a second stream, (a) alone,
(b) forked at the start of the forward, (c) forked at the start of the backward, against the plain step.  The step's own weight
gradients stay where they are, so (c) - plain is what ADDITIONAL matrix work costs where the step already has some, and
(b) - plain what it costs where it has none."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, ever_amd as er
from ever_amd import _C
dev = torch.device('cuda:0')
torch.manual_seed(2333)
model, inputs, *_ = bench.make_workload(er, 'c2', dev, bench.BATCH, 0)
model = model.to(dev).train()
opt = er.opt.FusedSGD(model.parameters(), lr=0.007, momentum=0.9, weight_decay=1e-4)
lib = _C.load()
B, H, C = 16, 128, 256
d = _C.ConvDesc(B, H, H, C, H, H, C, 3, 3, 1, 1, 1, 1, 1, 1)
x, dy = torch.randn(B, H, H, C, device=dev), torch.randn(B, H, H, C, device=dev)
dw = torch.empty(C, 3, 3, C, device=dev)
ws_b = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d))
ws = torch.empty(ws_b, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream(dev)
NX = int(os.environ.get('NX', 4))


def extra():
    for _ in range(NX):
        _C.call('evk_conv2d_wgrad_x3', ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), ws_b, side.cuda_stream)


def step(where):
    main = torch.cuda.current_stream()
    if where == 'fwd':
        side.wait_stream(main); extra()
    out = model(*inputs)
    loss = sum(v for k, v in out.items() if k.endswith('loss'))
    if where == 'bwd':
        side.wait_stream(main); extra()
    loss.backward()
    if where:
        main.wait_stream(side)
    opt.step(); opt.zero_grad(set_to_none=True)


def run(where, n=20):
    for _ in range(3): step(where)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step(where)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for _ in range(3): extra()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): extra()
torch.cuda.synchronize()
alone = (time.perf_counter() - t) / 10 * 1e3
for r in range(2):
    p, f, b = run(None), run('fwd'), run('bwd')
    print(f'round {r}: X alone {alone:.2f} ms; step {p:.2f} ms; X beside the forward {f:.2f} (+{f - p:.2f}); X beside the backward {b:.2f} (+{b - p:.2f})')
