"""dev tool: one-tap (1x1) convolutions through the C-ABI against torch CPU fp64 — forward and data gradient, strides 1/2,
ragged M and Cout, bias/ReLU epilogue — under the current EVK_X3_DMA* switches."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd import _C
from ever_amd.hip import functional as HF
dev = torch.device('cuda:0')
torch.manual_seed(0)
worst = 0.0
for (n, h, w, cin, cout, stride, bias, relu) in [(2, 16, 16, 64, 128, 1, False, False), (3, 17, 13, 32, 72, 1, True, True),
                                                 (2, 32, 32, 256, 64, 1, False, False), (2, 32, 32, 128, 256, 2, False, False),
                                                 (1, 9, 7, 96, 40, 1, True, False), (4, 64, 64, 64, 256, 1, False, False),
                                                 (2, 30, 30, 512, 128, 2, True, False)]:
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, 1, 1) * 0.1
    b = torch.randn(cout) if bias else None
    xr = x.double().requires_grad_()
    yr = torch.nn.functional.conv2d(xr, wt.double(), None if b is None else b.double(), stride=stride)
    if relu:
        yr = torch.relu(yr)
    g = torch.randn_like(yr)
    yr.backward(g)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    wg = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = HF.conv2d(xg, wg, None if b is None else b.to(dev), stride=stride, relu=relu)
    y.backward(g.float().to(dev).contiguous(memory_format=torch.channels_last))
    e1 = float((y.detach().cpu().double() - yr.detach()).abs().max() / yr.detach().abs().max())
    e2 = float((xg.grad.cpu().double() - xr.grad).abs().max() / xr.grad.abs().max())
    worst = max(worst, e1, e2)
    print(f'n{n} {h}x{w} {cin}->{cout} s{stride} bias{int(bias)} relu{int(relu)}: fwd {e1:.2e} dgrad {e2:.2e}')
assert worst < 2e-5, worst
print('check_dma ok', worst)
