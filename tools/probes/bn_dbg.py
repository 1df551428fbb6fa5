import os, sys, torch
sys.path.insert(0, os.getcwd())
import ever_amd as er
dev = torch.device('cuda:0')
torch.manual_seed(0)
shapes = [(16, 64, 77, 43), (16, 64, 76, 43), (2, 128, 154, 86)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for shape in shapes:
    x = (torch.randn(*shape) + 0.7)
    g = torch.randn(*shape)
    for relu in (True, False):
        bn = er.module.BatchNorm2d(shape[1]).to(dev)
        ref = torch.nn.BatchNorm2d(shape[1]).double()
        xr = x.double().requires_grad_()
        y = ref(xr)
        (torch.relu(y) if relu else y).backward(g.double())
        xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
        z = bn(xg, relu=relu)
        z.backward(g.to(dev).contiguous(memory_format=torch.channels_last))
        d = (xg.grad.cpu().double() - xr.grad).abs() / xr.grad.abs().max()
        err = float(d.max())
        eg = float((bn.weight.grad.cpu().double() - ref.weight.grad).abs().max() / ref.weight.grad.abs().max())
        eb = float((bn.bias.grad.cpu().double() - ref.bias.grad).abs().max() / ref.bias.grad.abs().max())
        print(shape, 'relu', relu, 'dx err', err, 'dgamma err', eg, 'dbeta err', eb)
        if err > 1e-4:
            bad = (d > 1e-4).permute(0, 2, 3, 1).reshape(-1, shape[1])      # [rows][C]
            rows = bad.any(1).nonzero().flatten()
            chans = bad.any(0).nonzero().flatten()
            print('   bad elements', int(bad.sum()), 'rows', rows[:8].tolist(), '...', rows[-8:].tolist(), 'n rows', len(rows), 'channels', chans.tolist()[:16], len(chans))
