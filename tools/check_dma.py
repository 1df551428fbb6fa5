"""dev tool / test helper: one-tap (1x1) convolutions through the C-ABI against torch CPU fp64 — forward and data gradient,
strides 1/2, ragged M and Cout, bias / ReLU / statistics epilogues, fp32 and packed operands — under the current switches.
tests/test_conv1x1_dma_gpu.py runs it in child processes with the LDS-DMA kernel (csrc/conv1x1_dma.hip) forced onto every
shape it can take (EVK_C1_DMA=2, or EVK_TUNE=1 EVK_X3_FORCE=<tile>: both are read once per process)."""
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
                                                 (2, 30, 30, 512, 128, 2, True, False), (1, 40, 24, 320, 384, 1, False, True),
                                                 # (round 5: 563 ragged row tiles x 2 column tiles, the second one partial — a persistent
                                                 # workgroup walks four or five tiles: ring across tile boundaries, store role, nk = 3)
                                                 (3, 160, 150, 96, 200, 1, True, False), (2, 128, 128, 64, 256, 1, False, False)]:   # (no ReLU on 14 M outputs: a mask bit that differs from fp64 within rounding of zero is not an error)
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

# raw C-ABI: packed operand bit-identical to fp32 under the same scale; statistics epilogue against the output it wrote;
# accumulate epilogue of the data gradient
lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=dev)
def scale(t):
    b = torch.zeros(int(lib.evk_absmax_words()), dtype=torch.int32, device=dev)
    _C.call('evk_absmax', t.data_ptr(), t.numel(), b.data_ptr(), aws.data_ptr(), st)
    return b
for (n, h, w, cin, cout) in [(4, 64, 64, 64, 256), (2, 96, 64, 256, 128), (1, 64, 64, 128, 512), (3, 50, 34, 96, 200),
                             (3, 160, 150, 96, 200)]:   # (the last: several tiles per persistent workgroup, ragged in M and Cout)
    d = _C.ConvDesc(n, h, w, cin, h, w, cout, 1, 1, 1, 1, 0, 0, 1, 1)
    x = (torch.randn(n, h, w, cin) + 0.25).to(dev); wt = (torch.randn(cout, 1, 1, cin) * 0.05).to(dev)
    dy = (torch.randn(n, h, w, cout) * 1e-3).to(dev); acc = torch.randn(n, h, w, cin).to(dev)
    bx, bw, bdy = scale(x), scale(wt), scale(dy)
    xp = torch.empty_like(x); _C.call('evk_pack_f16x2', x.data_ptr(), x.numel(), bx.data_ptr(), xp.data_ptr(), st)
    dyp = torch.empty_like(dy); _C.call('evk_pack_f16x2', dy.data_ptr(), dy.numel(), bdy.data_ptr(), dyp.data_ptr(), st)
    pf = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
    pd = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 1), dtype=torch.uint8, device=dev)
    _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 0, pf.data_ptr(), bw.data_ptr(), st)
    _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 1, pd.data_ptr(), bw.data_ptr(), st)
    cap = int(lib.evk_conv2d_stats_max_parts(ctypes.byref(d)))
    outs = []
    for src, flags in ((x, 0), (xp, 2)):
        for stats in (0, 1):
            y = torch.empty(n, h, w, cout, device=dev)
            parts = torch.zeros(max(cap, 1) * 3 * cout, device=dev)
            npart = ctypes.c_int32(0)
            _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), src.data_ptr(), bx.data_ptr(), pf.data_ptr(), bw.data_ptr(), None,
                    None, y.data_ptr(), flags, parts.data_ptr() if stats else None, cap if stats else 0, ctypes.byref(npart),
                    None, st)
            torch.cuda.synchronize()
            outs.append(y)
            if stats and npart.value > 0:
                rec = parts[:npart.value * 3 * cout].view(npart.value, 3, cout).double()
                cnt, mean, m2 = rec[:, 0], rec[:, 1], rec[:, 2]
                tot = cnt.sum(0)
                gm = (cnt * mean).sum(0) / tot
                var = (m2 + cnt * (mean - gm) ** 2).sum(0) / tot
                yd = y.double().view(-1, cout)
                assert float(tot[0]) == yd.shape[0], (float(tot[0]), yd.shape)
                em = float((gm - yd.mean(0)).abs().max() / yd.abs().max())
                ev = float((var - yd.var(0, unbiased=False)).abs().max() / yd.var(0, unbiased=False).max())
                assert em < 1e-6 and ev < 1e-5, (em, ev)
                print(f'   stats {cin}->{cout}: {npart.value} records, mean {em:.1e} var {ev:.1e}')
    for o in outs[1:]:
        assert torch.equal(outs[0], o), float((outs[0] - o).abs().max())
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.permute(0, 3, 1, 2).double().cpu()).permute(0, 2, 3, 1)
    e = float((outs[0].cpu().double() - ref).abs().max() / ref.abs().max())
    g = []
    for src, flags in ((dy, 0), (dyp, 4)):
        dx = torch.empty_like(x)
        _C.call('evk_conv2d_dgrad_f16x2_ex', ctypes.byref(d), src.data_ptr(), bdy.data_ptr(), pd.data_ptr(), bw.data_ptr(),
                acc.data_ptr(), dx.data_ptr(), None, flags, st)
        g.append(dx)
    torch.cuda.synchronize()
    assert torch.equal(g[0], g[1])
    gref = torch.einsum('nhwo,oi->nhwi', dy.double().cpu(), wt.view(cout, cin).double().cpu()) + acc.double().cpu()
    e2 = float((g[0].cpu().double() - gref).abs().max() / gref.abs().max())
    worst = max(worst, e, e2)
    print(f'raw {n}x{h}x{w} {cin}->{cout}: packed == fp32 operand (fwd, fwd+stats, dgrad+accum); fwd {e:.2e} dgrad {e2:.2e}')
assert worst < 2e-5, worst
print('check_dma ok', worst)
