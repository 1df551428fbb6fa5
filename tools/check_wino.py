"""dev tool / test helper: 3x3 stride-1 'same' convolutions through the C-ABI against torch CPU fp64 — forward and data
gradient, ragged patches, a partial last channel chunk, bias / ReLU / statistics / accumulate / scale-slot epilogues, fp32 and
packed operands — under the current switches.  tests/test_wino_gpu.py runs it in child processes with the Winograd F(2,3)
kernel (csrc/conv3x3_wino_x3.hip) forced onto every shape it can take (EVK_WINO=2 EVK_X3_HALO_MIN_WG=0; the switches are
read once per process) and with it off (EVK_WINO=0: the direct halo kernel, same checks)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd import _C
from ever_amd.hip import functional as HF
dev = torch.device('cuda:0')
torch.manual_seed(0)
worst = 0.0
# (n, h, w, cin, cout, bias, relu): whole patches, patches hanging over the bottom / right edge, odd widths, 4.5 chunks of
# channels, Cout that is not a multiple of the 128-wide tile
for (n, h, w, cin, cout, bias, relu) in [(2, 32, 32, 64, 128, False, False), (1, 48, 40, 72, 136, True, False),
                                         (2, 30, 23, 32, 64, True, True), (1, 64, 64, 256, 256, False, False),
                                         (3, 16, 16, 128, 192, False, False), (1, 77, 43, 40, 200, True, False),
                                         (2, 128, 128, 64, 128, False, False)]:
    x = torch.randn(n, cin, h, w) + 0.3
    wt = torch.randn(cout, cin, 3, 3) * 0.05
    b = torch.randn(cout) if bias else None
    xr = x.double().requires_grad_()
    yr = torch.nn.functional.conv2d(xr, wt.double(), None if b is None else b.double(), padding=1)
    if relu:
        yr = torch.relu(yr)
    g = torch.randn_like(yr)
    yr.backward(g)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    wg = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = HF.conv2d(xg, wg, None if b is None else b.to(dev), padding=1, relu=relu)
    y.backward(g.float().to(dev).contiguous(memory_format=torch.channels_last))
    yc, yd = y.detach().cpu().double(), yr.detach()
    if relu:   # (a mask bit that differs from fp64 within rounding of zero is not an error)
        keep = yd.abs() > 1e-4
        yc, yd = yc * keep, yd * keep
    e1 = float((yc - yd).abs().max() / yd.abs().max())
    e2 = float((xg.grad.cpu().double() - xr.grad).abs().max() / xr.grad.abs().max()) if not relu else 0.0
    worst = max(worst, e1, e2)
    print(f'n{n} {h}x{w} {cin}->{cout} bias{int(bias)} relu{int(relu)}: fwd {e1:.2e} dgrad {e2:.2e}')

# raw C-ABI: packed operand against fp32 under the same scale; statistics epilogue against the output it wrote; accumulate
# epilogue of the data gradient; the output's scale slots
lib = _C.load(); st = torch.cuda.current_stream().cuda_stream
aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=dev)
nw = int(lib.evk_absmax_words())
def scale(t):
    b = torch.zeros(nw, dtype=torch.int32, device=dev)
    _C.call('evk_absmax', t.data_ptr(), t.numel(), b.data_ptr(), aws.data_ptr(), st)
    return b
for (n, h, w, cin, cout) in [(2, 64, 64, 128, 128), (1, 48, 40, 72, 256), (2, 32, 48, 64, 384)]:
    d = _C.ConvDesc(n, h, w, cin, h, w, cout, 3, 3, 1, 1, 1, 1, 1, 1)
    x = (torch.randn(n, h, w, cin) + 0.25).to(dev); wt = (torch.randn(cout, 3, 3, cin) * 0.05).to(dev)
    dy = (torch.randn(n, h, w, cout) * 1e-3).to(dev); acc = torch.randn(n, h, w, cin).to(dev)
    bx, bw, bdy = scale(x), scale(wt), scale(dy)
    xp = torch.empty_like(x); _C.call('evk_pack_f16x2', x.data_ptr(), x.numel(), bx.data_ptr(), xp.data_ptr(), st)
    dyp = torch.empty_like(dy); _C.call('evk_pack_f16x2', dy.data_ptr(), dy.numel(), bdy.data_ptr(), dyp.data_ptr(), st)
    pf = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=dev)
    pd = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 1), dtype=torch.uint8, device=dev)
    _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 0, pf.data_ptr(), bw.data_ptr(), st)
    _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 1, pd.data_ptr(), bw.data_ptr(), st)
    cap = int(lib.evk_conv2d_stats_max_parts(ctypes.byref(d)))
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.permute(0, 3, 1, 2).double().cpu(), padding=1).permute(0, 2, 3, 1)
    outs = []
    for src, flags in ((x, 0), (xp, 2)):
        for stats in (0, 1):
            y = torch.empty(n, h, w, cout, device=dev)
            parts = torch.zeros(max(cap, 1) * 3 * cout, device=dev)
            npart = ctypes.c_int32(0)
            yb = torch.zeros(nw, dtype=torch.int32, device=dev)
            _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), src.data_ptr(), bx.data_ptr(), pf.data_ptr(), bw.data_ptr(), None,
                    None, y.data_ptr(), flags, parts.data_ptr() if stats else None, cap if stats else 0, ctypes.byref(npart),
                    None if stats else yb.data_ptr(), st)
            torch.cuda.synchronize()
            outs.append(y)
            if not stats:   # the output's operand-scale slots hold the bit image of max|y|
                assert int(yb.view(64, nw // 64)[:, 0].max()) == int(y.abs().max().view(torch.int32)), 'scale slots'
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
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[2], outs[3]), 'the statistics epilogue changed the output'
    # (packed = the operand rounded to its 22 bits BEFORE the input transform, fp32 = after it: equal to rounding, not bit for bit)
    ep = float((outs[0] - outs[2]).abs().max() / ref.abs().max())
    e = float((outs[0].cpu().double() - ref).abs().max() / ref.abs().max())
    g = []
    for src, flags in ((dy, 0), (dyp, 4)):
        dx = torch.empty_like(x)
        _C.call('evk_conv2d_dgrad_f16x2_ex', ctypes.byref(d), src.data_ptr(), bdy.data_ptr(), pd.data_ptr(), bw.data_ptr(),
                acc.data_ptr(), dx.data_ptr(), None, flags, st)
        g.append(dx)
    torch.cuda.synchronize()
    gref = torch.nn.grad.conv2d_input((n, cin, h, w), wt.permute(0, 3, 1, 2).double().cpu(), dy.permute(0, 3, 1, 2).double().cpu(),
                                      padding=1).permute(0, 2, 3, 1) + acc.double().cpu()
    e2 = float((g[0].cpu().double() - gref).abs().max() / gref.abs().max())
    e3 = float((g[0] - g[1]).abs().max() / gref.abs().max())
    worst = max(worst, e, e2, ep, e3)
    print(f'raw {n}x{h}x{w} {cin}->{cout}: fwd {e:.2e} dgrad+accum {e2:.2e}; packed vs fp32 operand {ep:.1e} / {e3:.1e}')
assert worst < 2e-5, worst
print('check_wino ok', worst)
