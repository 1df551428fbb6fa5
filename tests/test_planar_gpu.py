"""Planar operands of the f16x2 arithmetic and the weight gradient that reads them (csrc/conv_wgrad_tr.hip: both operands
global -> LDS by DMA, fragments by transposing LDS reads, nine-tap halo form for 3x3 / stride 1): the planar pair is the
packed word's (h, l) pair in another memory order, so the kernel must agree with the packed-operand kernels to
accumulation order (1e-5) and with a float64 convolution weight gradient at the f16x2 tolerance."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _bits(lib, t, aws, st):
    from ever_amd import _C
    b = torch.zeros(int(lib.evk_absmax_words()), dtype=torch.int32, device=t.device)
    _C.call('evk_absmax', t.data_ptr(), t.numel(), b.data_ptr(), aws.data_ptr(), st)
    return b


def test_planar_round_trip_equals_the_packed_word(cuda):
    from ever_amd import _C
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=cuda)
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(1 << 16, generator=g) * torch.logspace(-6, 2, 1 << 16)).to(cuda)
    bits = _bits(lib, x, aws, st)
    planar, packed = torch.empty_like(x), torch.empty_like(x)
    _C.call('evk_pack_planar_f16x2', x.data_ptr(), x.numel(), bits.data_ptr(), planar.data_ptr(), st)
    _C.call('evk_pack_f16x2', x.data_ptr(), x.numel(), bits.data_ptr(), packed.data_ptr(), st)
    a, b = torch.empty_like(x), torch.empty_like(x)
    _C.call('evk_unpack_planar_f16x2', planar.data_ptr(), x.numel(), bits.data_ptr(), a.data_ptr(), st)
    _C.call('evk_unpack_f16x2', packed.data_ptr(), x.numel(), bits.data_ptr(), b.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(a, b)                       # the same two fp16 terms
    w = packed.view(torch.int32)
    h = planar.view(torch.int16)[: x.numel()].to(torch.int32) & 0xffff
    lo = planar.view(torch.int16)[x.numel():].to(torch.int32) & 0xffff
    assert torch.equal(w & 0xffff, h) and torch.equal((w >> 16) & 0xffff, lo)


CASES = [
    # n, cin, h, w, cout, k, stride, pad
    (2, 64, 32, 32, 128, 3, 1, 1),      # nine-tap, one channel tile
    (1, 128, 16, 64, 192, 3, 1, 1),     # nine-tap, two channel tiles, Cout not a multiple of the 128-row tile
    (2, 64, 24, 24, 64, 3, 1, 1),       # W % 32 != 0: the segment form with padding
    (2, 128, 16, 16, 128, 3, 2, 1),     # strided 3x3
    (3, 256, 16, 24, 128, 1, 1, 0),     # 1x1
    (2, 128, 32, 32, 256, 1, 2, 0),     # strided 1x1 shortcut
    (1, 64, 8, 40, 64, 3, 1, 1),        # rows that wrap inside a step, pixel count not a multiple of the chunk
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_planar_weight_gradient(cuda, case):
    from ever_amd import _C
    import torch.nn.functional as TF
    n, cin, h, w, cout, k, s, pad = case
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=cuda)
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    d = _C.ConvDesc(n, h, w, cin, ho, wo, cout, k, k, s, s, pad, pad, 1, 1)
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, h, w, cin, generator=g).to(cuda)
    dy = torch.randn(n, ho, wo, cout, generator=g).to(cuda)
    bx, bd = _bits(lib, x, aws, st), _bits(lib, dy, aws, st)
    xq, dq, xp, dp = (torch.empty_like(t) for t in (x, dy, x, dy))
    _C.call('evk_pack_planar_f16x2', x.data_ptr(), x.numel(), bx.data_ptr(), xq.data_ptr(), st)
    _C.call('evk_pack_planar_f16x2', dy.data_ptr(), dy.numel(), bd.data_ptr(), dq.data_ptr(), st)
    _C.call('evk_pack_f16x2', x.data_ptr(), x.numel(), bx.data_ptr(), xp.data_ptr(), st)
    _C.call('evk_pack_f16x2', dy.data_ptr(), dy.numel(), bd.data_ptr(), dp.data_ptr(), st)
    wsb = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)
    dw_planar = torch.full((cout, k, k, cin), float('nan'), device=cuda)
    dw_packed = torch.empty_like(dw_planar)
    _C.call('evk_conv2d_wgrad_f16x2_ex', ctypes.byref(d), xq.data_ptr(), bx.data_ptr(), dq.data_ptr(), bd.data_ptr(),
            dw_planar.data_ptr(), None, ws.data_ptr(), wsb, 8 | 16, st)
    _C.call('evk_conv2d_wgrad_f16x2_ex', ctypes.byref(d), xp.data_ptr(), bx.data_ptr(), dp.data_ptr(), bd.data_ptr(),
            dw_packed.data_ptr(), None, ws.data_ptr(), wsb, 2 | 4, st)
    torch.cuda.synchronize()
    xd = x.double().cpu().permute(0, 3, 1, 2)
    wd = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    (TF.conv2d(xd, wd, None, s, pad) * dy.double().cpu().permute(0, 3, 1, 2)).sum().backward()
    ref = wd.grad.permute(0, 2, 3, 1)              # OHWI
    scale = ref.abs().max().item()
    a = dw_planar.double().cpu()
    assert torch.isfinite(a).all()
    assert (a - ref).abs().max().item() <= 1e-5 * scale, (a - ref).abs().max().item() / scale
    assert (a - dw_packed.double().cpu()).abs().max().item() <= 1e-5 * scale
    # EVK_CONV_WGRAD_SHARED (32): the launch shares the chip with another stream — the wide-tile kernels split for half of the
    # CUs.  Same workspace, same result to the accumulation order, both operand forms
    for flags, src in ((8 | 16, (xq, dq)), (2 | 4, (xp, dp))):
        dw_sh = torch.full_like(dw_planar, float('nan'))
        _C.call('evk_conv2d_wgrad_f16x2_ex', ctypes.byref(d), src[0].data_ptr(), bx.data_ptr(), src[1].data_ptr(), bd.data_ptr(),
                dw_sh.data_ptr(), None, ws.data_ptr(), wsb, flags | 32, st)
        torch.cuda.synchronize()
        b = dw_sh.double().cpu()
        assert torch.isfinite(b).all()
        assert (b - ref).abs().max().item() <= 1e-5 * scale, (flags, (b - ref).abs().max().item() / scale)


def test_planar_operands_come_in_pairs(cuda):
    from ever_amd import _C
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    d = _C.ConvDesc(1, 16, 16, 64, 16, 16, 64, 3, 3, 1, 1, 1, 1, 1, 1)
    x = torch.zeros(1, 16, 16, 64, device=cuda)
    b = torch.zeros(int(lib.evk_absmax_words()), dtype=torch.int32, device=cuda)
    wsb = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)
    dw = torch.empty(64, 3, 3, 64, device=cuda)
    with pytest.raises(Exception):
        _C.call('evk_conv2d_wgrad_f16x2_ex', ctypes.byref(d), x.data_ptr(), b.data_ptr(), x.data_ptr(), b.data_ptr(),
                dw.data_ptr(), None, ws.data_ptr(), wsb, 8, st)
    d2 = _C.ConvDesc(1, 16, 16, 48, 16, 16, 64, 3, 3, 1, 1, 1, 1, 1, 1)     # Cin % 64 != 0
    x2 = torch.zeros(1, 16, 16, 48, device=cuda)
    dw2 = torch.empty(64, 3, 3, 48, device=cuda)
    wsb2 = lib.evk_conv2d_wgrad_x3_workspace_bytes(ctypes.byref(d2))
    ws2 = torch.empty(wsb2, dtype=torch.uint8, device=cuda)
    with pytest.raises(Exception):
        _C.call('evk_conv2d_wgrad_f16x2_ex', ctypes.byref(d2), x2.data_ptr(), b.data_ptr(), x.data_ptr(), b.data_ptr(),
                dw2.data_ptr(), None, ws2.data_ptr(), wsb2, 8 | 16, st)
