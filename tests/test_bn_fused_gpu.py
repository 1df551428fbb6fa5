"""The one-launch BatchNorm backward for small maps (csrc/bn.hip: bn_bwd_fused_kernel — registers across two grid-wide
barriers) against the three-launch form of the same entry point (EVK_BN_NO_FUSE) and against float64: every ReLU-mask
mode, residual gradient, packed dx, ragged row counts, back-to-back launches with different grids (the barrier words are
never reset: each launch is handed the values they will have when its arrivals are complete)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(4096, 512), (1000, 64), (16384, 256), (777, 2048), (4100, 128), (96, 32), (65536, 128), (3, 8), (4096, 2048)]


def _bwd(lib, dy, x, y, gamma, beta, mean, invstd, flags, want_res, st):
    from ever_amd import _C
    rows, c = x.shape
    dev = x.device
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_res else None
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    wsb = lib.evk_bn_workspace_bytes(rows, c)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    bits = torch.zeros(int(lib.evk_absmax_words()), dtype=torch.int32, device=dev)
    _C.call('evk_bn_bwd', dy.data_ptr(), x.data_ptr(), None if y is None else y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
            mean.data_ptr(), invstd.data_ptr(), dx.data_ptr(), None if dres is None else dres.data_ptr(), dg.data_ptr(),
            db.data_ptr(), rows, c, flags, 1, ws.data_ptr(), wsb, bits.data_ptr(), st)
    return dx, dres, dg, db, bits


@pytest.mark.parametrize('mode', ['none', 'relu_from_x', 'relu_from_y_residual'])
def test_fused_backward_equals_three_launches(cuda, mode):
    from ever_amd import _C
    from ever_amd.hip import functional as F
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(5)
    NO_FUSE, PACK = 8, 2
    for rows, c in SHAPES:          # back to back: grids of different sizes on the same barrier words
        x = (torch.randn(rows, c, generator=g) * 2 + 0.5).to(cuda)
        dy = torch.randn(rows, c, generator=g).to(cuda)
        gamma = (torch.rand(c, generator=g) + 0.5).to(cuda)
        beta = (torch.randn(c, generator=g) * 0.2).to(cuda)
        mean = x.mean(0)
        invstd = 1.0 / torch.sqrt(x.var(0, unbiased=False) + 1e-5)
        res = torch.randn(rows, c, generator=g).to(cuda) if mode == 'relu_from_y_residual' else None
        yv = None
        relu = 0 if mode == 'none' else 1
        if res is not None:
            yv = torch.relu((x - mean) * invstd * gamma + beta + res)
        for pk in (0, PACK):
            a = _bwd(lib, dy, x, yv, gamma, beta, mean, invstd, relu | pk, res is not None, st)
            b = _bwd(lib, dy, x, yv, gamma, beta, mean, invstd, relu | pk | NO_FUSE, res is not None, st)
            torch.cuda.synchronize()
            dxa, dxb = a[0], b[0]
            if pk:      # packed words -> values
                ua, ub = torch.empty_like(x), torch.empty_like(x)
                _C.call('evk_unpack_f16x2', dxa.data_ptr(), x.numel(), a[4].data_ptr(), ua.data_ptr(), st)
                _C.call('evk_unpack_f16x2', dxb.data_ptr(), x.numel(), b[4].data_ptr(), ub.data_ptr(), st)
                torch.cuda.synchronize()
                dxa, dxb = ua, ub
                # both scales are upper bounds of max|dx|
                assert F.absmax_value(a[4]) >= int(dxa.abs().max().view(torch.int32)) - 4
            # float64 reference
            xd, gd = x.double(), dy.double()
            xh = (xd - mean.double()) * invstd.double()
            if relu:
                pre = xh * gamma.double() + beta.double() + (res.double() if res is not None else 0)
                gd = gd * (pre > 0)
            m1, m2 = gd.mean(0), (gd * xh).mean(0)
            ref = gamma.double() * invstd.double() * (gd - m1 - xh * m2)
            scale = ref.abs().max().item() + 1e-30
            tol = 3e-6 if pk else 2e-6
            # (ReLU-mask ties at exactly 0 differ between fp32 and fp64 pre-activations: compare where |pre| is clear)
            ok = torch.ones_like(ref, dtype=torch.bool) if not relu else (pre.abs() > 1e-5)
            for name, t in (('fused', dxa), ('three-launch', dxb)):
                err = ((t.double() - ref).abs() * ok).max().item() / scale
                assert err < 20 * tol, (name, rows, c, pk, err)
            assert ((dxa - dxb).abs().max().item()) / scale < 20 * tol, (rows, c, pk)
            for k in (2, 3):    # dgamma, dbeta
                d = (a[k] - b[k]).abs().max().item() / (b[k].abs().max().item() + 1e-30)
                assert d < 1e-5, (rows, c, k, d)
            if res is not None:
                assert torch.equal(a[1], b[1])
