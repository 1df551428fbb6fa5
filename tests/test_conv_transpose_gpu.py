"""nn.ConvTranspose2d on the HIP path (BASELINE.json north_star "transposed-conv lowered to MFMA").
The reference's hot path has no transposed convolution, so the oracle is torch.nn.ConvTranspose2d on the CPU (fp64):
forward, input / weight / bias gradients at 1e-4, both arithmetics, through the C-ABI (evk_conv_transpose2d_*)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    # cin, cout, k, stride, pad, out_pad, dil, bias, n, h, w
    (32, 16, 2, 2, 0, 0, 1, True, 2, 9, 7),       # the 2x up-sampling deconvolution of decoder heads
    (16, 32, 3, 2, 1, 1, 1, True, 2, 8, 8),       # k3 s2 p1 op1: exact doubling
    (64, 64, 4, 2, 1, 0, 1, False, 3, 6, 10),     # k4 s2 p1
    (8, 24, 3, 1, 1, 0, 1, True, 1, 11, 5),       # stride 1 = correlation with the flipped kernel
    (24, 8, 3, 3, 0, 2, 1, False, 2, 5, 4),       # stride 3 with output_padding 2
    (16, 16, 3, 2, 2, 1, 2, True, 2, 7, 9),       # dilation 2
    (128, 64, 2, 2, 0, 0, 1, False, 2, 16, 16),   # wide enough for the 128-row tiles
]


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize('case', CASES)
def test_conv_transpose2d_matches_torch(cuda, conv_math, case):
    import ever_amd as er
    cin, cout, k, s, p, op, dil, bias, n, h, w = case
    torch.manual_seed(hash(case) % 1000)
    ref = torch.nn.ConvTranspose2d(cin, cout, k, s, p, op, bias=bias, dilation=dil).double()
    m = er.module.ConvTranspose2d(cin, cout, k, s, p, op, bias=bias, dilation=dil)
    m.load_state_dict({kk: v.float() for kk, v in ref.state_dict().items()})
    assert m.weight.stride() == (k * k * cout, 1, k * cout, cout) or k == 1
    m = m.to(cuda)
    x = torch.randn(n, cin, h, w)
    xr = x.double().requires_grad_()
    yr = ref(xr)
    g = torch.randn_like(yr)
    yr.backward(g)
    xg = x.to(cuda).requires_grad_()
    y = m(xg)
    assert tuple(y.shape) == tuple(yr.shape)
    y.backward(g.float().to(cuda))
    torch.cuda.synchronize()
    assert _rel(y.detach().cpu(), yr.detach()) < 1e-4
    assert _rel(xg.grad.cpu(), xr.grad) < 1e-4
    assert _rel(m.weight.grad.cpu(), ref.weight.grad) < 1e-4
    assert m.weight.grad.stride() == m.weight.stride() or k == 1
    if bias:
        assert _rel(m.bias.grad.cpu(), ref.bias.grad) < 1e-4


def test_conv_transpose2d_is_the_adjoint_of_conv2d(cuda):
    """<conv(u), z> = <u, conv_transpose(z)> with the SAME weight memory: the size-independent property the operator is
    defined by, at a size the CPU oracle would not finish quickly (16 x 64 x 128 x 128 -> 256 x 256)."""
    import ever_amd as er
    from ever_amd.hip import functional as HF
    torch.manual_seed(3)
    conv = er.module.Conv2d(64, 32, 3, 2, 1, bias=False).to(cuda)       # C: u[16,64,256,256] -> z[16,32,128,128]
    u = torch.randn(16, 64, 256, 256, device=cuda)
    z = torch.randn(16, 32, 128, 128, device=cuda)
    with torch.no_grad():
        cu = conv(u)
        # the ConvTranspose2d parameter [Cin_t=32, Cout_t=64, 3, 3] sharing the convolution's OHWI memory
        ctz = HF.conv_transpose2d(z, conv.weight, None, 2, 1, 1, 1)
    assert tuple(ctz.shape) == tuple(u.shape)
    lhs = float((cu.double() * z.double()).sum())
    rhs = float((u.double() * ctz.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs)), (lhs, rhs)
