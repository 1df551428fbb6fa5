"""The LDS-DMA one-tap convolution kernel (csrc/conv1x1_dma.hip; reference call sites: the 1x1 convolutions of
ever/module/_resnets.py:72-112, fpn.py:23-37, fs_relation.py:23-53).  The dispatcher gives it the wide-output shapes only,
and its switches are read once per process, so the parity check (tools/check_dma.py: forward + data gradient vs torch fp64,
strides 1/2, ragged M / Cout, bias / ReLU / accumulate / statistics epilogues, packed == fp32 operand bit for bit) runs in
child processes with the kernel forced onto every shape it can take, in each of its tile / ring forms."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('env', [dict(EVK_C1_DMA='2'),                              # two-stage ring, 128-wide tiles (production form)
                                 dict(EVK_C1_PS2='2'),                              # three-role persistent form (conv1x1_ps2.hip) wherever it applies
                                 dict(EVK_TUNE='1', EVK_X3_FORCE='s128'),           # software-pipelined form with loader waves (conv1x1_sp.hip), ring of four
                                 dict(EVK_TUNE='1', EVK_X3_FORCE='t64'),            # ... ring of three, 64-wide tiles
                                 dict(EVK_TUNE='1', EVK_X3_FORCE='e64'),            # ... 64-wide tiles
                                 dict(EVK_TUNE='1', EVK_X3_FORCE='d256'),           # three-stage ring, 256-wide tiles
                                 dict(EVK_C1_DMA='0')],                             # the register-staged kernels, same checks
                         ids=lambda e: '-'.join(e.values()))
def test_dma_one_tap_convolution_matches_torch(cuda, env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_dma.py')], env=dict(os.environ, **env),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and 'check_dma ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_dispatch_takes_the_wide_one_tap_layers(cuda):
    """in process, default switches: a 64 -> 256 layer on a 128 x 128 map goes to the DMA kernel (visible as a different
    summation order than the same layer cut into two narrower outputs would not show — so check against fp64 instead, and
    that the result is deterministic)"""
    import ctypes
    from ever_amd import _C
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    n, h, cin, cout = 2, 128, 64, 256
    d = _C.ConvDesc(n, h, h, cin, h, h, cout, 1, 1, 1, 1, 0, 0, 1, 1)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, h, h, cin, generator=g).to(cuda)
    wt = (torch.randn(cout, 1, 1, cin, generator=g) * 0.05).to(cuda)
    aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=cuda)
    bits = []
    for t in (x, wt):
        b = torch.zeros(int(lib.evk_absmax_words()), dtype=torch.int32, device=cuda)
        _C.call('evk_absmax', t.data_ptr(), t.numel(), b.data_ptr(), aws.data_ptr(), st)
        bits.append(b)
    pf = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=cuda)
    _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 0, pf.data_ptr(), bits[1].data_ptr(), st)
    ys = []
    for _ in range(2):
        y = torch.empty(n, h, h, cout, device=cuda)
        z = ctypes.c_int32(0)
        _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), x.data_ptr(), bits[0].data_ptr(), pf.data_ptr(), bits[1].data_ptr(),
                None, None, y.data_ptr(), 0, None, 0, ctypes.byref(z), None, st)
        ys.append(y)
    torch.cuda.synchronize()
    assert torch.equal(ys[0], ys[1])
    ref = torch.einsum('nhwi,oi->nhwo', x.double(), wt.view(cout, cin).double())
    assert float((ys[0].double() - ref).abs().max() / ref.abs().max()) < 2e-6
