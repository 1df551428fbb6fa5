"""The Winograd F(2,3) 3x3 kernel (csrc/conv3x3_wino_x3.hip; reference call sites: the 3x3 / stride-1 convolutions of
ever/module/_resnets.py:21-29, fpn.py:72-73,165).  The dispatcher gives it the matrix-bound layers only (whole 128-wide
column tiles, at least one 16 x 16 patch per CU) and its switch is read once per process, so the parity check
(tools/check_wino.py: forward + data gradient vs torch fp64 on whole and ragged patches, 4.5 channel chunks, bias / ReLU /
accumulate / statistics / scale-slot epilogues, packed vs fp32 operand) runs in child processes with the kernel forced onto
every shape it can take, and once with it off (the direct halo kernel under the same checks)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('mode', ['2', '0'])
def test_winograd_3x3_matches_torch(cuda, mode):
    env = dict(os.environ, EVK_WINO=mode, EVK_X3_HALO_MIN_WG='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_wino.py')], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and 'check_wino ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_default_dispatch_takes_the_wide_128_maps_and_is_deterministic(cuda):
    """in process, default switches: 3x3x256 on a 128 x 128 map at batch 16 is a Winograd launch (1.5x fewer MFMA passes), the
    same layer on a 32 x 32 map is not; both are within fp32 rounding of an fp64 convolution and repeat bit for bit"""
    import ctypes
    from ever_amd import _C
    lib = _C.load()
    st = torch.cuda.current_stream().cuda_stream
    aws = torch.zeros(lib.evk_absmax_workspace_bytes(), dtype=torch.uint8, device=cuda)
    nw = int(lib.evk_absmax_words())
    for n, h, c in ((16, 128, 256), (16, 32, 256)):
        d = _C.ConvDesc(n, h, h, c, h, h, c, 3, 3, 1, 1, 1, 1, 1, 1)
        g = torch.Generator().manual_seed(5)
        x = (torch.randn(n, h, h, c, generator=g) + 0.2).to(cuda)
        wt = (torch.randn(c, 3, 3, c, generator=g) * 0.03).to(cuda)
        bits = []
        for t in (x, wt):
            b = torch.zeros(nw, dtype=torch.int32, device=cuda)
            _C.call('evk_absmax', t.data_ptr(), t.numel(), b.data_ptr(), aws.data_ptr(), st)
            bits.append(b)
        pf = torch.empty(lib.evk_conv2d_split_weight_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device=cuda)
        _C.call('evk_conv2d_split_weight_f16x2', ctypes.byref(d), wt.data_ptr(), 0, pf.data_ptr(), bits[1].data_ptr(), st)
        ys = []
        for _ in range(2):
            y = torch.empty(n, h, h, c, device=cuda)
            z = ctypes.c_int32(0)
            _C.call('evk_conv2d_fwd_f16x2', ctypes.byref(d), x.data_ptr(), bits[0].data_ptr(), pf.data_ptr(), bits[1].data_ptr(),
                    None, None, y.data_ptr(), 0, None, 0, ctypes.byref(z), None, st)
            ys.append(y)
        torch.cuda.synchronize()
        assert torch.equal(ys[0], ys[1])
        ref = torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).double().cpu(), wt.permute(0, 3, 1, 2).double().cpu(), padding=1)
        err = float((ys[0][:1].permute(0, 3, 1, 2).cpu().double() - ref).abs().max() / ref.abs().max())
        assert err < 3e-6, (h, err)
