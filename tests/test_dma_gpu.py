"""The LDS-DMA one-tap convolution kernel (csrc/conv_igemm_x3dma.hip) is off by default (EVK_X3_DMA=1 turns it on; it
is measured at parity with the register-staged kernels).  Its switch is read once per process, so its parity check
(tools/check_dma.py: forward + data gradient vs torch fp64, strides 1/2, ragged M / Cout, bias / ReLU) runs in a child
process with the switch set, through the same C-ABI entry points as everything else."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('cfg', ['0', '3', '5'])
def test_dma_one_tap_convolution_matches_torch(cuda, cfg):
    env = dict(os.environ, EVK_X3_DMA='1', EVK_X3_DMA_CFG=cfg)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_dma.py')], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and 'check_dma ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
