"""The LDS-DMA one-tap convolution kernel (csrc/experimental/conv_igemm_x3dma.hip) is NOT in the product library since
round 3 (`make -C ever_amd/csrc EXPERIMENTAL=1` compiles it in, EVK_X3_DMA=1 then selects it; it is measured at parity
with the register-staged kernels).  This test runs only against such a build (EVK_WITH_X3DMA=1 in the environment).  Its switch is read once per process, so its parity check
(tools/check_dma.py: forward + data gradient vs torch fp64, strides 1/2, ragged M / Cout, bias / ReLU) runs in a child
process with the switch set, through the same C-ABI entry points as everything else."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('EVK_WITH_X3DMA') != '1', reason='experimental kernel not in the default build')]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('cfg', ['0', '3', '5'])
def test_dma_one_tap_convolution_matches_torch(cuda, cfg):
    env = dict(os.environ, EVK_X3_DMA='1', EVK_X3_DMA_CFG=cfg)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_dma.py')], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and 'check_dma ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
