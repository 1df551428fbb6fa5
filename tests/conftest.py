import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _host_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    import torch
    torch.set_num_threads(min(_host_cores(), 16))  # the oracle runs on the CPU; do not oversubscribe a cgroup quota


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


@pytest.fixture(params=['f16x2', 'bf16x3', 'f32'])
def conv_math(request):
    """Run the test under the three fp32-grade convolution arithmetics (fp16 2-term split = default, bf16 3-term split,
    exact fp32 MFMA = yardstick)."""
    from ever_amd.hip import functional as HF
    prev = HF.set_conv_math(request.param)
    yield request.param
    HF.set_conv_math(prev)
