"""The HIP gradient-exchange path at a REAL world size of 2 on one MI355X.

Two processes share cuda:0 over a gloo group with device tensors (RCCL refuses duplicate devices); see
tests/world2_gpu_worker.py for what each case holds.  Reference: ever/trainer/th_ddp_trainer.py:13-30,
ever/module/loss.py:20-23,46-48, ever/core/launcher.py:196,317-321."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', 'world2_gpu_worker.py')
CASES = ['flat_equals_torch_ddp', 'forward_times_2', 'sync_bn_two_ranks', 'dice_two_ranks', 'trainer_three_steps']


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.fixture(scope='module')
def world2_results(cuda, tmp_path_factory):
    out = tmp_path_factory.mktemp('world2')
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4',
                   EVK_WGRAD_SHARED='0')   # (FlatGradDDP == torch DDP bit for bit: side-stream launches split as if alone)
        procs.append(subprocess.Popen([sys.executable, WORKER, str(out)] + CASES, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=900)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append('TIMEOUT (two processes sharing the device hung)\n' + p.communicate()[0])
    res = []
    for r in range(2):
        path = out / f'rank{r}.json'
        res.append(json.loads(path.read_text()) if path.exists() else {})
    return res, logs, [p.returncode for p in procs]


@pytest.mark.parametrize('case', CASES)
def test_world2_on_one_gpu(world2_results, case):
    res, logs, rcs = world2_results
    for r in range(2):
        got = res[r].get(case)
        assert got is not None, f'rank {r} never reached {case} (rc {rcs[r]}):\n{logs[r][-3000:]}'
        assert got.startswith('ok'), f'rank {r}: {got}'
