"""`bench.py --gpus N` starts its own N ranks when no launcher did (VERDICT r3 missing item 2).

Reference contract: an `env://` rendezvous prepared by torchrun, one process per GPU
(`/root/reference/ever/trainer/th_ddp_trainer.py:13-30`).  No GPU here: `--dry-launch` takes the same
path up to the process group (gloo) and reports what every rank saw."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_gpus_2_dry_launch_starts_two_ranks():
    out = _run(['--gpus', '2', '--dry-launch'])
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [ln for ln in out.stdout.splitlines() if ln.startswith('{"dry_launch"')]
    assert len(rows) == 1, out.stdout          # rank 0 alone prints
    d = json.loads(rows[0])
    assert d['n_gpus'] == 2 and d['gpus_arg'] == 2
    assert sorted(r['rank'] for r in d['ranks']) == [0, 1]
    assert sorted(r['local_rank'] for r in d['ranks']) == [0, 1]
    assert all(r['world_size_env'] == 2 for r in d['ranks'])
    assert len({r['pid'] for r in d['ranks']}) == 2   # two processes, not two threads


def test_gpus_2_without_two_gpus_fails_loudly():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('this node has two GPUs')
    out = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'])
    assert out.returncode != 0
    assert 'GPU' in out.stderr and '--gpus 2' in out.stderr, out.stderr[-500:]
    assert '"metric"' not in out.stdout   # no n_gpus line from fewer devices than asked for


def test_flag_and_launcher_must_agree():
    out = _run(['--gpus', '4', '--dry-launch'], env_extra={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert out.returncode != 0 and 'WORLD_SIZE=1' in out.stderr
