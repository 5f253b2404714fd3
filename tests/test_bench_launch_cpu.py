"""`python bench.py --gpus N` must work from a plain python invocation (the driver runs it that way at N = 1 and under
torch.distributed.run at N > 1): the self-launcher starts one process per rank before anything touches a GPU. Exercised here
with the CPU dry run (gloo, toy network through the real BucketedGradAllReduce)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _run(cmd):
    env = dict(os.environ, OMP_NUM_THREADS='2')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout        # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_plain_python_invocation_self_launches_two_ranks():
    out = _run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '4', '--warmup', '1', '--dry-run-cpu'])
    assert out['n_gpus'] == 2 and out['steps'] == 4 and out['warmup'] == 1 and out['dry_run'] is True
    assert out['config']['parallelism'] == 'dp2' and out['scaling'] == 'weak'


@pytest.mark.timeout(300)
def test_driver_style_torchrun_invocation():
    out = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                '--master-port', '29533', 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run-cpu'])
    assert out['n_gpus'] == 2 and out['steps'] == 3


def test_self_launch_happens_before_any_gpu_call():
    """static check: nothing between the argument parser and the self-launch branch may touch torch.cuda"""
    src = (ROOT / 'bench.py').read_text()
    main = src[src.index('def main():'):]
    head = main[:main.index('raise SystemExit(self_launch(args.gpus))')]
    assert 'torch.cuda' not in head and 'import mmmm_amd' not in head and 'from mmmm_amd' not in head
