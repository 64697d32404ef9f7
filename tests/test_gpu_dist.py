"""The N>1 launch contract rehearsed on ONE GPU: two ranks (gloo rendezvous, both on cuda:0) run bench.py exactly as
the driver launches it for N GPUs -- sharded gallery embed + one all_gather, images sharded, max-over-ranks timing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks(cuda):
    env = dict(os.environ, CVPCE_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1',
           '--images-per-gpu', '1', '--gallery', '96', '--image-size', '640', '--no-cpu-baseline', '--no-roofline']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, 'rank 0 prints exactly one JSON line'
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['unit'] == 'images/s' and d['value'] > 0
    assert d['config']['gallery'] == 96 and d['config']['images_per_gpu'] == 1
