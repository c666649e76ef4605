"""GPU: the N > 1 code path of bench.py on a 1-GPU box -- two ranks on the same device over gloo
(IAMX_BENCH_DEBUG_ONE_GPU=1; RCCL refuses two ranks per device, the collectives are the only thing
that differs).  Each rank packs its own images, the stores are all-gathered, the pairs are dealt
out, and rank 0's self-check compares pairs INCLUDING one whose other image was packed by rank 1
with oracle/cpu_ref.c -- a buffer missing from the all-gather (the train norms once were) turns
up here instead of on the 8-GPU node."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_verifies_pairs_across_ranks():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, IAMX_BENCH_DEBUG_ONE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(REPO, 'bench.py'),
           '--gpus', '2', '--steps', '1', '--warmup', '0', '--images', '64', '--no-sift',
           '--no-cpu-baseline', '--ba-iters', '1']
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['unresolved'] == 0
    assert d['verified_pairs'] == 64 and d['verify']['survivors_checked'] > 1000
    assert d['config']['pairs_per_step'] == 64 * 63 // 2
    # 30 % of every image is a noisy copy of its predecessor: ~860 survivors per adjacent pair and direction
    assert d['survivors_per_step'] > 63 * 2 * 700 and d['candidates_per_step'] >= d['survivors_per_step']
    assert d['ba'] is not None and d['ba']['parallelism'] == 'point-shard x2'
    # the product's per-round exchange inside the timed step: rank 0 saw every rank's survivor counts
    assert d['gather']['survivor_count_seen_by_rank0'] == d['survivors_per_step']
    assert len(d['per_rank_seconds']) == 2 and len(d['per_rank_sweep_seconds']) == 2
