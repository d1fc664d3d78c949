"""bench.py's own multi-rank path on ONE GPU (two ranks sharing the device, gloo moving the CUDA gradient buckets):
`--strong` semantics (a GLOBAL batch split over the ranks, here 4 crops -> B_local = 2), the JSON contract
fields the driver reads, and replica consistency after the run (rank 0's parameters are broadcast at the first step; the
all-reduced gradients and the deterministic fused Adam must keep every rank bit-identical)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _run(extra, nproc=2):
    env = dict(os.environ, PNNP_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(REPO, 'bench.py'), '--gpus', str(nproc), '--steps', '3', '--warmup', '1',
           '--size', '64', '--no-cpu-baseline'] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_strong_scaling_two_ranks_b_local_2():
    out = _run(['--strong', '--batch', '4'])
    assert out['scaling'] == 'strong' and out['n_gpus'] == 2
    assert out['config']['crops_per_gpu'] == 2 and out['config']['global_batch'] == 4
    assert out['steps'] == 3 and out['warmup'] == 1 and out['value'] > 0
    assert abs(out['value'] - 4 * 3 / (out['ms_per_step'] * 3e-3)) < 1e-6 * out['value']       # whole-job crops / max-over-ranks time
    assert out['replica_checksum_spread'] == 0.0
    assert out['final_loss'] == out['final_loss'] and out['final_loss'] < 1.0                  # finite


@pytest.mark.parametrize('overlap', [False, True])
def test_weak_scaling_two_ranks(overlap):
    """Both reducer modes of bench.py: one collective after the backward pass (default) and buckets launched from inside it."""
    out = _run(['--batch', '2'] + (['--overlap-allreduce'] if overlap else []))
    assert out['scaling'] == 'weak' and out['config']['crops_per_gpu'] == 2 and out['config']['global_batch'] == 4
    assert out['replica_checksum_spread'] == 0.0
    # the event pair around reducer.finish() is a WAIT only with overlap; without it it brackets the whole collective
    key, other = ('allreduce_wait_ms_per_step', 'allreduce_ms_per_step') if overlap else ('allreduce_ms_per_step', 'allreduce_wait_ms_per_step')
    assert out[key]['overlap'] == overlap and other not in out


def test_strong_scaling_uneven_global_batch_and_per_rank_record():
    """A global batch of 5 over 2 ranks (3 + 2 crops): bench.py no longer refuses it; the line carries the per-rank step times
    and the all-reduce wait (event pair around reducer.finish()) the scaling record needs."""
    out = _run(['--strong', '--batch', '5'])
    assert out['config']['global_batch'] == 5 and out['config']['crops_per_gpu'] == 3      # rank 0's shard
    assert abs(out['value'] - 5 * 3 / (out['ms_per_step'] * 3e-3)) < 1e-6 * out['value']
    assert out['replica_checksum_spread'] == 0.0
    pr = out['per_rank_ms_per_step']
    assert len(pr['all']) == 2 and 0 < pr['min'] <= pr['max'] <= out['ms_per_step'] * 1.05
    w = out['allreduce_ms_per_step']                                                          # default mode: one exposed collective
    assert w['mean_min_over_ranks'] >= 0.0 and w['worst_step_any_rank'] >= w['mean_max_over_ranks'] >= w['mean_min_over_ranks']
    assert w['grad_bytes'] > 0


@pytest.mark.parametrize('global_batch', [16, 128])
def test_world_8_strong_overlap_rehearsal(global_batch):
    """the launch line the driver uses on an 8-GPU node -- `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` -- with
    `--strong --overlap-allreduce`, rehearsed as EIGHT ranks sharing this box's one GPU (gloo moves the CUDA buckets; the number is meaningless, the
    path is the driver's): global batches of 16 (2 crops per rank: SURVEY 8(d) C4's strong-scaling point) and 128 (16 per rank: the weak-scaling
    shard), the persistent split scoped to the overlapped backward pass, eight replicas bit-identical at the end, one JSON line from rank 0."""
    out = _run(['--strong', '--batch', str(global_batch), '--overlap-allreduce'], nproc=8)
    assert out['scaling'] == 'strong' and out['n_gpus'] == 8
    assert out['config']['global_batch'] == global_batch and out['config']['crops_per_gpu'] == global_batch // 8
    assert out['config']['parallelism'] == 'dp8'
    assert abs(out['value'] - global_batch * 3 / (out['ms_per_step'] * 3e-3)) < 1e-6 * out['value']
    assert out['replica_checksum_spread'] == 0.0
    assert len(out['per_rank_ms_per_step']['all']) == 8
    w = out['allreduce_wait_ms_per_step']
    assert w['overlap'] is True and w['grad_bytes'] > 0
    assert out['final_loss'] == out['final_loss'] and out['final_loss'] < 1.0
