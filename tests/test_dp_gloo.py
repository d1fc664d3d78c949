"""Data-parallel plumbing on CPU with gloo (world size 2): the bucketed gradient all-reduce
that HipTrainStep drives from the backward pass, and the crop sharding rule."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pnnp_amd.trainer import BucketedAllReduce, get_cos_lr, shard_crops


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, bucket_bytes, offsets, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(n, generator=g)
        mine = flat.clone()
        red = BucketedAllReduce(flat, bucket_bytes=bucket_bytes, overlap=True)      # the bucketed, overlapping mode
        assert red.world == world
        # buckets tile [0, n) exactly, no overlap
        cover = np.zeros(n, np.int32)
        for s, e in red.buckets:
            cover[s:e] += 1
        assert (cover == 1).all()
        for step in range(2):                       # reset() makes the reducer reusable every step
            if step:
                flat.copy_(mine)
            red.reset()
            launched = []
            for off in offsets:                     # backward finishes the buffer from its end
                before = red.pending
                red.ready(off)
                launched.append(before - red.pending)
                # a bucket is launched only once it lies entirely inside the finished range [off, n): what is still
                # pending starts below `off` and has not been touched by any collective yet
                for s, e in red.buckets[:red.pending + 1]:
                    assert s < off
                    assert torch.equal(flat[s:e], mine[s:e])
                for s, e in red.buckets[red.pending + 1:]:
                    assert s >= off
            red.finish()
            assert red.pending == -1
            other = torch.randn(n, generator=torch.Generator().manual_seed(100 + (1 - rank)))
            assert torch.allclose(flat, mine + other, atol=1e-6)
        # default mode: nothing is launched from ready(), finish() reduces the whole buffer in one collective -- same result
        flat2 = mine.clone()
        red2 = BucketedAllReduce(flat2, bucket_bytes=bucket_bytes)
        assert not red2.overlap
        red2.reset()
        for off in offsets:
            red2.ready(off)
            assert torch.equal(flat2, mine) and not red2.works           # untouched until finish()
        red2.finish()
        assert torch.allclose(flat2, mine + other, atol=1e-6)
        if rank == 0:
            out.put(launched)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n,bucket_bytes', [(1000, 1024), (7760484 // 16, 1 << 20), (37, 1 << 20)])
def test_bucketed_all_reduce_gloo(n, bucket_bytes):
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    offsets = [int(n * f) for f in (0.9, 0.6, 0.6, 0.25, 0.0)]
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, bucket_bytes, offsets, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    launched = q.get()
    assert sum(launched) <= max(1, -(-n * 4 // bucket_bytes)) + 1


def test_single_process_reducer_is_a_noop():
    flat = torch.arange(10, dtype=torch.float32)
    r = BucketedAllReduce(flat, bucket_bytes=16)
    r.ready(5); r.finish()
    assert torch.equal(flat, torch.arange(10, dtype=torch.float32))


def test_shard_crops_partition():
    for gb in (16, 17, 5):
        for world in (1, 2, 4, 8):
            spans = [shard_crops(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_lr_schedule_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'misc.npz'))
    lr = np.array([get_cos_lr(int(s), period=200, peak=10, lr=1e-4) for s in g['lr_steps']])
    np.testing.assert_allclose(lr, g['lr'], rtol=1e-15)


def test_shard_of_a_global_batch_gives_counter_base_and_gradient_weight():
    """HipTrainStep.shard: the sampler's counter base is the shard's first global crop (NOT rank * B_local, which collides when
    shards differ by one crop) and the local mean-gradient counts B_local * world / B_global under the 1/world of the all-reduce."""
    import types
    from pnnp_amd._lib import PnnpError
    from pnnp_amd.trainer import HipTrainStep
    net = types.SimpleNamespace(engine=None)
    # weak scaling: every rank holds B crops
    assert HipTrainStep(net, rank=3, world=8).shard(16) == (48, 1.0)
    # strong scaling, 5 crops over 2 ranks: [0,3) and [3,5)
    a = HipTrainStep(net, rank=0, world=2, global_batch=5).shard(3)
    b = HipTrainStep(net, rank=1, world=2, global_batch=5).shard(2)
    assert a == (0, 3 * 2 / 5) and b == (3, 2 * 2 / 5)
    assert abs((a[1] + b[1]) / 2 - 1.0) < 1e-12                    # the weights average to 1 over the ranks
    covered = []
    for r in range(8):                                             # 13 crops over 8 ranks: bases tile [0, 13) without overlap
        lo, hi = shard_crops(13, r, 8)
        base, w = HipTrainStep(net, rank=r, world=8, global_batch=13).shard(hi - lo)
        assert base == lo and abs(w - (hi - lo) * 8 / 13) < 1e-12
        covered += list(range(base, base + hi - lo))
    assert covered == list(range(13))
    with pytest.raises(PnnpError):                                 # a rank handed the wrong number of crops is an error, not a collision
        HipTrainStep(net, rank=1, world=2, global_batch=5).shard(3)


def _worker8(rank, world, port, n, gb, out):
    """One of 8 ranks of config 5's data-parallel step on CPU: the rank owns shard_crops(gb) of a global batch, forms the mean "gradient" of
    its crops (a fixed per-crop vector standing in for the backward pass), weights it as HipTrainStep.shard prescribes, all-reduces through
    BucketedAllReduce (both modes) and applies Adam's 1/world: every rank must end with the mean over the GLOBAL batch."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lo, hi = shard_crops(gb, rank, world)
        per_crop = [torch.randn(n, generator=torch.Generator().manual_seed(1000 + c)) for c in range(gb)]
        want = torch.stack(per_crop).mean(0)
        weight = (hi - lo) * world / gb                                  # HipTrainStep.shard()[1]
        for overlap in (True, False):
            flat = torch.stack(per_crop[lo:hi]).mean(0) * weight         # the local mean-gradient, weighted (pnnp_l1_clamp_loss_w_f32's grad_weight)
            red = BucketedAllReduce(flat, bucket_bytes=1 << 12, overlap=overlap)
            assert red.world == world
            red.reset()
            for off in (int(n * 0.7), int(n * 0.3), 0):
                red.ready(off)
            red.finish()
            got = flat / world                                           # Adam's grad_scale = 1 / world
            assert torch.allclose(got, want, atol=2e-6), (rank, overlap, float((got - want).abs().max()))
        if rank == 0:
            out.put((lo, hi))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_config5_shards_reduce_to_the_global_mean():
    """the N = 8 logic rehearsed on CPU -- config 5's 12 crops over 8 ranks are shards of 2,2,2,2,1,1,1,1 crops;
    weighted local means, the bucketed all-reduce (overlapping and default mode) and Adam's 1/world give the mean over the global batch on
    every rank."""
    assert [shard_crops(12, r, 8) for r in range(8)] == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 9), (9, 10), (10, 11), (11, 12)]
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, 5000, 12, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get() == (0, 2)
