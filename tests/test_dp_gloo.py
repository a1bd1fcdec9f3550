"""N>1 data-parallel path on CPU: world_size 2, gloo.  Covers the gradient bucket reducer driven by
the engine's backward watermarks, the initial parameter broadcast, the coalesced scalar mean and
the cross-rank averaging of tracker vectors (fix of the reference's rank-0-only statistics)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vaehip.dp import GradBucketReducer, allreduce_mean_, broadcast_params
        n = 10_000
        flat = torch.full((n,), float(rank + 1))
        params = torch.arange(n, dtype=torch.float32) * (rank + 1)
        broadcast_params(params, 0)
        ok_bcast = torch.equal(params, torch.arange(n, dtype=torch.float32))
        red = GradBucketReducer(flat, bucket_mb=4096 * 4 / (1 << 20))  # 4096-element buckets
        red.begin()
        # the engine reports watermarks from the end of the arena downwards, not on bucket boundaries
        red.ready(9000)
        launched_a = list(red.launched)
        red.ready(5000)
        red.ready(4097)
        launched_b = list(red.launched)
        red.finish()
        ok_mean = bool(torch.allclose(flat, torch.full((n,), (1 + world) * world / 2 / world)))
        sc = allreduce_mean_(torch.tensor([float(rank), 2.0 * rank, 1.0]))
        # second step re-uses the reducer
        flat.fill_(float(rank))
        red.begin()
        red.finish()
        ok_again = bool(torch.allclose(flat, torch.full((n,), (world - 1) / 2)))
        q.put((rank, ok_bcast, ok_mean, ok_again, launched_a, launched_b, red.buckets, sc.tolist()))
    finally:
        dist.destroy_process_group()


def test_bucketed_gradient_mean_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_bcast, ok_mean, ok_again, la, lb, buckets, sc in out:
        assert ok_bcast and ok_mean and ok_again, rank
        assert buckets[0] == (5904, 10000) and buckets[-1][0] == 0
        assert la == []                       # nothing final yet: the top bucket starts at 5904 < 9000
        assert lb == [(5904, 10000)]          # [1808,5904) is not final at watermark 4097
        assert sc == pytest.approx([0.5, 1.0, 1.0])
