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


def _accum_worker(rank, world, port, q):
    """HipTrainer's accumulated update across ranks with the device kernels replaced by host stand-ins: the sum of the
    earlier micro-batches is all-reduced from the start of the due micro-batch, that micro-batch's own gradient through
    the watermark reducer, and the two means add up to the mean over ranks of the window's sum."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from models.sdxl_vae_wrapper import SDXLVAEWrapper
        from vaehip.trainer import HipTrainer
        w = SDXLVAEWrapper("synthetic:%d" % (rank + 1))  # replicas start different: the trainer broadcasts rank 0's
        tr = HipTrainer(w, lr=1.0, lr_warmup_steps=0, max_train_steps=10, gradient_accumulation_steps=3, bucket_mb=64.0,
                        time_comm=True)
        chk = float(w.vae.arena.flat.double().sum())
        grad = w.vae.arena.grad
        eng = w.vae.engine
        seen = []

        def fake_fwd_bwd(pv, eps, klw, sample, gen, grad_scale=1.0):
            grad.fill_(float(pv) * grad_scale)
            if eng.reducer is not None:   # what the tape does: watermarks from the top of the arena down
                eng.reducer.ready(grad.numel() // 2)
                eng.reducer.ready(0)
            return {"scalars": torch.zeros(3)}
        eng.forward_backward = fake_fwd_bwd
        tr._add = lambda a, b, out: torch.add(a, b, out=out)
        tr.optimizer.step = lambda: seen.append((float(grad[0]), float(grad[-1]), float(grad.min()), float(grad.max())))
        vals = [[3.0, 6.0, 9.0, 12.0], [30.0, 60.0, 90.0, 120.0]][rank]
        tr.train_step(vals[0]); tr.train_step(vals[1]); tr.train_step(vals[2])       # full window
        tr.train_step(vals[3], end_of_dataloader=True)                                # partial window of one
        q.put((rank, chk, seen, tr.global_step, tr.exposed_comm_ms() >= 0.0))
    finally:
        dist.destroy_process_group()


def test_accumulated_update_exchange_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, c0, s0, g0, t0), (_, c1, s1, g1, t1) = out
    assert c0 == c1                                   # parameters broadcast from rank 0
    assert g0 == g1 == 2 and t0 and t1
    want1 = ((3 + 6 + 9) / 3 + (30 + 60 + 90) / 3) / 2  # mean over ranks of the window's 1/N-scaled sum
    want2 = (12 / 3 + 120 / 3) / 2
    for s in (s0, s1):
        assert s[0] == pytest.approx((want1,) * 4) and s[1] == pytest.approx((want2,) * 4)


def _agree_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import train
        good = {"pixel_values": torch.zeros(2, 3, 8, 8)}
        # rank 1's second batch fails to decode (None), rank 0's fourth has no samples: both ranks must skip both
        batches = [good, None if rank == 1 else good, good, {"pixel_values": torch.zeros(0, 3, 8, 8)} if rank == 0 else good, good]
        seen = [(ok, last, b is not None and b["pixel_values"].shape[0] > 0) for b, ok, last in train.agreed_batches(batches, world)]
        q.put((rank, seen, list(train.agreed_batches([], world))))
    finally:
        dist.destroy_process_group()


def test_batch_validity_is_agreed_one_step_ahead_world2():
    """train.py's control plane: a batch is trained only when it is valid on EVERY rank (the reference crashes on a None batch,
    data_utils.py:215; with one rank skipping alone the gradient exchange would hang); the agreement runs as an asynchronous
    gloo all-reduce started one batch ahead"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, seen, empty in out:
        assert [s[0] for s in seen] == [True, False, True, False, True], (rank, seen)   # the same decisions on both ranks
        assert [s[1] for s in seen] == [False, False, False, False, True] and empty == []
    assert {r: [s[2] for s in seen] for r, seen, _ in out} == {0: [True, True, True, False, True], 1: [True, False, True, True, True]}


def test_batch_agreement_without_a_process_group():
    import train
    good = {"pixel_values": torch.zeros(1, 3, 8, 8)}
    assert [(ok, last) for _, ok, last in train.agreed_batches([good, None, good], 1)] == [(True, False), (False, False), (True, True)]
