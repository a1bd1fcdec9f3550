"""Data-parallel train step on the GPU with the real engine: 2 ranks.  On a one-GPU box both ranks sit on cuda:0 with
the gloo backend (RCCL refuses two ranks on one device); where two devices are visible the same scenario also runs on
"nccl" (= RCCL), one device per rank: ReduceOp.AVG, async collectives on RCCL's stream against kernels launched through
the raw stream pointer, and the stream wait before the norm / AdamW kernels.  Checks that the bucketed
all-reduce driven by the backward watermarks produces exactly the single-process result for the
mean gradient, that replicas stay identical, and that tracker vectors are averaged across ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, accum=1, backend="gloo"):
    import sys
    for p in (os.path.join(ROOT, "vae-channel-dynamics_amd", "src"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    devno = rank if backend == "nccl" else 0
    torch.cuda.set_device(devno)  # before any other GPU call
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", devno))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import vae_oracle as vo
        from models.sdxl_vae_wrapper import SDXLVAEWrapper
        from tracking.monitor import ActivityMonitor
        from vaehip.trainer import HipTrainer
        dev = torch.device("cuda", devno)
        w = SDXLVAEWrapper("synthetic:7", device=dev)
        if rank == 1:  # replicas start different: the trainer must broadcast rank 0's parameters
            with torch.no_grad():
                w.vae.arena.flat.mul_(1.5)
        tr = HipTrainer(w, lr=1e-3, lr_warmup_steps=0, max_train_steps=100, kl_weight=1e-4, max_grad_norm=1.0,
                        scheduler_steps_per_update=1, bucket_mb=32.0,
                        gradient_accumulation_steps=accum)
        mon = ActivityMonitor(w, {"enabled": True, "track_interval": 1, "target_layers": [
            {"name": "vae.encoder.down_blocks.0.resnets.0.norm1", "capture_point": "output",
             "metrics": ["mean_abs_activation_per_channel"]}]})
        p0 = w.vae.arena.flat.clone()
        R, B = 32, 2
        nmb = world * accum  # micro-batches per update over all ranks
        xs = [vo.synthetic_pixels(B, R, 42, 10 + r).to(dev) for r in range(nmb)]
        es = [vo.synthetic_eps(B, R, 42, 10 + r).to(dev) for r in range(nmb)]
        for j in range(accum):
            assert tr.global_step == 0
            tr.train_step(xs[rank * accum + j], es[rank * accum + j])
        assert tr.global_step == 1 and tr.sync_gradients
        launched = list(tr.reducer.launched) if accum == 1 else []
        mon.step(1)
        stat = mon.get_data_for_step(1)["vae.encoder.down_blocks.0.resnets.0.norm1.output"]["mean_abs_activation_per_channel"]
        after = w.vae.arena.flat.clone()
        # single-process reference on this rank: same start, mean of the two ranks' gradients
        w2 = SDXLVAEWrapper("synthetic:7", device=dev)
        with torch.no_grad():
            w2.vae.arena.flat.copy_(p0)
        from vaehip.optim import FusedAdamW  # plain optimizer, no DP hooks
        opt = FusedAdamW(w2.vae, lr=1e-3, max_grad_norm=1.0)
        gsum = torch.zeros_like(w2.vae.arena.grad)
        stats = []
        for r in range(nmb):
            got = []
            h = w2.vae.engine.add_tracker(w2.vae.get_submodule("encoder.down_blocks.0.resnets.0.norm1"), "output", got.append)
            w2.vae.engine.forward_backward(xs[r], es[r], 1e-4)
            h.remove()
            stats.append(got[0])
            gsum += w2.vae.arena.grad
        w2.vae.arena.grad.copy_(gsum / nmb)
        opt.step()
        ref = w2.vae.arena.flat
        err = float((after - ref).abs().max() / ref.abs().max())
        stat_ref = (torch.stack(stats).mean(0)).cpu().numpy()  # monitor: mean over this step's forwards and over ranks
        stat_err = float(abs(stat - stat_ref).max() / abs(stat_ref).max())
        chk = after.double().sum().item()
        q.put((rank, err, stat_err, chk, launched, float((p0 - after).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("accum,backend", [(1, "gloo"), (2, "gloo"), (1, "nccl"), (3, "nccl")])
def test_two_rank_step_equals_mean_gradient_step(cuda, accum, backend):
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one device per rank; this box has one GPU (the driver's multi-GPU bench covers nccl)")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, accum, backend)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, e0, s0, c0, l0, d0), (r1, e1, s1, c1, l1, d1) = out
    assert e0 < 4e-6 and e1 < 4e-6, (e0, e1)       # == single-process step with the mean gradient
    assert c0 == c1                                 # replicas bitwise identical after the step
    assert s0 < 1e-6 and s1 < 1e-6                  # tracker vector = mean over ranks
    assert d0 > 0                                   # parameters moved
    if accum == 1:
        # buckets were launched from the top of the arena downwards and cover it exactly once
        assert l0 == l1 and l0[0][1] == 83_653_872 and l0[-1][0] == 0
        assert all(l0[i][0] == l0[i + 1][1] for i in range(len(l0) - 1))


def _one_rank_worker(port, q):
    import sys
    for p in (os.path.join(ROOT, "vae-channel-dynamics_amd", "src"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        import vae_oracle as vo
        from models.sdxl_vae_wrapper import SDXLVAEWrapper
        from vaehip.trainer import HipTrainer
        R, B = 32, 2
        xs = [vo.synthetic_pixels(B, R, 42, 10 + r).to(dev) for r in range(3)]
        es = [vo.synthetic_eps(B, R, 42, 10 + r).to(dev) for r in range(3)]
        out = {}
        for name, kw in (("exchange", dict(one_rank_exchange=True, time_comm=True, bucket_mb=32.0)), ("plain", {})):
            for accum in (1, 3):
                w = SDXLVAEWrapper("synthetic:7", device=dev)
                tr = HipTrainer(w, lr=1e-3, lr_warmup_steps=0, max_train_steps=100, kl_weight=1e-4, max_grad_norm=1.0,
                                gradient_accumulation_steps=accum, **kw)
                assert tr.exchanging == (name == "exchange")
                for j in range(3):
                    tr.train_step(xs[j], es[j])
                assert tr.global_step == (3 if accum == 1 else 1)
                torch.cuda.synchronize()
                out[(name, accum)] = w.vae.arena.flat.clone()
                if name == "exchange":
                    launched = list(tr.reducer.launched)
                    assert launched[0][1] == w.vae.arena.grad.numel() and launched[-1][0] == 0
                    assert all(launched[i][0] == launched[i + 1][1] for i in range(len(launched) - 1))
                    assert tr.reducer._avg  # the RCCL branch: ReduceOp.AVG, no host-side division
                    ms = tr.exposed_comm_ms()
                    assert 0.0 <= ms < 1e4
        same = all(torch.equal(out[("exchange", a)], out[("plain", a)]) for a in (1, 3))
        moved = float((out[("plain", 1)] - SDXLVAEWrapper("synthetic:7", device=dev).vae.arena.flat).abs().max())
        q.put((same, moved, dist.get_backend()))
    except BaseException:  # the parent must see the failure instead of waiting for the queue
        import traceback
        q.put((False, -1.0, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_group_runs_the_exchange(cuda):
    """The `nccl` (= RCCL) lines on a one-GPU box: a one-rank process group, the trainer told to exchange anyway
    (`one_rank_exchange`): communicator set-up with `device_id`, bucketed async `ReduceOp.AVG` all-reduces issued from the
    backward watermarks against kernels launched through the raw stream pointer, the stream waits of `finish()`, the
    accumulated-sum reducer of gradient accumulation, the parameter broadcast.  On one rank every collective is an identity, so the
    parameters after three steps must equal, bit for bit, those of a trainer that never touches torch.distributed."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), q))
    p.start()
    same, moved, backend = q.get(timeout=600)
    p.join(timeout=120)
    assert backend == "nccl", backend
    assert p.exitcode == 0
    assert same
    assert moved > 0
