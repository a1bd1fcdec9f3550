"""Data parallelism for the flat gradient arena: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests).

Replaces the DDP wrap created by accelerate.prepare in the reference (src/train.py:204-211):
the gradient mean is a handful of large all-reduces over contiguous arena slices, launched from
the END of the arena (decoder, whose gradients are final first) while the encoder backward is
still running.  Device-agnostic on purpose: the planner and collectives are tested on CPU tensors.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def plan_buckets(total: int, bucket_elems: int) -> List[Tuple[int, int]]:
    """[lo, hi) slices covering [0,total) from the end, each ~bucket_elems (multiple of 4 elements)."""
    bucket_elems = max(4, (bucket_elems // 4) * 4)
    out = []
    hi = total
    while hi > 0:
        lo = max(0, hi - bucket_elems)
        lo -= lo % 4
        out.append((lo, hi))
        hi = lo
    return out


class GradBucketReducer:
    """engine calls ready(low): every gradient at arena offset >= low is final.  Buckets whose
    whole range is final are all-reduced asynchronously; finish() waits and averages."""

    def __init__(self, flat_grad: torch.Tensor, group=None, bucket_mb: float = 64.0, time_finish: bool = False,
                 one_rank_exchange: bool = False):
        self.flat = flat_grad
        # one_rank_exchange: issue the collectives on a ONE-rank group as well (they are identities there).  A self-test of the
        # RCCL lines on a one-GPU box -- communicator, ReduceOp.AVG, async work objects against kernels launched through the raw
        # stream pointer, the stream waits of finish() -- not something a training run sets.
        self.one_rank_exchange = bool(one_rank_exchange)
        # exposed communication: how long the compute stream sits in finish() waiting for collectives that the
        # backward pass did not hide.  On the GPU a work.wait() only makes the stream wait (the host runs on), so the
        # figure comes from two events around the waits; read it with exposed_ms() after a synchronize.
        self.time_finish = bool(time_finish)
        self._events = []
        self.last_finish_ms = 0.0
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (self.one_rank_exchange and dist.is_initialized())
        self.buckets = plan_buckets(flat_grad.numel(), int(bucket_mb * (1 << 20) / 4))
        self._next = 0
        self._works = []
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self._avg = backend == "nccl"
        self.launched: List[Tuple[int, int]] = []

    def begin(self):
        self._next = 0
        self._works = []
        self.launched = []

    def ready(self, low: int):
        if not self.active:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= low:
            lo, hi = self.buckets[self._next]
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            w = dist.all_reduce(self.flat[lo:hi], op=op, group=self.group, async_op=True)
            self._works.append((w, lo, hi))
            self.launched.append((lo, hi))
            self._next += 1

    def finish(self):
        if not self.active:
            return
        self.ready(0)
        ev = None
        t0 = 0.0
        if self.time_finish:
            if self.flat.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            else:
                import time
                t0 = time.perf_counter()
        for w, lo, hi in self._works:
            w.wait()
        if ev is not None:
            ev[1].record()
            self._events.append(ev)
        elif self.time_finish:
            import time
            self.last_finish_ms = (time.perf_counter() - t0) * 1e3
        if not self._avg:
            for w, lo, hi in self._works:
                self.flat[lo:hi].mul_(1.0 / self.world)
        self._works = []

    def exposed_ms(self, reset: bool = True) -> float:
        """sum over the finish() calls since the last reset of the time the compute stream waited for collectives
        (synchronises the device)."""
        if self._events:
            torch.cuda.synchronize(self.flat.device)
            tot = sum(a.elapsed_time(b) for a, b in self._events)
            if reset:
                self._events = []
            return float(tot)
        return float(self.last_finish_ms)


def broadcast_params(flat: torch.Tensor, src: int = 0, group=None, one_rank_exchange: bool = False):
    """initial replica sync (DDP does this at wrap time) and re-sync after an intervention
    (fixes the reference's rank-0-only nudge divergence)."""
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or one_rank_exchange):
        dist.broadcast(flat, src=src, group=group)


def allreduce_mean_(t: torch.Tensor, group=None, one_rank_exchange: bool = False) -> torch.Tensor:
    """coalesced logging scalars / tracker vectors (train.py:292-294 uses 3 gathers + 3 .item())."""
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or one_rank_exchange):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.mul_(1.0 / dist.get_world_size(group))
    return t
