"""One optimizer step of the reference's hot loop (src/train.py:283-306) on the HIP engine:
fwd + loss + bwd (+ bucketed RCCL gradient mean) + fused clip/AdamW + LambdaLR, no host sync."""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist

from . import ops
from .dp import GradBucketReducer, allreduce_mean_, broadcast_params
from .optim import FusedAdamW


def lr_lambda_factory(warmup: int, max_steps: int) -> Callable[[int], float]:
    """train.py:197-200 (note: gives lr = 0 on the very first optimizer step)."""
    def fn(step: int) -> float:
        if step < warmup:
            return float(step) / float(max(1, warmup))
        progress = float(step - warmup) / float(max(1, max_steps - warmup))
        return max(0.0, 1.0 - min(1.0, progress))
    return fn


class HipTrainer:
    def __init__(self, wrapper, *, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=1.0,
                 kl_weight=1e-6, lr_warmup_steps=100, max_train_steps=1000, scheduler_steps_per_update: int = 1,
                 bucket_mb: float = 64.0, generator: Optional[torch.Generator] = None, mixed_precision: str = "no",
                 gradient_accumulation_steps: int = 1, checkpoint_decoder: bool = False, time_comm: bool = False,
                 one_rank_exchange: bool = False):
        self.wrapper = wrapper
        self.vae = wrapper.vae
        self.kl_weight = float(kl_weight)
        self.generator = generator
        self.vae.engine.set_precision(mixed_precision)
        self.vae.engine.checkpoint_decoder = bool(checkpoint_decoder)
        # accelerator.accumulate (train.py:286): the loss of every micro-batch is divided by N, gradients add up over N
        # micro-batches, ranks exchange only on the N-th, and clip / AdamW / LR schedule run once per N micro-batches
        self.accum_steps = int(gradient_accumulation_steps)
        if self.accum_steps < 1:
            raise ValueError("gradient_accumulation_steps must be >= 1")
        self.micro_step = 0
        self._accum: Optional[torch.Tensor] = None
        self.bucket_mb = float(bucket_mb)
        self.time_comm = bool(time_comm)
        self.optimizer = FusedAdamW(self.vae, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                    max_grad_norm=max_grad_norm)
        self.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lr_lambda_factory(lr_warmup_steps, max_train_steps))
        # accelerate steps the scheduler num_processes times per optimizer step (accelerate/scheduler.py:72-82)
        self.scheduler_steps_per_update = int(scheduler_steps_per_update)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # the gradient exchange runs when there is someone to exchange with -- or, as a self-test of the RCCL path on a one-GPU
        # box (dp.GradBucketReducer.one_rank_exchange), on a one-rank process group
        self.one_rank_exchange = bool(one_rank_exchange) and dist.is_initialized()
        self.exchanging = self.world > 1 or self.one_rank_exchange
        self.reducer: Optional[GradBucketReducer] = None   # made on first use (arena.grad exists after the first backward)
        self._accum_reducer: Optional[GradBucketReducer] = None
        if self.exchanging:
            broadcast_params(self.vae.arena.flat, 0, one_rank_exchange=self.one_rank_exchange)
        self.global_step = 0
        self.last = None

    @property
    def sync_gradients(self) -> bool:
        """True when the LAST train_step call ended with an optimizer update (accelerator.sync_gradients)."""
        return self.micro_step == 0

    def _make_reducer(self, flat: torch.Tensor) -> GradBucketReducer:
        return GradBucketReducer(flat, bucket_mb=self.bucket_mb, time_finish=self.time_comm, one_rank_exchange=self.one_rank_exchange)

    def _accumulating_step(self, pixel_values, eps, end_of_dataloader: bool):
        """one micro-batch of an N-micro-batch update; returns (result, update_due).  The update is due on the N-th
        micro-batch, or on the LAST batch of a dataloader pass whatever the count (accelerate's `accumulate` forces
        sync_gradients there, accelerator.py:_do_sync; the 1/N loss scale stays, as in accelerate).
        Exchange (world > 1): the mean over ranks is linear, so the sum of the earlier micro-batches (`_accum`) is
        all-reduced in buckets from the START of the due micro-batch -- under its whole forward + backward -- and that
        micro-batch's own gradient goes through the watermark-driven reducer under its backward; the two means are then
        added.  Nothing is exchanged on the other micro-batches (DDP no_sync)."""
        eng = self.vae.engine
        grad = self.vae.arena.grad
        due = (self.micro_step + 1 == self.accum_steps) or bool(end_of_dataloader)
        first = self.micro_step == 0
        exchange = due and self.exchanging
        acc_red = None
        if exchange:
            if not first:
                if self._accum_reducer is None or self._accum_reducer.flat.data_ptr() != self._accum.data_ptr():
                    self._accum_reducer = self._make_reducer(self._accum)
                acc_red = self._accum_reducer
                acc_red.begin()
                acc_red.ready(0)  # every bucket of the earlier micro-batches' sum is final: all in flight now
            if self.reducer is None or self.reducer.flat.data_ptr() != grad.data_ptr():
                self.reducer = self._make_reducer(grad)
            self.reducer.begin()
            eng.reducer = self.reducer
        try:
            res = eng.forward_backward(pixel_values, eps, self.kl_weight, True, self.generator, grad_scale=1.0 / self.accum_steps)
        finally:
            eng.reducer = None
        grad = self.vae.arena.grad
        if exchange:
            self.reducer.finish()
            if acc_red is not None:
                acc_red.finish()
        self.micro_step += 1
        if first:
            if not due:
                if self._accum is None or self._accum.shape != grad.shape:
                    self._accum = torch.empty_like(grad)
                self._accum.copy_(grad)
            return res, due            # a one-micro-batch update (flush right after an update): grad is already the sum
        dst = grad if due else self._accum   # the sum ends up in arena.grad, where the optimizer reads it
        self._add(self._accum, grad, dst)
        return res, due

    @staticmethod
    def _add(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor):
        ops.lib.call("vae_add", ops._p(a), ops._p(b), a.numel(), ops._p(out), ops._stream())

    def train_step(self, pixel_values: torch.Tensor, eps: Optional[torch.Tensor] = None, end_of_dataloader: bool = False):
        """one micro-batch; returns the engine result dict, result['scalars'] = device tensor [mse, kl, total] of THIS
        rank.  With gradient_accumulation_steps = N the optimizer / scheduler / global_step advance on every N-th call
        and on a call with end_of_dataloader=True (`sync_gradients` tells which)."""
        eng = self.vae.engine
        if self.accum_steps > 1:
            res, due = self._accumulating_step(pixel_values, eps, end_of_dataloader)
            self.last = res
            if not due:
                return res
            self.micro_step = 0
        else:
            if self.exchanging:
                if self.reducer is None or self.reducer.flat.data_ptr() != self.vae.arena.grad.data_ptr():
                    self.reducer = self._make_reducer(self.vae.arena.grad)
                self.reducer.begin()
                eng.reducer = self.reducer
            try:
                res = eng.forward_backward(pixel_values, eps, self.kl_weight, True, self.generator)
            finally:
                eng.reducer = None
            if self.exchanging:
                self.reducer.finish()
        self._update()
        self.last = res
        return res

    @property
    def pending_micro_batches(self) -> int:
        """micro-batches accumulated since the last optimizer update"""
        return self.micro_step

    def flush(self):
        """optimizer update from the micro-batches accumulated so far, without a new one (a dataloader pass that ends
        on a batch every rank skipped).  No-op when nothing is pending."""
        if self.micro_step == 0:
            return
        grad = self.vae.arena.grad
        grad.copy_(self._accum)
        if self.exchanging:
            allreduce_mean_(grad, one_rank_exchange=self.one_rank_exchange)
        self.micro_step = 0
        self._update()

    def _update(self):
        self.optimizer.step()
        for _ in range(self.scheduler_steps_per_update):
            self.lr_scheduler.step()
        self.global_step += 1

    def exposed_comm_ms(self) -> float:
        """time the compute stream waited in reducer.finish() since the last call (0 when the exchange was hidden under
        the backward pass or the world is 1); needs time_comm=True; synchronises the device"""
        return sum(r.exposed_ms() for r in (self.reducer, self._accum_reducer) if r is not None)

    @torch.no_grad()
    def eval_step(self, pixel_values: torch.Tensor):
        """validation forward (train.py:75-78): deterministic latents, sum-reduced MSE and KL."""
        res = self.vae.engine.forward_eval(pixel_values, None, False, self.kl_weight)
        n = pixel_values.numel()
        b = pixel_values.shape[0]
        sc = res["scalars"]
        return {"rec_sum": sc[0] * n, "kl_sum": sc[1] * b, **res}
