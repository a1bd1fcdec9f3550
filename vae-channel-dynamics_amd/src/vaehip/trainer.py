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
                 gradient_accumulation_steps: int = 1, checkpoint_decoder: bool = False):
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
        self.optimizer = FusedAdamW(self.vae, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                    max_grad_norm=max_grad_norm)
        self.lr_scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lr_lambda_factory(lr_warmup_steps, max_train_steps))
        # accelerate steps the scheduler num_processes times per optimizer step (accelerate/scheduler.py:72-82)
        self.scheduler_steps_per_update = int(scheduler_steps_per_update)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.reducer = None
        if self.world > 1:
            broadcast_params(self.vae.arena.flat, 0)
            self.reducer = GradBucketReducer(self.vae.arena.grad, bucket_mb=bucket_mb)
        self.global_step = 0
        self.last = None

    @property
    def sync_gradients(self) -> bool:
        """True when the LAST train_step call ended with an optimizer update (accelerator.sync_gradients)."""
        return self.micro_step == 0

    def _accumulating_step(self, pixel_values, eps):
        """one micro-batch of an N-micro-batch update; returns (result, update_due)."""
        eng = self.vae.engine
        res = eng.forward_backward(pixel_values, eps, self.kl_weight, True, self.generator, grad_scale=1.0 / self.accum_steps)
        grad = self.vae.arena.grad
        self.micro_step += 1
        if self.micro_step == 1:
            if self._accum is None or self._accum.shape != grad.shape:
                self._accum = torch.empty_like(grad)
            self._accum.copy_(grad)
            return res, False
        last = self.micro_step == self.accum_steps
        dst = grad if last else self._accum   # the sum ends up in arena.grad, where the optimizer reads it
        ops.lib.call("vae_add", ops._p(self._accum), ops._p(grad), grad.numel(), ops._p(dst), ops._stream())
        if last and self.world > 1:           # one exchange per update (no_sync on the other micro-batches)
            allreduce_mean_(grad)
        return res, last

    def train_step(self, pixel_values: torch.Tensor, eps: Optional[torch.Tensor] = None):
        """one micro-batch; returns the engine result dict, result['scalars'] = device tensor [mse, kl, total] of THIS
        rank.  With gradient_accumulation_steps = N the optimizer / scheduler / global_step advance on every N-th call
        (`sync_gradients` tells which)."""
        eng = self.vae.engine
        if self.accum_steps > 1:
            res, due = self._accumulating_step(pixel_values, eps)
            self.last = res
            if not due:
                return res
            self.micro_step = 0
        else:
            if self.reducer is not None:
                if self.reducer.flat.data_ptr() != self.vae.arena.grad.data_ptr():
                    self.reducer = GradBucketReducer(self.vae.arena.grad, bucket_mb=self.bucket_mb)
                self.reducer.begin()
                eng.reducer = self.reducer
            try:
                res = eng.forward_backward(pixel_values, eps, self.kl_weight, True, self.generator)
            finally:
                eng.reducer = None
            if self.reducer is not None:
                self.reducer.finish()
        self.optimizer.step()
        for _ in range(self.scheduler_steps_per_update):
            self.lr_scheduler.step()
        self.global_step += 1
        self.last = res
        return res

    @torch.no_grad()
    def eval_step(self, pixel_values: torch.Tensor):
        """validation forward (train.py:75-78): deterministic latents, sum-reduced MSE and KL."""
        res = self.vae.engine.forward_eval(pixel_values, None, False, self.kl_weight)
        n = pixel_values.numel()
        b = pixel_values.shape[0]
        sc = res["scalars"]
        return {"rec_sum": sc[0] * n, "kl_sum": sc[1] * b, **res}
