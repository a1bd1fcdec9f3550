"""Fused clip + AdamW over the flat parameter arena (replaces clip_grad_norm_ + torch.optim.AdamW,
reference src/train.py:184-187,301-302).  One grad-norm reduction and one update kernel per step,
no host synchronisation: the clip coefficient is read from device memory by the update kernel.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    """A torch.optim.Optimizer (so LambdaLR / accelerate can drive `param_groups[0]['lr']`)
    whose step() is two HIP kernels over `vae.arena.flat` / `.grad`."""

    def __init__(self, vae, lr: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: float = 0.0):
        self._vae = vae
        params = list(vae.parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = float(max_grad_norm)
        self._arena_id = None
        self.exp_avg: Optional[torch.Tensor] = None
        self.exp_avg_sq: Optional[torch.Tensor] = None
        self.sqnorm: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None
        self.step_count = 0

    def _ensure(self):
        a = self._vae.arena
        if self._arena_id != id(a) or self.exp_avg is None or self.exp_avg.device != a.flat.device:
            old_m, old_v = self.exp_avg, self.exp_avg_sq
            self.exp_avg = torch.zeros_like(a.flat)
            self.exp_avg_sq = torch.zeros_like(a.flat)
            if old_m is not None and old_m.numel() == a.flat.numel():
                self.exp_avg.copy_(old_m)
                self.exp_avg_sq.copy_(old_v)
            self.sqnorm = torch.zeros(1, device=a.flat.device, dtype=torch.float32)
            self._ws = torch.empty(2048, device=a.flat.device, dtype=torch.float32)
            self._arena_id = id(a)
        return a

    def clip_grad_norm_(self, max_norm: float):
        """accelerate-style entry: just arms the fused clip for the next step()."""
        self.max_grad_norm = float(max_norm)

    @torch.no_grad()
    def step(self, closure=None):
        a = self._ensure()
        if a.flat.device.type != "cuda":
            raise RuntimeError("FusedAdamW needs the parameter arena on the GPU (no CPU fallback)")
        g = self.param_groups[0]
        self.step_count += 1
        ops.sqnorm(a.grad, self.sqnorm, self._ws)
        ops.adamw(a.flat, a.grad, self.exp_avg, self.exp_avg_sq, self.sqnorm, self.max_grad_norm, g["lr"],
                  g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_count)

    def zero_grad(self, set_to_none: bool = True):
        # gradients are overwritten (not accumulated) by engine.forward_backward: nothing to clear
        return None

    def grad_norm(self) -> torch.Tensor:
        """device scalar: global L2 norm of the last step's (unclipped) gradients."""
        return torch.sqrt(self.sqnorm[0])

    def state_dict(self):
        self._ensure()
        return {"step": self.step_count, "exp_avg": self.exp_avg.detach().cpu(), "exp_avg_sq": self.exp_avg_sq.detach().cpu(),
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}],
                "max_grad_norm": self.max_grad_norm}

    def load_state_dict(self, sd):
        self._ensure()
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for k, v in sd["param_groups"][0].items():
            self.param_groups[0][k] = v
        self.max_grad_norm = float(sd.get("max_grad_norm", self.max_grad_norm))
