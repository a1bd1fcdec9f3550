"""CPU oracle for the SDXL-VAE train-step hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product path (vae-channel-dynamics_amd/) never does; it fails
loudly when libvaehip.so is missing.

What it restates (reference = olegroshka/vae-channel-dynamics, paths relative
to the reference root):

* model forward composition   src/models/sdxl_vae_wrapper.py:42-77
  (encode -> sample/mode -> decode, NO scaling factor applied)
* loss                        src/train.py:289-291
* validation reductions       src/train.py:53-97
* clip + AdamW + LambdaLR     src/train.py:184-202,300-304
* tracker metric              src/tracking/monitor.py:56-80,146-216

The arithmetic of the VAE itself lives in the third-party `diffusers`
package (requirements.txt:8, unpinned; the sdxl-vae config.json records
_diffusers_version 0.18.0.dev0).  diffusers is not installed in the build
container nor vendored in the reference, and no SDXL-VAE weights exist offline,
so the network below is a restatement of diffusers' published
AutoencoderKL/Encoder/Decoder/ResnetBlock2D/Downsample2D/Upsample2D/Attention/
DiagonalGaussianDistribution algorithm written from the layer table in
SURVEY.md section 8a.  Structural anchors that ARE checked: parameter count
83,653,863 and the module names used by the reference's YAMLs.

PARITY STATUS: the tracker / classifier / nudger / dead-weight semantics are
pinned by golden vectors produced by importing the reference's own torch-only
modules against this module tree (tests/golden/make_golden.py).  The VAE
numerics at the diffusers boundary are **parity unpinned** (no fixture of the
reference pins conv/GroupNorm/attention outputs).

Module names follow diffusers exactly so the reference's ActivityMonitor /
RegionClassifier / InterventionHandler / DeadNeuronTracker run on it unchanged.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

SDXL_VAE_CONFIG = dict(
    in_channels=3, out_channels=3, latent_channels=4,
    block_out_channels=(128, 256, 512, 512), layers_per_block=2,
    norm_num_groups=32, scaling_factor=0.13025, sample_size=1024,
)
GN_EPS = 1e-6


# --------------------------------------------------------------------------
# portable deterministic generator (counter hash -> float); no RNG state,
# identical on every platform, so 335 MB of weights never need committing.
# --------------------------------------------------------------------------
def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    x = (x ^ (x >> np.uint64(16))) * np.uint64(0x7FEB352D) & np.uint64(0xFFFFFFFF)
    x = (x ^ (x >> np.uint64(15))) * np.uint64(0x846CA68B) & np.uint64(0xFFFFFFFF)
    x = x ^ (x >> np.uint64(16))
    return x


def hash_uniform(n: int, seed: int, stream: int) -> np.ndarray:
    """n floats in [-1, 1), function of (seed, stream, index) only."""
    idx = np.arange(n, dtype=np.uint64)
    key = np.uint64((seed * 0x9E3779B1 + stream * 0x85EBCA77 + 0x1234567) & 0xFFFFFFFF)
    h = _mix32((idx * np.uint64(0x2545F491) + key) & np.uint64(0xFFFFFFFF))
    h = _mix32((h + key + idx) & np.uint64(0xFFFFFFFF))
    # 24 significant bits -> exactly representable in fp32
    return ((h >> np.uint64(8)).astype(np.float64) / float(1 << 23) - 1.0).astype(np.float32)


def hash_normal(n: int, seed: int, stream: int) -> np.ndarray:
    """approx N(0,1): sum of 4 uniforms scaled (Irwin-Hall); exact fp32 ops in fp64 then cast."""
    acc = np.zeros(n, dtype=np.float64)
    for j in range(4):
        acc += hash_uniform(n, seed, stream * 4 + j + 1000).astype(np.float64)
    return (acc * math.sqrt(3.0 / 4.0)).astype(np.float32)


def _stream_id(name: str) -> int:
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h & 0x7FFFFFFF


def synthetic_state_dict(model: nn.Module, seed: int = 42) -> Dict[str, torch.Tensor]:
    """Kaiming-uniform-like deterministic init (bound = 1/sqrt(fan_in)), GN gamma=1 beta=0.

    Values are generated in the parameter's logical (O,I,H,W) row-major order.
    """
    out = {}
    for name, p in model.named_parameters():
        shape = tuple(p.shape)
        n = p.numel()
        if name.endswith("weight") and p.ndim == 1:      # GroupNorm gamma
            # 1 + small deterministic jitter so per-channel stats are not degenerate
            v = 1.0 + 0.25 * hash_uniform(n, seed, _stream_id(name))
        elif name.endswith("bias") and ("norm" in name.split(".")[-2]):
            v = 0.1 * hash_uniform(n, seed, _stream_id(name))
        else:
            fan_in = int(np.prod(shape[1:])) if p.ndim > 1 else None
            if fan_in is None:  # conv/linear bias: fan_in of its weight
                wname = name[: -len("bias")] + "weight"
                w = dict(model.named_parameters())[wname]
                fan_in = int(np.prod(tuple(w.shape)[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            v = bound * hash_uniform(n, seed, _stream_id(name))
        out[name] = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32))).reshape(shape).clone()
    return out


def synthetic_pixels(batch: int, res: int, seed: int = 42, step: int = 0) -> torch.Tensor:
    """B x 3 x R x R uniform in [-1,1) (matches Normalize(0.5,0.5) range, reference data_utils.py:28-29)."""
    v = hash_uniform(batch * 3 * res * res, seed, 777 + step)
    return torch.from_numpy(v).reshape(batch, 3, res, res).clone()


def synthetic_eps(batch: int, res: int, seed: int = 42, step: int = 0) -> torch.Tensor:
    """noise for latent_dist.sample(): B x 4 x R/8 x R/8."""
    v = hash_normal(batch * 4 * (res // 8) * (res // 8), seed, 999 + step)
    return torch.from_numpy(v).reshape(batch, 4, res // 8, res // 8).clone()


# --------------------------------------------------------------------------
# network (diffusers module names)
# --------------------------------------------------------------------------
class ResnetBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=GN_EPS, affine=True)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.norm2 = nn.GroupNorm(32, cout, eps=GN_EPS, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = nn.Conv2d(cin, cout, 1, 1, 0) if cin != cout else None

    def forward(self, x):
        h = self.conv1(self.nonlinearity(self.norm1(x)))
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 2, 0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0.0))


class Upsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 1, 1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Attention(nn.Module):
    """single head, d = C, scale C^-0.5, residual, GroupNorm in front."""

    def __init__(self, c: int):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, c, eps=GN_EPS, affine=True)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        r = x
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        s = torch.bmm(q, k.transpose(1, 2)) * (c ** -0.5)
        p = torch.softmax(s, dim=-1)
        o = torch.bmm(p, v)
        o = self.to_out[1](self.to_out[0](o))
        return o.transpose(1, 2).reshape(b, c, h, w) + r


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout) for i in range(n)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout) for i in range(n)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNetMidBlock2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attentions = nn.ModuleList([Attention(c)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c), ResnetBlock2D(c, c)])

    def forward(self, x):
        x = self.resnets[0](x)
        x = self.attentions[0](x)
        return self.resnets[1](x)


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc = cfg["block_out_channels"]
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, 1, 1)
        self.down_blocks = nn.ModuleList()
        c = boc[0]
        for i, co in enumerate(boc):
            self.down_blocks.append(DownEncoderBlock2D(c, co, cfg["layers_per_block"], i != len(boc) - 1))
            c = co
        self.mid_block = UNetMidBlock2D(c)
        self.conv_norm_out = nn.GroupNorm(32, c, eps=GN_EPS)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(c, 2 * cfg["latent_channels"], 3, 1, 1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc = cfg["block_out_channels"]
        rev = list(reversed(boc))
        self.conv_in = nn.Conv2d(cfg["latent_channels"], rev[0], 3, 1, 1)
        self.up_blocks = nn.ModuleList()
        self.mid_block = UNetMidBlock2D(rev[0])
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(UpDecoderBlock2D(c, co, cfg["layers_per_block"] + 1, i != len(boc) - 1))
            c = co
        self.conv_norm_out = nn.GroupNorm(32, c, eps=GN_EPS)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(c, cfg["out_channels"], 3, 1, 1)

    def forward(self, z):
        x = self.conv_in(z)
        x = self.mid_block(x)
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class DiagonalGaussianDistribution:
    """reference call sites: sdxl_vae_wrapper.py:64,66; train.py:78,290."""

    def __init__(self, moments: torch.Tensor, eps: Optional[torch.Tensor] = None):
        self.mean, logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        self._eps = eps

    def sample(self):
        eps = self._eps if self._eps is not None else torch.randn_like(self.mean)
        return self.mean + self.std * eps

    def mode(self):
        return self.mean

    def kl(self):
        return 0.5 * torch.sum(self.mean.pow(2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class _Out:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class OracleAutoencoderKL(nn.Module):
    def __init__(self, cfg: Optional[dict] = None):
        super().__init__()
        cfg = dict(SDXL_VAE_CONFIG if cfg is None else cfg)
        self.config = _Cfg(cfg)
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        lc = cfg["latent_channels"]
        self.quant_conv = nn.Conv2d(2 * lc, 2 * lc, 1)
        self.post_quant_conv = nn.Conv2d(lc, lc, 1)
        self._eps_next: Optional[torch.Tensor] = None

    def encode(self, x):
        m = self.quant_conv(self.encoder(x))
        d = DiagonalGaussianDistribution(m, self._eps_next)
        return _Out(latent_dist=d)

    def decode(self, z):
        return _Out(sample=self.decoder(self.post_quant_conv(z)))


class OracleWrapper(nn.Module):
    """restates SDXLVAEWrapper.forward (sdxl_vae_wrapper.py:42-77)."""

    def __init__(self, seed: int = 42, cfg: Optional[dict] = None):
        super().__init__()
        self.vae = OracleAutoencoderKL(cfg)
        self.vae.load_state_dict(synthetic_state_dict(self.vae, seed))
        self.scaling_factor = self.vae.config.scaling_factor

    def forward(self, pixel_values, sample_posterior: bool = True, eps: Optional[torch.Tensor] = None):
        self.vae._eps_next = eps
        latent_dist = self.vae.encode(pixel_values).latent_dist
        latents = latent_dist.sample() if sample_posterior else latent_dist.mode()
        recon = self.vae.decode(latents).sample
        return {"reconstruction": recon, "latent_dist": latent_dist, "latents_sampled": latents}


# --------------------------------------------------------------------------
# train step pieces
# --------------------------------------------------------------------------
def lr_lambda(step: int, warmup: int, max_steps: int) -> float:
    """train.py:197-200."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, max_steps - warmup))
    return max(0.0, 1.0 - min(1.0, progress))


def losses(out, pixel_values, kl_weight: float):
    """train.py:289-291."""
    rec = F.mse_loss(out["reconstruction"].float(), pixel_values.float(), reduction="mean")
    kl = out["latent_dist"].kl().mean()
    return rec, kl, rec + kl_weight * kl


def mean_abs_per_channel(t: torch.Tensor) -> np.ndarray:
    """monitor.py:66-67."""
    return t.abs().mean(dim=[0] + list(range(2, t.ndim))).detach().cpu().numpy()


class OracleTrainer:
    """fwd + loss + bwd + clip + AdamW + LambdaLR exactly as train.py:184-202,283-306."""

    def __init__(self, wrapper: OracleWrapper, lr=1e-5, betas=(0.9, 0.999), wd=1e-2, eps=1e-8,
                 warmup=100, max_steps=1000, kl_weight=1e-6, max_grad_norm=1.0):
        self.w = wrapper
        self.opt = torch.optim.AdamW(wrapper.parameters(), lr=lr, betas=betas, weight_decay=wd, eps=eps)
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda s: lr_lambda(s, warmup, max_steps))
        self.kl_weight, self.max_grad_norm = kl_weight, max_grad_norm

    def step(self, pixel_values, eps):
        out = self.w(pixel_values, sample_posterior=True, eps=eps)
        rec, kl, total = losses(out, pixel_values, self.kl_weight)
        total.backward()
        gn = torch.nn.utils.clip_grad_norm_(self.w.parameters(), self.max_grad_norm) \
            if self.max_grad_norm > 0 else torch.tensor(float("nan"))
        lr_used = self.sched.get_last_lr()[0]
        self.opt.step()
        self.sched.step()
        self.opt.zero_grad(set_to_none=True)
        return dict(rec=float(rec.detach()), kl=float(kl.detach()), total=float(total.detach()), grad_norm=float(gn), lr=lr_used)


def count_params(m: nn.Module) -> int:
    return sum(p.numel() for p in m.parameters())
