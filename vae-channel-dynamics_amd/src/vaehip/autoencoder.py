"""SDXL-VAE module tree (diffusers `AutoencoderKL` names) whose parameters live in one flat
HBM arena and whose math runs on the HIP engine.

This is the model boundary the reference reaches through
`AutoencoderKL.from_pretrained(...)` (reference src/models/sdxl_vae_wrapper.py:27-34);
the tree mirrors the diffusers names the reference's YAMLs and plugins address
(e.g. `encoder.down_blocks.0.resnets.0.norm1`, configs/experiment_imagenette_baseline.yaml:56-59;
`decoder.up_blocks.2.resnets.0.conv_shortcut.weight`, configs/experiment_cifar10_test.yaml:117).

Leaf modules subclass nn.Conv2d / nn.GroupNorm / nn.Linear so `isinstance` checks of
RegionClassifier (classifier.py:56) and DeadNeuronTracker (deadneuron.py:62) keep working, and
all parameters are ordinary leaf nn.Parameters that third parties may mutate in place
(nudger.py:140) -- the kernels read the live arena every step.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

SDXL_VAE_CONFIG = {
    "_class_name": "AutoencoderKL",
    "act_fn": "silu",
    "block_out_channels": [128, 256, 512, 512],
    "down_block_types": ["DownEncoderBlock2D"] * 4,
    "up_block_types": ["UpDecoderBlock2D"] * 4,
    "in_channels": 3,
    "out_channels": 3,
    "latent_channels": 4,
    "layers_per_block": 2,
    "norm_num_groups": 32,
    "sample_size": 1024,
    "scaling_factor": 0.13025,
    "force_upcast": True,
}
GN_EPS = 1e-6
WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"


class _Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e


def _engine_of(m: nn.Module):
    eng = getattr(m, "_vae_engine", None)
    if eng is None:
        raise RuntimeError(f"{type(m).__name__} is not attached to a VAE engine")
    return eng()


# ----------------------------------------------------------------------------- leaves
class HipConv2d(nn.Conv2d):
    """kind: c3 (3x3 s1 p1), c1 (1x1), c3s2 (pad (0,1,0,1) + 3x3 s2), c3up (nearest 2x + 3x3 s1 p1)."""

    def __init__(self, cin, cout, kind: str, **kw):
        k = 1 if kind == "c1" else 3
        stride = 2 if kind == "c3s2" else 1
        pad = 0 if kind in ("c1", "c3s2") else 1
        super().__init__(cin, cout, k, stride, pad, **kw)
        self.kind = kind

    def forward(self, x):  # stand-alone inference call through the same kernel
        return _engine_of(self).leaf_conv(self, x)


class HipGroupNorm(nn.GroupNorm):
    def forward(self, x):
        return _engine_of(self).leaf_groupnorm(self, x)


class HipLinear(nn.Linear):
    def forward(self, x):
        return _engine_of(self).leaf_linear(self, x)


class HipSiLU(nn.SiLU):
    def forward(self, x):
        raise RuntimeError("SiLU is fused into the consuming convolution; call the parent block instead")


# ----------------------------------------------------------------------------- blocks
class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.norm1 = HipGroupNorm(32, cin, eps=GN_EPS, affine=True, **kw)
        self.conv1 = HipConv2d(cin, cout, "c3", **kw)
        self.norm2 = HipGroupNorm(32, cout, eps=GN_EPS, affine=True, **kw)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = HipConv2d(cout, cout, "c3", **kw)
        self.nonlinearity = HipSiLU()
        self.conv_shortcut = HipConv2d(cin, cout, "c1", **kw) if cin != cout else None

    def forward(self, x):
        return _engine_of(self).block_call(self, x)


class Downsample2D(nn.Module):
    def __init__(self, c, **kw):
        super().__init__()
        self.conv = HipConv2d(c, c, "c3s2", **kw)

    def forward(self, x):
        return _engine_of(self).block_call(self, x)


class Upsample2D(nn.Module):
    def __init__(self, c, **kw):
        super().__init__()
        self.conv = HipConv2d(c, c, "c3up", **kw)

    def forward(self, x):
        return _engine_of(self).block_call(self, x)


class Attention(nn.Module):
    def __init__(self, c, **kw):
        super().__init__()
        self.group_norm = HipGroupNorm(32, c, eps=GN_EPS, affine=True, **kw)
        self.to_q = HipLinear(c, c, **kw)
        self.to_k = HipLinear(c, c, **kw)
        self.to_v = HipLinear(c, c, **kw)
        self.to_out = nn.ModuleList([HipLinear(c, c, **kw), nn.Dropout(0.0)])

    def forward(self, x):
        return _engine_of(self).block_call(self, x)


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n, add_down, **kw):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, **kw) for i in range(n)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, **kw)]) if add_down else None

    def forward(self, x):
        return _engine_of(self).block_call(self, x)


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n, add_up, **kw):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, **kw) for i in range(n)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout, **kw)]) if add_up else None

    def forward(self, x):
        return _engine_of(self).block_call(self, x)


class UNetMidBlock2D(nn.Module):
    def __init__(self, c, **kw):
        super().__init__()
        # execution order: resnets[0], attentions[0], resnets[1]
        self.attentions = nn.ModuleList([Attention(c, **kw)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, **kw), ResnetBlock2D(c, c, **kw)])

    def forward(self, x):
        return _engine_of(self).block_call(self, x)


class Encoder(nn.Module):
    def __init__(self, cfg, **kw):
        super().__init__()
        boc = cfg["block_out_channels"]
        self.conv_in = HipConv2d(cfg["in_channels"], boc[0], "c3", **kw)
        self.down_blocks = nn.ModuleList()
        c = boc[0]
        for i, co in enumerate(boc):
            self.down_blocks.append(DownEncoderBlock2D(c, co, cfg["layers_per_block"], i != len(boc) - 1, **kw))
            c = co
        self.mid_block = UNetMidBlock2D(c, **kw)
        self.conv_norm_out = HipGroupNorm(32, c, eps=GN_EPS, **kw)
        self.conv_act = HipSiLU()
        self.conv_out = HipConv2d(c, 2 * cfg["latent_channels"], "c3", **kw)

    def forward(self, x):
        return _engine_of(self).block_call(self, x)


class Decoder(nn.Module):
    def __init__(self, cfg, **kw):
        super().__init__()
        rev = list(reversed(cfg["block_out_channels"]))
        # registered in execution order so the parameter arena is backward-monotone (DP buckets)
        self.conv_in = HipConv2d(cfg["latent_channels"], rev[0], "c3", **kw)
        self.mid_block = UNetMidBlock2D(rev[0], **kw)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(UpDecoderBlock2D(c, co, cfg["layers_per_block"] + 1, i != len(rev) - 1, **kw))
            c = co
        self.conv_norm_out = HipGroupNorm(32, c, eps=GN_EPS, **kw)
        self.conv_act = HipSiLU()
        self.conv_out = HipConv2d(c, cfg["out_channels"], "c3", **kw)

    def forward(self, x):
        return _engine_of(self).block_call(self, x)


# ----------------------------------------------------------------------------- arena
class ParamArena:
    """All parameters in one fp32 buffer (and one grad buffer), each parameter a view of it.

    Conv weights keep the logical OIHW shape with OHWI (channels_last) memory so the kernels'
    implicit-GEMM weight tiles are contiguous along the contraction dimension.  Segment starts
    are aligned to 32 B (float4 loads; 16-B loads from the bf16 image of the arena).  The fused AdamW / grad-norm / all-reduce run over the
    flat buffers; gaps stay zero."""

    def __init__(self, module: nn.Module, device: torch.device):
        self.entries: List[Tuple[str, nn.Parameter, int, int]] = []
        self.device = torch.device(device)
        owners = []
        off = 0
        for mname, mod in module.named_modules():
            for pname, p in mod._parameters.items():
                if p is None:
                    continue
                owners.append((mod, pname, (mname + "." if mname else "") + pname, p, off))
                off += (p.numel() + 7) // 8 * 8  # 32-B segments: the bf16 image of a segment (engine, bf16 mode) is 16-B aligned too
        self.total = off
        self.flat = torch.zeros(off, device=device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=device, dtype=torch.float32)
        self.offset_of: Dict[int, int] = {}
        for mod, pname, name, p, o in owners:
            view = self.view_of(self.flat, p, o)
            if p.device.type == "meta":
                # first materialisation: the meta placeholder is replaced by a real leaf Parameter
                p = nn.Parameter(view, requires_grad=p.requires_grad)
                mod._parameters[pname] = p
            else:
                # re-homing (e.g. after .to(device)): keep the Parameter object (optimizers hold it)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
                p.grad = None
            self.entries.append((name, p, o, p.numel()))
            self.offset_of[id(p)] = o

    @staticmethod
    def view_of(flat: torch.Tensor, p: torch.Tensor, off: int) -> torch.Tensor:
        seg = flat[off: off + p.numel()]
        if p.ndim == 4:
            o, i, kh, kw = p.shape
            return seg.view(o, kh, kw, i).permute(0, 3, 1, 2)
        return seg.view(p.shape)

    def grad_view(self, p: nn.Parameter, target: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.view_of(self.grad if target is None else target, p, self.offset_of[id(p)])

    def attach_grads(self):
        for _, p, o, _n in self.entries:
            p.grad = self.view_of(self.grad, p, o)

    def owns(self, module: nn.Module) -> bool:
        base = self.flat.data_ptr()
        for _, p, o, _n in self.entries:
            if p.device != self.flat.device or p.data_ptr() != base + 4 * o:
                return False
        return True

    def block_low_offset(self, module: nn.Module) -> int:
        offs = [self.offset_of[id(p)] for p in module.parameters()]
        return min(offs) if offs else self.total


# ----------------------------------------------------------------------------- top level
class DiagonalGaussianDistribution:
    """posterior of the encoder (reference call sites sdxl_vae_wrapper.py:64,66; train.py:78,290).
    Operates on NCHW-logical moments; tiny tensors, plain torch ops on the device."""

    def __init__(self, moments: torch.Tensor, generator: Optional[torch.Generator] = None):
        self.parameters = moments
        self.mean, logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        self._generator = generator
        self._eps: Optional[torch.Tensor] = None

    def sample(self, eps: Optional[torch.Tensor] = None) -> torch.Tensor:
        if eps is None:
            eps = self._eps
        if eps is None:
            eps = torch.randn(self.mean.shape, device=self.mean.device, dtype=self.mean.dtype, generator=self._generator)
        return self.mean + self.std * eps

    def mode(self) -> torch.Tensor:
        return self.mean

    def kl(self) -> torch.Tensor:
        return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])


class _EncOut:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class _DecOut:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKLHip(nn.Module):
    def __init__(self, config: Optional[dict] = None, device: Optional[torch.device] = None):
        super().__init__()
        cfg = dict(SDXL_VAE_CONFIG)
        if config:
            cfg.update(config)
        self.config = _Config(cfg)
        kw = {"device": "meta"}
        # registration order == execution order (encoder, quant, post_quant, decoder)
        self.encoder = Encoder(cfg, **kw)
        lc = cfg["latent_channels"]
        self.quant_conv = HipConv2d(2 * lc, 2 * lc, "c1", **kw)
        self.post_quant_conv = HipConv2d(lc, lc, "c1", **kw)
        self.decoder = Decoder(cfg, **kw)
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self.arena = ParamArena(self, dev)
        from .engine import Engine  # late import: engine needs the classes above
        self.engine = Engine(self)
        import weakref
        ref = weakref.ref(self.engine)
        for m in self.modules():
            object.__setattr__(m, "_vae_engine", ref)

    # -- nn.Module plumbing ------------------------------------------------
    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        p0 = next(self.parameters())
        if p0.dtype != torch.float32:
            raise RuntimeError("AutoencoderKLHip keeps fp32 master weights; reduced precision is a compute mode, "
                               "not a parameter dtype (see SURVEY 3.4: bf16 parameters are a reference quirk)")
        if not self.arena.owns(self):
            self.arena = ParamArena(self, p0.device)
        return self

    @property
    def device(self):
        return self.arena.flat.device

    @property
    def dtype(self):
        return torch.float32

    # -- API used by the reference wrapper -----------------------------------
    def encode(self, x: torch.Tensor) -> _EncOut:
        moments = self.engine.encode_autograd(x)
        return _EncOut(DiagonalGaussianDistribution(moments))

    def decode(self, z: torch.Tensor) -> _DecOut:
        return _DecOut(self.engine.decode_autograd(z))

    def forward(self, sample: torch.Tensor, sample_posterior: bool = False):
        d = self.encode(sample).latent_dist
        z = d.sample() if sample_posterior else d.mode()
        return self.decode(z)

    # -- weights ------------------------------------------------------------
    @torch.no_grad()
    def init_synthetic(self, seed: int = 42):
        """deterministic random init (no pretrained weights offline): U(-1/sqrt(fan_in), 1/sqrt(fan_in)),
        GroupNorm gamma = 1 + small jitter, beta small (so per-channel statistics are not degenerate)."""
        gen = torch.Generator().manual_seed(int(seed))
        params = dict(self.named_parameters())
        for name, p in params.items():
            if p.ndim == 1 and name.endswith("weight"):
                v = 1.0 + 0.25 * (torch.rand(p.shape, generator=gen) * 2 - 1)
            elif p.ndim == 1 and "norm" in name.split(".")[-2]:
                v = 0.1 * (torch.rand(p.shape, generator=gen) * 2 - 1)
            else:
                w = p if p.ndim > 1 else params[name[: -len("bias")] + "weight"]
                bound = 1.0 / math.sqrt(w[0].numel())
                v = bound * (torch.rand(p.shape, generator=gen) * 2 - 1)
            p.copy_(v.to(p.device))
        return self

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = _convert_legacy_attention_keys(dict(state_dict))
        own = dict(self.named_parameters())
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"state_dict mismatch: missing {missing[:5]} unexpected {unexpected[:5]}")
        with torch.no_grad():
            for k, p in own.items():
                if k in sd:
                    t = sd[k]
                    if tuple(t.shape) != tuple(p.shape):
                        t = t.reshape(p.shape)  # legacy attention weights are [C,C,1,1]
                    p.copy_(t.to(device=p.device, dtype=torch.float32))
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def save_pretrained(self, save_directory: str):
        """config.json + diffusion_pytorch_model.safetensors with diffusers key names (train.py:412)."""
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        cfg = dict(self.config)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        save_file(sd, os.path.join(save_directory, WEIGHTS_NAME), metadata={"format": "pt"})

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, device=None) -> "AutoencoderKLHip":
        """`path`: a local directory written by save_pretrained / diffusers (config.json + safetensors),
        or `synthetic[:seed]` for deterministic random weights.  Hub names need network access, which
        this build does not assume: they raise, like the reference does when loading fails
        (sdxl_vae_wrapper.py:38-40)."""
        if torch_dtype not in (None, torch.float32, torch.bfloat16, torch.float16):
            raise ValueError(f"unsupported torch_dtype {torch_dtype}")
        if path.startswith("synthetic"):
            seed = int(path.split(":", 1)[1]) if ":" in path else 42
            return cls(device=device).init_synthetic(seed)
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"'{path}' is not a local directory. Download the VAE once (config.json + {WEIGHTS_NAME}) and point "
                f"model.pretrained_vae_name at it, or use 'synthetic:<seed>'.")
        with open(os.path.join(path, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if k in SDXL_VAE_CONFIG}
        from safetensors.torch import load_file
        wpath = os.path.join(path, WEIGHTS_NAME)
        if not os.path.exists(wpath):
            raise FileNotFoundError(wpath)
        m = cls(cfg, device=device)
        m.load_state_dict(load_file(wpath))
        return m


_LEGACY = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def _convert_legacy_attention_keys(sd: dict) -> dict:
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        if "attentions" in parts and len(parts) >= 2 and parts[-2] in _LEGACY:
            parts[-2] = _LEGACY[parts[-2]]
            k = ".".join(parts)
        out[k] = v
    return out
