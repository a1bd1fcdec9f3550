"""Tensor-level wrappers over the C ABI.  Activations are NHWC fp32 CUDA tensors.

Every function enqueues on torch's current HIP stream and never synchronises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional, Tuple

import torch

from .lib import ConvGeom, IgemmArgs, WgradArgs, lib

XF_NONE, XF_AFFINE, XF_AFFINE_SILU = 0, 1, 2
PREC_F32, PREC_BF16 = 0, 1
# arithmetic of the conv contractions (set by the engine from training.mixed_precision); tensors stay fp32
PRECISION = PREC_F32


# GroupNorm statistics of a conv output from that conv's epilogue (conv_fwd(gstat_groups=)); False = always re-read the tensor
FUSED_GN_STATS = True

# bf16 image of a parameter arena (base address of the fp32 arena, its size in bytes, base address of the image):
# weights that live inside the arena are handed to the bf16 kernels as `Wh` (set by the engine per forward)
WEIGHTS16: Optional[Tuple[int, int, int]] = None


def pack_bf16(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst (bf16 storage, same numel) = round-to-nearest-even image of the fp32 tensor `src`."""
    assert src.is_contiguous() and dst.is_contiguous() and dst.numel() == src.numel() and dst.element_size() == 2
    lib.call("vae_pack_bf16", _p(src), src.numel(), _p(dst), _stream())
    return dst


def _wh(w: torch.Tensor):
    if PRECISION != PREC_BF16 or WEIGHTS16 is None:
        return None
    base, nbytes, base16 = WEIGHTS16
    off = w.data_ptr() - base
    return C.c_void_p(base16 + off // 2) if 0 <= off < nbytes else None


class precision:
    """context manager: arithmetic of the conv contractions inside the block (PREC_F32 | PREC_BF16)"""

    def __init__(self, prec: int):
        self.prec = prec

    def __enter__(self):
        global PRECISION
        self.prev, PRECISION = PRECISION, self.prec

    def __exit__(self, *exc):
        global PRECISION
        PRECISION = self.prev
        return False


class option:
    """context manager over the library's process-wide kernel-selection switches (vae_set_option: "flat_conv", "no_wino",
    "no_wino4", "no_wide", "no_thin_mfma", "no_wgrad_dma", and the count "wide_reserved_cus"); the environment (VAEHIP_FLAT_CONV / VAEHIP_NO_WINO / ...) only
    gives their initial values"""

    def __init__(self, name: str, value: int = 1):
        self.name, self.value = name.encode(), int(value)

    def __enter__(self):
        self.prev = lib.query("vae_get_option", self.name)
        lib.call("vae_set_option", self.name, self.value)

    def __exit__(self, *exc):
        lib.call("vae_set_option", self.name, self.prev)
        return False


def get_option(name: str) -> int:
    return int(lib.query("vae_get_option", name.encode()))


MODE_FWD, MODE_UP2X, MODE_DGRAD, MODE_DGRAD_S2, MODE_UP2X_DGRAD = 0, 1, 2, 3, 4
GN_GROUPS = 32
GN_EPS = 1e-6


class LaunchProfiler:
    """optional HIP-event timing per kernel instantiation (bench.py's live roofline figures; events go on the launch
    stream).  A record carries two FLOP counts: `flops` = the ALGORITHMIC work of the reference formulation (direct
    convolution: 9 taps, also for an upsampler's phase launches) and `executed` = the multiply-adds the kernel really
    issues on the matrix pipe (16/36 of that for the Winograd kernels, the taps of its mask for a phase convolution).
    HBM-bound helpers (reductions) are recorded with zero FLOPs."""

    def __init__(self):
        self.records = []  # (key, flops, executed, start_event, end_event)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, flops, executed, e0, e1 in self.records:
            d = out.setdefault(key, {"launches": 0, "flops": 0.0, "executed": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["executed"] += executed
            d["ms"] += e0.elapsed_time(e1)
        return out


PROFILER: Optional[LaunchProfiler] = None
WINO_EXECUTED = 16.0 / 36.0  # F(2x2,3x3) / F(3x3,2x2): 16 multiplications per 36 direct multiply-adds
WINO4_EXECUTED = 36.0 / 144.0  # F(4x4,3x3): 36 per 144


def _kernel_name(fn: str, a) -> str:
    buf = C.create_string_buffer(128)
    lib.call(fn, C.byref(a), buf, 128)
    return buf.value.decode()


def _timed(key, flops: float, executed: float, fn: str, *args):
    """lib.call(fn, *args), bracketed by two events on the launch stream when a profiler is installed; `key` may be a
    callable (evaluated only when profiling)"""
    if PROFILER is None:
        lib.call(fn, *args)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.call(fn, *args)
    e1.record()
    PROFILER.records.append((key() if callable(key) else key, flops, executed, e0, e1))


def _launch_igemm(a: IgemmArgs):
    if PROFILER is None:
        lib.call("vae_igemm_rows", C.byref(a), _stream())
        return
    # executed taps per row: the parity-class stride-2 dgrad meets 9/4 taps per row on average (that IS the direct
    # convolution's count), a phase convolution of an upsampler only those of its tap mask (algorithmic: all 9 at high resolution)
    taps = 2.25 if a.g.mode == MODE_DGRAD_S2 else a.g.taps
    ex_taps = bin(a.tapmask).count("1") if a.tapmask else taps
    name = _kernel_name("vae_igemm_kernel_name", a)
    base = 2.0 * a.M * a.N * a.K * a.batch
    if "upwino" in name:  # conv3x3 over the virtual 2x upsample: 36 multiply-adds per low-resolution pixel and channel pair, 9 executed
        lowres = a.M / 4.0 if a.g.mode == MODE_UP2X else float(a.M)
        _timed(name, 2.0 * lowres * a.N * a.K * 36, 2.0 * lowres * a.N * a.K * 9, "vae_igemm_rows", C.byref(a), _stream())
        return
    frac = WINO4_EXECUTED if "wino4" in name else (WINO_EXECUTED if "wino" in name else 1.0)
    _timed(name, base * taps, base * ex_taps * frac, "vae_igemm_rows", C.byref(a), _stream())


def _launch_wgrad(a: WgradArgs):
    if PROFILER is None:
        lib.call("vae_wgrad", C.byref(a), _stream())
        return
    ex_taps = bin(a.tapmask).count("1") if a.tapmask else a.g.taps
    base = 2.0 * a.M * a.N * a.npix * a.batch
    _timed(_kernel_name("vae_wgrad_kernel_name", a), base * a.g.taps, base * ex_taps, "vae_wgrad", C.byref(a), _stream())


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _b16(t: Optional[torch.Tensor]) -> int:
    """1 when the tensor is stored as bf16 (the *_bf16 flags of the C ABI)"""
    return 1 if (t is not None and t.dtype == torch.bfloat16) else 0


def _chk(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise ValueError(f"{name}: expected a float32 CUDA tensor, got {t.dtype} on {t.device}")


def _chk_c(t: torch.Tensor, name: str):
    _chk(t, name)
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous NHWC tensor, strides {t.stride()}")


def ohwi(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d weight (logical OIHW, channels_last memory) -> [O,KH,KW,I] contiguous view; nn.Linear -> [O,1,1,I]."""
    if w.ndim == 2:
        v = w.view(w.shape[0], 1, 1, w.shape[1])
    else:
        v = w.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        raise ValueError("weight memory is not OHWI (channels_last); parameters must live in the VAE arena")
    return v


class Stats(NamedTuple):
    mean: torch.Tensor   # [B, G]
    rstd: torch.Tensor   # [B, G]
    scale: torch.Tensor  # [B, C]
    shift: torch.Tensor  # [B, C]


# ------------------------------------------------------------------ conv geometry
def out_hw(kind: str, H: int, W: int) -> Tuple[int, int]:
    if kind in ("c3", "c1"):
        return H, W
    if kind == "c3s2":
        return (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
    if kind == "c3up":
        return 2 * H, 2 * W
    raise ValueError(kind)


def _fwd_geom(kind: str, B: int, H: int, W: int, Cs: int) -> ConvGeom:
    Ho, Wo = out_hw(kind, H, W)
    if kind == "c3":
        return ConvGeom(B, H, W, Cs, Ho, Wo, 9, 1, 1, 1, MODE_FWD)
    if kind == "c1":
        return ConvGeom(B, H, W, Cs, Ho, Wo, 1, 1, 0, 0, MODE_FWD)
    if kind == "c3s2":
        return ConvGeom(B, H, W, Cs, Ho, Wo, 9, 2, 0, 0, MODE_FWD)
    if kind == "c3up":
        return ConvGeom(B, H, W, Cs, Ho, Wo, 9, 1, 1, 1, MODE_UP2X)
    raise ValueError(kind)


# bf16 mode stores activations as bf16 (round 3): every conv output with at least ACT16_MIN_C channels, the residual stream
# and the gradients of both -- what autocast keeps for the reference (src/train.py:147-154).  GroupNorm statistics, tracker
# sums, the loss, the narrow tensors (image, latents, moments) and the parameters stay fp32.  False = fp32 storage with bf16
# images next to it (round 2's layout; kept for the kernel tests and as the A/B switch).
ACT_BF16 = True
ACT16_MIN_C = 32


def act16() -> bool:
    return PRECISION == PREC_BF16 and ACT_BF16 and WEIGHTS16 is not None


def to_f32(t: torch.Tensor) -> torch.Tensor:
    """fp32 copy of a bf16-stored tensor (the tensor itself when it is fp32)"""
    if t.dtype == torch.float32:
        return t
    out = torch.empty(t.shape, device=t.device, dtype=torch.float32)
    lib.call("vae_unpack_bf16", _p(t.contiguous()), t.numel(), _p(out), _stream())
    return out


def to_bf16(t: torch.Tensor) -> torch.Tensor:
    if t.dtype == torch.bfloat16:
        return t
    return pack_bf16(t.contiguous(), torch.empty(t.shape, device=t.device, dtype=torch.bfloat16))


def _like(t: torch.Tensor, want16: bool) -> torch.Tensor:
    return to_bf16(t) if want16 else to_f32(t)


def act_image_ok(kind: str, x_shape, Co: int, Ci: int) -> bool:
    """bf16 mode: may this layer's forward and wgrad read a bf16 image of the transformed input (gn_apply_bf16)?"""
    if PRECISION != PREC_BF16 or WEIGHTS16 is None or kind != "c3":
        return False
    B, H, W, Cs = x_shape
    g = _fwd_geom(kind, B, H, W, Cs)
    return bool(lib.query("vae_bf16_act_image_ok", C.byref(g), Co, Ci))


# fp32 mode: a Winograd workgroup covers 64 output channels, so GroupNorm+SiLU fused into its halo staging is recomputed
# Cout/64 times per element (and once more per 128 channels in the wgrad).  Writing the transformed tensor once
# (vae_gn_apply: 8 B/element) and running forward + wgrad without a transform is cheaper (tools/microbench_wino.py, 512
# channels @64^2: fused forward 1.38 vs 1.20 ms, fused wgrad 1.43 vs 1.27, the extra pass 0.07 ms).  Measured on the whole step
# (bench.py, 256^2, batch 16): never 214.8 ms / 15.5 GiB peak, Cin >= 512: 209.6 / 17.7, >= 256: 207.2 / 20.7, >= 128 (all
# of them): 205.2 ms / 25.4 GiB.  VAEHIP_ACT32_MIN_CIN overrides (a large value = always fuse: the lowest peak memory).
ACT_IMAGE32_MIN_CIN = int(os.environ.get("VAEHIP_ACT32_MIN_CIN", "128"))


def act_image32_ok(kind: str, x_shape, Co: int, Ci: int) -> bool:
    if PRECISION != PREC_F32 or not WINOGRAD or kind != "c3" or Ci < ACT_IMAGE32_MIN_CIN or get_option("no_wino") or get_option("flat_conv"):
        return False
    B, H, W, Cs = x_shape
    return H % 8 == 0 and W % 16 == 0 and Ci % 32 == 0 and Co % 64 == 0 and Cs == Ci


# bf16 mode with fp32 storage (ACT_BF16 = False) keeps gradients that only feed bf16 kernels as bf16 images (the dgrad outputs
# of the halo-tile kernels and the GroupNorm-backward outputs).  False = fp32 gradients everywhere (the round-1 behaviour).
GRAD_IMAGES = True


def grad_image_ok(kind: str, x_shape, Co: int, Ci: int) -> bool:
    """bf16 mode: may the OUTPUT gradient of this layer (forward input x_shape) be handed to its dgrad and wgrad as a bf16 image?"""
    if PRECISION != PREC_BF16 or WEIGHTS16 is None or kind != "c3" or not GRAD_IMAGES:
        return False
    B, H, W, Cs = x_shape
    g = _fwd_geom(kind, B, H, W, Cs)
    return bool(lib.query("vae_bf16_grad_image_ok", C.byref(g), int(Co), int(Ci)))


def _grad16(dy: torch.Tensor) -> Optional[torch.Tensor]:
    """the bf16 form of a gradient tensor: the tensor itself when it is stored as bf16, or the image its producer attached"""
    if dy.dtype == torch.bfloat16:
        return dy
    return getattr(dy, "_b16", None)


def gn_apply_bf16(x: torch.Tensor, st: "Stats", xf: int) -> torch.Tensor:
    """bf16(XF(x)) as a [B,H,W,C] bfloat16 tensor: the activation image conv_fwd(a16=) / conv_wgrad(x16=) read."""
    B, H, W, Cc = x.shape
    y = torch.empty((B, H, W, Cc), device=x.device, dtype=torch.bfloat16)
    lib.call("vae_gn_apply_bf16", _p(x), _b16(x), _p(st.scale), _p(st.shift), B, H * W, Cc, xf, _p(y), _stream())
    return y


# fp32 mode: forward and dgrad of the plain 3x3 stride-1 layers as Winograd F(2x2,3x3) (16 instead of 36 multiplications per
# 2x2 outputs; csrc/conv3_wino.hip).  False = the direct halo-tile kernels (also what the library option "no_wino" selects)
WINOGRAD = True


def _wino(a: IgemmArgs, dev):
    """transformed weights for the launch `a` describes when the Winograd kernel serves it (a.Wu is set), else None"""
    if not WINOGRAD or PRECISION != PREC_F32 or not lib.query("vae_wino_ok", C.byref(a)):
        return None
    wu = torch.empty((int(lib.query("vae_wino_weight_floats", C.byref(a))),), device=dev, dtype=torch.float32)
    lib.call("vae_wino_weights", C.byref(a), _p(wu), _stream())
    a.Wu = _p(wu)
    return wu


# conv3x3(nearest_upsample_2x(x)) as four phase convolutions on the low-resolution x (2x2 effective kernels: 16 instead
# of 36 tap-MACs per low-resolution pixel, forward and dgrad); False = the virtual-upsample kernel
PHASE_UPCONV = True
_PHASE_MASK = {0: (0, 1), 1: (1, 2)}  # parity -> rows (columns) of the 3x3 window on the low-resolution grid


def _phase_tapmask(pa: int, pb: int) -> int:
    return sum(1 << (kh * 3 + kw) for kh in _PHASE_MASK[pa] for kw in _PHASE_MASK[pb])


def upconv_phase_weights(wv: torch.Tensor) -> torch.Tensor:
    """wv: OHWI memory [Co,3,3,Ci] -> [4,Co,3,3,Ci] effective kernels per output parity (vae_upconv_phase_weights)."""
    Co, kh, kw, Ci = wv.shape
    we = torch.empty((4, Co, 3, 3, Ci), device=wv.device, dtype=torch.float32)
    lib.call("vae_upconv_phase_weights", _p(wv), Co, Ci, _p(we), _stream())
    return we


def _phase_args(x_lo_shape, Co, Ci, dgrad: bool) -> IgemmArgs:
    """argument block of one phase convolution on the low-resolution grid (pointers / views filled by the caller)"""
    B, H, W, _ = x_lo_shape
    a = IgemmArgs()
    if not dgrad:
        a.g = ConvGeom(B, H, W, Ci, H, W, 9, 1, 1, 1, MODE_FWD)
        a.M, a.N, a.K, a.ldc = B * H * W, Co, Ci, Co
        a.sn, a.sk, a.st = 9 * Ci, 1, Ci
    else:
        a.g = ConvGeom(B, H, W, Co, H, W, 9, 1, 1, 1, MODE_DGRAD)
        a.M, a.N, a.K, a.ldc = B * H * W, Ci, Co, Ci
        a.sn, a.sk, a.st = 1, 9 * Ci, Ci
    a.batch, a.sAb, a.sWb, a.sCb = 1, 0, 0, 0
    a.xf, a.alpha, a.prec = XF_NONE, 1.0, PRECISION
    return a


def _phase_weight_ptrs(a: IgemmArgs, we: torch.Tensor, ph: int, we16: Optional[torch.Tensor]):
    a.W = _p(we[ph])
    a.Wh = _p(we16[ph]) if we16 is not None else None


def _phase_weights(wv):
    """effective kernels of the four phases, and their bf16 image in bf16 mode"""
    we = upconv_phase_weights(wv)
    we16 = None
    if PRECISION == PREC_BF16:
        we16 = torch.empty(we.shape, device=we.device, dtype=torch.bfloat16)
        pack_bf16(we, we16)
    return we, we16


def _upconv_wino_fwd(x, wv, bias):
    """fp32: conv3x3(nearest_upsample_2x(x)) with 9 multiplications per low-resolution pixel and channel pair
    (csrc/conv3_upwino.hip) -> [B,2H,2W,Co], or None when the kernel does not serve the layer"""
    if not WINOGRAD or PRECISION != PREC_F32 or x.dtype != torch.float32:
        return None
    B, H, W, Cs = x.shape
    Co, _, _, Ci = wv.shape
    a = IgemmArgs()
    a.g = ConvGeom(B, H, W, Cs, 2 * H, 2 * W, 9, 1, 1, 1, MODE_UP2X)
    a.M, a.N, a.K, a.ldc = B * 4 * H * W, Co, Ci, Co
    a.sn, a.sk, a.st = 9 * Ci, 1, Ci
    a.batch, a.sAb, a.sWb, a.sCb = 1, 0, 0, 0
    a.xf, a.alpha, a.prec = XF_NONE, 1.0, PRECISION
    out = torch.empty((B, 2 * H, 2 * W, Co), device=x.device, dtype=torch.float32)
    a.A, a.W, a.C, a.bias = _p(x), _p(wv), _p(out), _p(bias)
    wu = _wino(a, x.device)
    if wu is None:
        return None
    _launch_igemm(a)
    return out


def _upconv_wino_dgrad(dy, wv, in_hw):
    """fp32: dy [B,2H,2W,Co] -> gradient wrt the low-resolution input [B,H,W,Ci] (3x3 dgrad + 2x2 sum-pool in one pass), or None"""
    if not WINOGRAD or PRECISION != PREC_F32 or dy is None or dy.dtype != torch.float32:
        return None
    B, Hy, Wy, Co = dy.shape
    H, W = in_hw
    _, _, _, Ci = wv.shape
    a = IgemmArgs()
    a.g = ConvGeom(B, Hy, Wy, Co, H, W, 9, 1, 1, 1, MODE_UP2X_DGRAD)
    a.M, a.N, a.K, a.ldc = B * H * W, Ci, Co, Ci
    a.sn, a.sk, a.st = 1, 9 * Ci, Ci
    a.batch, a.sAb, a.sWb, a.sCb = 1, 0, 0, 0
    a.xf, a.alpha, a.prec = XF_NONE, 1.0, PRECISION
    out = torch.empty((B, H, W, Ci), device=dy.device, dtype=torch.float32)
    a.A, a.W, a.C = _p(dy), _p(wv), _p(out)
    wu = _wino(a, dy.device)
    if wu is None:
        return None
    _launch_igemm(a)
    return out


def _upconv_phase_fwd(x, wv, bias, want16: bool):
    """-> [B,2H,2W,Co] (bf16 when want16 and the kernel can write it) or None when the halo-tile kernels do not serve the
    low-resolution geometry.  x: fp32, or bf16 (then it IS the operand image)."""
    B, H, W, Cs = x.shape
    Co, _, _, Ci = wv.shape
    if Cs != Ci:
        return None
    a = _phase_args(x.shape, Co, Ci, False)
    a.A, a.W = _p(x), _p(wv)
    a.bias = _p(bias)
    a.c_step = 2
    a.Wh = _wh(wv)  # (eligibility of the bf16 kernel: any aligned image will do for the query)
    a.tapmask = _phase_tapmask(0, 0)
    xb = x.dtype == torch.bfloat16
    x16 = None
    if PRECISION == PREC_BF16 and a.Wh is not None and Cs % 8 == 0:
        # bf16 image of the low-resolution input (a resnet output, no GroupNorm in front): the wide-tile kernel takes the four
        # phase convolutions as 2x2 tap blocks
        a.A16 = a.A  # placeholder with the right alignment for the query
        a.out_bf16 = 1 if want16 else 0
        ok = bool(lib.query("vae_conv_phase_ok", C.byref(a)))
        if ok and want16 and not lib.query("vae_conv_io16_ok", C.byref(a)):
            a.out_bf16, want16 = 0, False
        if ok:
            x16 = x if xb else pack_bf16(x, torch.empty(x.shape, device=x.device, dtype=torch.bfloat16))
            a.A16 = _p(x16)
            if not xb:
                x._b16 = x16  # the layer's weight gradient reads the same image
        else:
            a.A16, a.out_bf16 = None, 0
    if x16 is None:
        if xb or not lib.query("vae_conv_phase_ok", C.byref(a)):
            return None
        want16 = False
    out = torch.empty((B, 2 * H, 2 * W, Co), device=x.device, dtype=torch.bfloat16 if want16 else torch.float32)
    a.C = _p(out)
    we, we16 = _phase_weights(wv)
    for pa in (0, 1):
        for pb in (0, 1):
            _phase_weight_ptrs(a, we, pa * 2 + pb, we16)
            a.tapmask, a.c_oy, a.c_ox = _phase_tapmask(pa, pb), pa, pb
            _launch_igemm(a)
    return out


def _upconv_phase_dgrad(dy, wv, in_hw, dy16=None):
    """dy [B,2H,2W,Co] (fp32, or None when only the bf16 form dy16 exists) -> gradient wrt the LOW-resolution input
    [B,H,W,Ci], fp32 (the four phases add up through the residual input), or None when not served"""
    src = dy if dy is not None else dy16
    B, Hy, Wy, Co = src.shape
    H, W = in_hw
    _, _, _, Ci = wv.shape
    a = _phase_args((B, H, W, Ci), Co, Ci, True)
    a.A, a.W = _p(src), _p(wv)
    a.a_step = 2
    a.Wh = _wh(wv)
    a.tapmask = _phase_tapmask(0, 0)
    use16 = False
    if PRECISION == PREC_BF16 and a.Wh is not None and Co % 8 == 0:
        a.A16 = a.A
        use16 = bool(lib.query("vae_conv_phase_ok", C.byref(a)))
        if use16:
            if dy16 is None:
                dy16 = pack_bf16(dy, torch.empty(dy.shape, device=dy.device, dtype=torch.bfloat16))
            a.A16 = _p(dy16)
        else:
            a.A16 = None
    if not use16 and (dy is None or not lib.query("vae_conv_phase_ok", C.byref(a))):
        return None
    out = torch.empty((B, H, W, Ci), device=src.device, dtype=torch.float32)
    a.C = _p(out)
    we, we16 = _phase_weights(wv)
    first = True
    for pa in (0, 1):
        for pb in (0, 1):
            _phase_weight_ptrs(a, we, pa * 2 + pb, we16)
            a.tapmask, a.a_oy, a.a_ox = _phase_tapmask(pa, pb), pa, pb
            a.res = None if first else _p(out)  # the four phases add up
            _launch_igemm(a)
            first = False
    return out


def _chk_act(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16)):
        raise ValueError(f"{name}: expected a float32 or bfloat16 CUDA tensor, got {t.dtype} on {t.device}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous NHWC tensor, strides {t.stride()}")


def conv_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], kind: str, *,
             xf: int = XF_NONE, stats: Optional[Stats] = None, res: Optional[torch.Tensor] = None,
             track: Optional[torch.Tensor] = None, a16: Optional[torch.Tensor] = None,
             gstat_groups: Optional[int] = None, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """x [B,H,W,Cs] (Cs >= Cin, extra channels must be zero-weighted i.e. Cin is taken from w), stored as fp32 or bf16.
    a16: bf16 image of XF(x) (then xf / stats are not applied again; x only gives the geometry).
    gstat_groups: the output feeds a GroupNorm with that many groups: where the kernel has a statistics epilogue the
    partial sums are attached to the returned tensor (`_gstat`) and gn_stats() on it skips its pass over the tensor.
    out_dtype: storage of the result; None = bf16 when bf16 mode stores activations as bf16 (act16()) and the layer has at
    least ACT16_MIN_C output channels, else fp32.  A kernel that cannot honour the storage of an operand gets fp32 copies."""
    _chk_act(x, "conv_fwd.x")
    wv = ohwi(w)
    Co, kh, kw, Ci = wv.shape
    want16 = (out_dtype == torch.bfloat16) if out_dtype is not None else (act16() and Co >= ACT16_MIN_C)
    if a16 is not None:
        assert a16.shape == x.shape and a16.dtype == torch.bfloat16 and a16.is_contiguous()
        xf = XF_NONE
    if kind == "c3up" and xf == XF_NONE and res is None and track is None and a16 is None and not want16:
        out = _upconv_wino_fwd(x, wv, bias)
        if out is not None:
            return out
    if kind == "c3up" and PHASE_UPCONV and xf == XF_NONE and res is None and track is None and a16 is None:
        out = _upconv_phase_fwd(x, wv, bias, want16)
        if out is not None:
            return out if (out.dtype == torch.bfloat16) == want16 else _like(out, want16)
    B, H, W, Cs = x.shape
    taps = kh * kw
    assert taps == (1 if kind == "c1" else 9) and Ci <= Cs, (kind, wv.shape, x.shape)
    g = _fwd_geom(kind, B, H, W, Cs)
    if xf != XF_NONE and not lib.query("vae_xf_fusable_rows", C.byref(g), B * g.Ho * g.Wo, Ci):
        x, xf = gn_apply(x, stats, xf), XF_NONE  # tiny spatial size: several batch items per tile
    if res is not None:
        _chk_act(res, "conv_fwd.res")
        assert res.shape == (B, g.Ho, g.Wo, Co)
        res = _like(res, want16)  # the residual is stored like the output
    out = torch.empty((B, g.Ho, g.Wo, Co), device=x.device, dtype=torch.bfloat16 if want16 else torch.float32)
    a = IgemmArgs()
    a.A, a.W, a.C, a.bias, a.res = _p(x), _p(wv), _p(out), _p(bias), _p(res)
    xb = x.dtype == torch.bfloat16
    if a16 is not None:
        a.A16 = _p(a16)
    elif xb and xf == XF_NONE:
        a.A16 = _p(x)   # a bf16 tensor IS its own image; the dispatcher turns it into a_bf16 for the flat kernels
    elif xb:
        a.a_bf16 = 1    # bf16 storage with a transform on load: the flat / <= 4-channel kernels
    a.out_bf16, a.res_bf16 = int(want16), _b16(res)
    if xf != XF_NONE:
        assert stats is not None and stats.scale.shape == (B, Cs)
        a.scale, a.shift = _p(stats.scale), _p(stats.shift)
    a.track = _p(track)
    a.g = g
    a.M, a.N, a.K, a.ldc = B * g.Ho * g.Wo, Co, Ci, Co
    a.sn, a.sk, a.st = taps * Ci, 1, Ci
    a.batch, a.sAb, a.sWb, a.sCb = 1, 0, 0, 0
    a.xf, a.alpha, a.prec, a.Wh = xf, 1.0, PRECISION, _wh(wv)
    if track is not None:
        assert track.numel() >= ((a.M + 127) // 128) * Co
    if (a.A16 or a.a_bf16 or a.out_bf16 or a.res_bf16) and not lib.query("vae_conv_io16_ok", C.byref(a)):
        # the kernel serving this launch takes fp32 storage only (unvectorised shapes, a tracked halo-tile output, ...)
        x32 = to_f32(a16) if a16 is not None else to_f32(x)
        y = conv_fwd(x32, w, bias, kind, xf=xf, stats=stats, res=None if res is None else to_f32(res), track=track,
                     gstat_groups=gstat_groups, out_dtype=torch.float32)
        if not want16:
            return y
        y16 = to_bf16(y)
        if hasattr(y, "_gstat"):
            y16._gstat = y._gstat
        return y16
    wu = _wino(a, x.device)  # (kept alive until the launch is enqueued; the allocator orders its reuse on the stream)
    if gstat_groups and FUSED_GN_STATS:
        a.gstat_groups = int(gstat_groups)
        nch = lib.query("vae_conv_gstat_chunks", C.byref(a))
        if nch > 0:
            ws = torch.empty((B, nch, int(gstat_groups), 2), device=x.device, dtype=torch.float32)
            a.gstat = _p(ws)
            out._gstat = (ws, int(gstat_groups), nch)
    _launch_igemm(a)
    return out


GNB_EPILOGUE = True  # GroupNorm-backward partial sums from the dgrad epilogue where the serving kernel has one (vae_conv_gnb_chunks)


def _gnb_key(x, st, gamma, beta, silu) -> tuple:
    """identity of the GroupNorm(+SiLU) a dgrad epilogue's backward sums were computed for: input, statistics, affine
    parameters, activation flag and shape"""
    return (x.data_ptr(), st.mean.data_ptr(), st.rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bool(silu), tuple(x.shape))


class GnCtx(NamedTuple):
    """the GroupNorm (+SiLU) whose output the convolution read: what its backward needs besides dL/d(output)"""
    x: torch.Tensor       # the GroupNorm input, NHWC, as stored
    st: "Stats"
    gamma: torch.Tensor
    beta: torch.Tensor
    silu: bool
    groups: int


def conv_dgrad(dy: torch.Tensor, w: torch.Tensor, kind: str, in_hw: Tuple[int, int], out_bf16: bool = False,
               out_dtype: Optional[torch.dtype] = None, gnb: Optional[GnCtx] = None) -> torch.Tensor:
    """gradient wrt the conv input (the XF'ed tensor); dy [B,Ho,Wo,Co] -> [B,H,W,Ci].
    gnb: the result is dL/d silu(gn(x)) of this GroupNorm and goes to gn_bwd: where the kernel serving the launch can, its
    epilogue also leaves gn_bwd's first pass (the per-chunk sums over x and the result), attached as `_gnb = (ws, nchunk)`.
    dy: fp32 (optionally with a bf16 image attached, `_b16`) or a bf16 tensor.  Storage of the result: out_dtype, or bf16 when
    act16() and Ci >= ACT16_MIN_C, or when out_bf16 is asked for and the halo-tile kernel serving the layer can write it (fp32
    storage mode: the caller feeds it to gn_bwd only), else fp32."""
    dy16 = _grad16(dy)
    dy32 = dy if dy.dtype == torch.float32 else None
    _chk_act(dy, "conv_dgrad.dy")
    wv = ohwi(w)
    Co, kh, kw, Ci = wv.shape
    taps = kh * kw
    B, Hy, Wy, Cy = dy.shape
    assert Cy == Co
    H, W = in_hw
    forced = out_dtype is not None
    want16 = (out_dtype == torch.bfloat16) if forced else (act16() and Ci >= ACT16_MIN_C)
    use16 = dy16 is not None and (dy32 is None or grad_image_ok(kind, (B, H, W, Ci), Co, Ci))
    if kind == "c3up" and not want16:
        out = _upconv_wino_dgrad(dy32, wv, in_hw)
        if out is not None:
            return out
    if kind == "c3up" and PHASE_UPCONV:
        out = _upconv_phase_dgrad(dy32, wv, in_hw, dy16)
        if out is not None:
            return _like(out, want16)
    if kind == "c3up":
        Hr, Wr, stride, pad = 2 * H, 2 * W, 1, 1
    elif kind == "c3s2":
        Hr, Wr, stride, pad = H, W, 2, 0
    elif kind == "c3":
        Hr, Wr, stride, pad = H, W, 1, 1
    else:
        Hr, Wr, stride, pad = H, W, 1, 0
    mode = MODE_DGRAD
    if kind == "c3s2" and H % 2 == 0 and W % 2 == 0 and (B * H * W // 4) % 128 == 0:
        mode = MODE_DGRAD_S2  # parity-class-major rows: only the taps a class meets are computed (9/4 instead of 9)
    g = ConvGeom(B, Hy, Wy, Co, Hr, Wr, taps, stride, pad, pad, mode)
    a = IgemmArgs()
    src = dy16 if use16 else dy32
    a.A, a.W = _p(dy32 if dy32 is not None else dy16), _p(wv)  # with A16 the kernel reads the image; A only gives the alignment
    a.A16 = _p(dy16) if use16 else None
    a.g = g
    a.M, a.N, a.K, a.ldc = B * Hr * Wr, Ci, Co, Ci
    a.sn, a.sk, a.st = 1, taps * Ci, Ci
    a.batch, a.sAb, a.sWb, a.sCb = 1, 0, 0, 0
    a.xf, a.alpha, a.prec, a.Wh = XF_NONE, 1.0, PRECISION, _wh(wv)
    pool = kind == "c3up"  # (the virtual-upsample fallback: the high-resolution gradient is summed 2x2 in fp32)
    o16 = False
    if not pool and (want16 or (out_bf16 and not forced and PRECISION == PREC_BF16 and GRAD_IMAGES and kind == "c3")):
        a.out_bf16 = 1
        a.C = a.A  # placeholder with the right alignment for the query
        o16 = bool(lib.query("vae_conv_io16_ok", C.byref(a)))
        a.out_bf16 = 1 if o16 else 0
    if a.A16 and not lib.query("vae_conv_io16_ok", C.byref(a)):  # an fp32-only kernel: hand it an fp32 copy of the gradient
        return _like(conv_dgrad(to_f32(dy16), w, kind, in_hw, out_dtype=torch.float32), want16)
    out = torch.empty((B, Hr, Wr, Ci), device=src.device, dtype=torch.bfloat16 if o16 else torch.float32)
    a.C = _p(out)
    wu = _wino(a, src.device)
    # (a result that is re-stored as bf16 below drops its attributes: no sums are computed for it)
    if gnb is not None and GNB_EPILOGUE and not pool and gnb.x.shape == out.shape and not (want16 and out.dtype != torch.bfloat16):
        a.gnb_x, a.gnb_x_bf16 = _p(gnb.x), _b16(gnb.x)
        a.gnb_mean, a.gnb_rstd, a.gnb_gamma, a.gnb_beta = _p(gnb.st.mean), _p(gnb.st.rstd), _p(gnb.gamma), _p(gnb.beta)
        a.gnb_groups, a.gnb_silu = int(gnb.groups), int(gnb.silu)
        nch = lib.query("vae_conv_gnb_chunks", C.byref(a))
        if nch > 0:
            ws = torch.empty((B, nch, Ci, 2), device=src.device, dtype=torch.float32)
            a.gnb_ws = _p(ws)
            # (what the sums belong to: gn_bwd takes them only for exactly this GroupNorm, see _gnb_key)
            out._gnb = (ws, nch, _gnb_key(gnb.x, gnb.st, gnb.gamma, gnb.beta, gnb.silu))
        else:
            a.gnb_x = None
    _launch_igemm(a)
    if pool:
        pooled = torch.empty((B, H, W, Ci), device=src.device, dtype=torch.float32)
        lib.call("vae_sumpool2x2", _p(out), B, H, W, Ci, _p(pooled), _stream())
        out = pooled
    if want16 and out.dtype != torch.bfloat16:
        return to_bf16(out)
    return out


def _upconv_phase_wgrad(dy, x, gv, bgrad_out, dy16=None) -> bool:
    """weight (and bias) gradient of conv3x3(nearest_upsample_2x(x)) from four phase weight gradients on the low-resolution
    grid (each computes the 4 taps of its 2x2 effective kernel), folded back into the 3x3 gradient; False = not served.
    bf16 mode: both operands as bf16 images (x at low resolution, dy at high resolution through the strided view); either
    may be a bf16-stored tensor (dy None: only dy16 exists)."""
    Co, _, _, Ci = gv.shape
    B, H, W, Cs = x.shape
    if Cs != Ci:
        return False
    xb = x.dtype == torch.bfloat16
    a = WgradArgs()
    a.dY, a.X = _p(dy if dy is not None else dy16), _p(x)
    a.g = ConvGeom(B, H, W, Cs, H, W, 9, 1, 1, 1, MODE_FWD)
    a.M, a.N, a.ldy, a.npix, a.nsplit = Co, Ci, Co, B * H * W, 1
    a.batch, a.sYb, a.sXb, a.sOb = 1, 0, 0, 0
    a.xf, a.alpha, a.prec = XF_NONE, 1.0, PRECISION
    a.y_step, a.tapmask = 2, _phase_tapmask(0, 0)
    if not lib.query("vae_wgrad_phase_ok", C.byref(a)):
        return False
    keep = []
    if PRECISION == PREC_BF16 and Cs % 8 == 0 and Co % 8 == 0:
        x16 = x if xb else getattr(x, "_b16", None)
        if x16 is None:
            x16 = pack_bf16(x, torch.empty(x.shape, device=x.device, dtype=torch.bfloat16))
        if dy16 is None:
            dy16 = pack_bf16(dy, torch.empty(dy.shape, device=dy.device, dtype=torch.bfloat16))
            dy._b16 = dy16  # the dgrad that follows reads the same image
        keep = [x16, dy16]
        a.X16, a.dY16 = _p(x16), _p(dy16)
    elif xb or dy is None:
        return False  # the fp32 halo-tile kernel needs fp32 operands
    ns, fus = C.c_int32(0), C.c_int32(0)
    lib.call("vae_wgrad_plan", C.byref(a), C.byref(ns), C.byref(fus))
    ns = ns.value
    a.nsplit = ns
    n = Co * 9 * Ci
    dwe = torch.empty((4, n), device=x.device, dtype=torch.float32)
    dbe = torch.empty((4, Co), device=x.device, dtype=torch.float32) if bgrad_out is not None else None
    partial = torch.empty((ns, n), device=x.device, dtype=torch.float32) if ns > 1 else None
    bpart = torch.empty((ns, Co), device=x.device, dtype=torch.float32) if bgrad_out is not None else None
    for pa in (0, 1):
        for pb in (0, 1):
            ph = pa * 2 + pb
            a.tapmask, a.y_oy, a.y_ox = _phase_tapmask(pa, pb), pa, pb
            if ns == 1:
                a.out = _p(dwe[ph])
            else:
                a.partial = _p(partial)
            a.bias_partial = _p(bpart)
            _launch_wgrad(a)
            if ns > 1 and bpart is not None:
                lib.call("vae_reduce_splits2", _p(partial), ns, n, _p(dwe[ph]), _p(bpart), Co, _p(dbe[ph]), _stream())
            elif ns > 1:
                lib.call("vae_reduce_splits", _p(partial), ns, n, _p(dwe[ph]), _stream())
            elif bpart is not None:
                lib.call("vae_reduce_splits", _p(bpart), ns, Co, _p(dbe[ph]), _stream())
    lib.call("vae_upconv_fold_wgrad", _p(dwe), _p(dbe), Co, Ci, _p(gv), _p(bgrad_out), _stream())
    del keep
    return True


def _wgrad_wino(a: WgradArgs, gv: torch.Tensor, bgrad_out: Optional[torch.Tensor], dev) -> bool:
    """fp32 plain 3x3 stride-1 layers: Winograd F(3x3,2x2) weight gradient (csrc/wgrad3_wino.hip) into a transform-domain slab,
    then the fixed-order reduction + output transform.  False when the layer is not served."""
    if not WINOGRAD or PRECISION != PREC_F32:
        return False
    ns = C.c_int32(0)
    lib.call("vae_wgrad_wino_plan", C.byref(a), C.byref(ns))
    ns = ns.value
    if ns <= 0:
        return False
    npos = int(lib.query("vae_wgrad_wino_positions", C.byref(a)))  # 16, or 9 for an upsampler convolution
    slab = torch.empty((ns, npos * a.N * a.M), device=dev, dtype=torch.float32)
    bpart = torch.empty((ns, a.M), device=dev, dtype=torch.float32) if bgrad_out is not None else None
    a.nsplit, a.partial, a.bias_partial = ns, _p(slab), _p(bpart)
    fl = 2.0 * a.M * a.N * a.npix * 9  # (npix = output pixels: the direct convolution's count in both cases)
    if npos == 9:
        _timed("wgrad3_upwino_kernel", fl, fl * 0.25, "vae_wgrad_wino", C.byref(a), _stream())
    else:
        _timed(f"wgrad3_wino_kernel<{a.xf}>", fl, fl * WINO_EXECUTED, "vae_wgrad_wino", C.byref(a), _stream())
    scratch = torch.empty((npos * a.N * a.M,), device=dev, dtype=torch.float32) if ns > 1 else None
    _timed("wgrad_wino_reduce (split sum + output transform)", 0.0, 0.0, "vae_wgrad_wino_reduce", _p(slab), ns, npos, a.N, a.M,
           _p(scratch), _p(gv), _p(bpart), _p(bgrad_out), _stream())
    return True


def conv_wgrad(dy: torch.Tensor, x: torch.Tensor, kind: str, wgrad_out: torch.Tensor,
               bgrad_out: Optional[torch.Tensor], *, xf: int = XF_NONE, stats: Optional[Stats] = None,
               x16: Optional[torch.Tensor] = None):
    """writes dW into `wgrad_out` (a view with the weight's OHWI memory) and db into `bgrad_out`.
    dy and x: fp32 or bf16 tensors; x16: bf16 image of XF(x) (as conv_fwd's a16)."""
    dy16 = _grad16(dy)
    dy32 = dy if dy.dtype == torch.float32 else None
    _chk_act(dy, "conv_wgrad.dy")
    _chk_act(x, "conv_wgrad.x")
    if x16 is not None:
        assert x16.shape == x.shape and x16.dtype == torch.bfloat16 and x16.is_contiguous()
        xf = XF_NONE
    gv = ohwi(wgrad_out)
    Co, kh, kw, Ci = gv.shape
    taps = kh * kw
    B, H, W, Cs = x.shape
    g = _fwd_geom(kind, B, H, W, Cs)
    assert dy.shape == (B, g.Ho, g.Wo, Co), (dy.shape, (B, g.Ho, g.Wo, Co))
    if (kind == "c3up" and PRECISION == PREC_F32 and WINOGRAD and xf == XF_NONE and x16 is None and dy32 is not None
            and x.dtype == torch.float32):
        a = WgradArgs()  # the 9-position scheme of csrc/wgrad3_upwino.hip (fp32)
        a.dY, a.X, a.g = _p(dy32), _p(x), g
        a.M, a.N, a.ldy, a.npix, a.nsplit = Co, Ci, Co, B * g.Ho * g.Wo, 1
        a.batch, a.sYb, a.sXb, a.sOb = 1, 0, 0, 0
        a.xf, a.alpha, a.prec = XF_NONE, 1.0, PRECISION
        if _wgrad_wino(a, gv, bgrad_out, x.device):
            return
    if kind == "c3up" and PHASE_UPCONV and xf == XF_NONE and x16 is None and _upconv_phase_wgrad(dy32, x, gv, bgrad_out, dy16):
        return
    npix = B * g.Ho * g.Wo
    xb = x.dtype == torch.bfloat16
    use16 = dy16 is not None and (dy32 is None or grad_image_ok(kind, x.shape, Co, Ci))
    a = WgradArgs()
    a.dY, a.X = _p(dy32 if dy32 is not None else dy16), _p(x)
    a.dY16 = _p(dy16) if use16 else None
    a.g = g
    a.M, a.N, a.ldy, a.npix, a.nsplit = Co, Ci, Co, npix, 1
    a.batch, a.sYb, a.sXb, a.sOb = 1, 0, 0, 0
    a.xf, a.alpha, a.prec = xf, 1.0, PRECISION
    if x16 is not None:
        a.X16 = _p(x16)
    elif xb and xf == XF_NONE:
        a.X16 = _p(x)  # a bf16 tensor is its own image (the dispatcher turns it into x_bf16 for the flat kernels)
    elif xb:
        a.x_bf16 = 1
    if xf != XF_NONE:
        assert stats is not None
        a.scale, a.shift = _p(stats.scale), _p(stats.shift)
    if _wgrad_wino(a, gv, bgrad_out, x.device):
        return
    ns, fus = C.c_int32(0), C.c_int32(0)
    lib.call("vae_wgrad_plan", C.byref(a), C.byref(ns), C.byref(fus))
    if xf != XF_NONE and not fus.value:  # tiny spatial size: several batch items per split
        x = gn_apply(x, stats, xf)
        a.X, a.xf, a.scale, a.shift, a.x_bf16 = _p(x), XF_NONE, None, None, 0
        lib.call("vae_wgrad_plan", C.byref(a), C.byref(ns), C.byref(fus))
    if (a.X16 or a.dY16 or a.x_bf16) and not lib.query("vae_wgrad_io16_ok", C.byref(a)):
        # an fp32-only kernel: fp32 copies of the operands (x16 already holds XF(x))
        x32 = to_f32(x16) if x16 is not None else to_f32(x)
        return conv_wgrad(dy32 if dy32 is not None else to_f32(dy16), x32, kind, wgrad_out, bgrad_out, xf=xf, stats=stats)
    ns = ns.value
    a.nsplit = ns
    partial = None
    if ns == 1:
        a.out = _p(gv)
    else:
        partial = torch.empty((ns, Co * taps * Ci), device=x.device, dtype=torch.float32)
        a.partial = _p(partial)
    bpart = None
    if bgrad_out is not None:  # bias gradient = column sums of dY, folded into the wgrad kernel
        bpart = torch.empty((ns, Co), device=x.device, dtype=torch.float32)
        a.bias_partial = _p(bpart)
    _launch_wgrad(a)
    if partial is not None and bpart is not None:  # one launch for both reductions
        lib.call("vae_reduce_splits2", _p(partial), ns, Co * taps * Ci, _p(gv), _p(bpart), Co, _p(bgrad_out), _stream())
    elif partial is not None:
        lib.call("vae_reduce_splits", _p(partial), ns, Co * taps * Ci, _p(gv), _stream())
    elif bpart is not None:
        lib.call("vae_reduce_splits", _p(bpart), ns, Co, _p(bgrad_out), _stream())


# ------------------------------------------------------------------ GroupNorm
_GN_TARGET = 1024  # workgroups per GroupNorm streaming pass (B * nchunk); tools/gn_bench.py: best of 256..4096 on MI355X


def _gn_nchunk(B: int, HW: int, Cc: int) -> int:
    pr = max(1, 256 // (Cc // 4))
    return max(1, min(_GN_TARGET // max(B, 1), HW // (16 * pr) if HW >= 16 * pr else 1))


def gn_stats(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, G: int = GN_GROUPS, eps: float = GN_EPS) -> Stats:
    _chk_act(x, "gn_stats.x")
    B, H, W, Cc = x.shape
    HW = H * W
    dev = x.device
    fused = getattr(x, "_gstat", None)  # partial sums left by the conv epilogue that produced x (conv_fwd(gstat_groups=G))
    if fused is not None and fused[1] == G and FUSED_GN_STATS:
        ws, nch = fused[0], fused[2]
    else:
        nch = _gn_nchunk(B, HW, Cc)
        ws = torch.empty((B, nch, G, 2), device=dev, dtype=torch.float32)
        lib.call("vae_gn_stats_partial", _p(x), _b16(x), B, HW, Cc, G, nch, _p(ws), _stream())
    mean = torch.empty((B, G), device=dev, dtype=torch.float32)
    rstd = torch.empty((B, G), device=dev, dtype=torch.float32)
    scale = torch.empty((B, Cc), device=dev, dtype=torch.float32)
    shift = torch.empty((B, Cc), device=dev, dtype=torch.float32)
    lib.call("vae_gn_stats_final", _p(ws), B, HW, Cc, G, nch, _p(gamma), _p(beta), eps, _p(mean), _p(rstd),
             _p(scale), _p(shift), _stream())
    return Stats(mean, rstd, scale, shift)


def gn_apply(x: torch.Tensor, st: Stats, xf: int) -> torch.Tensor:
    B, H, W, Cc = x.shape
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    lib.call("vae_gn_apply", _p(x), _b16(x), _p(st.scale), _p(st.shift), B, H * W, Cc, xf, _p(y), _stream())
    return y


def gn_track(x: torch.Tensor, st: Stats) -> torch.Tensor:
    """mean over (b,h,w) of |gn(x)| per channel (monitor.py:66), no tensor materialised."""
    B, H, W, Cc = x.shape
    HW = H * W
    nch = _gn_nchunk(B, HW, Cc)
    ws = torch.empty((B * nch, Cc), device=x.device, dtype=torch.float32)
    out = torch.empty((Cc,), device=x.device, dtype=torch.float32)
    lib.call("vae_gn_track_partial", _p(x), _b16(x), _p(st.scale), _p(st.shift), B, HW, Cc, nch, _p(ws), _stream())
    lib.call("vae_track_final", _p(ws), B * nch, Cc, 1.0 / float(B * HW), _p(out), _stream())
    return out


def conv_track_buffer(M: int, Co: int, device) -> torch.Tensor:
    return torch.empty(((M + 127) // 128, Co), device=device, dtype=torch.float32)


def track_final(ws: torch.Tensor, count: int) -> torch.Tensor:
    rows, Cc = ws.shape
    out = torch.empty((Cc,), device=ws.device, dtype=torch.float32)
    lib.call("vae_track_final", _p(ws), rows, Cc, 1.0 / float(count), _p(out), _stream())
    return out


def gn_bwd(x: torch.Tensor, g: torch.Tensor, st: Stats, gamma: torch.Tensor, beta: torch.Tensor, silu: bool,
           add: Optional[torch.Tensor], dgamma: torch.Tensor, dbeta: torch.Tensor, G: int = GN_GROUPS,
           want32: bool = True, want16: bool = False) -> torch.Tensor:
    """x: the GroupNorm input as stored (fp32 or bf16); g: fp32 or bf16; add (the residual-path gradient) is brought to x's
    storage.  Returns dx as fp32 (want32; with the bf16 image attached as `_b16` when want16 too) or as a bf16 tensor alone
    (want16 only: bf16 storage mode, or a gradient that feeds bf16 convolution kernels and nothing else)."""
    _chk_act(x, "gn_bwd.x")
    g16 = g.dtype == torch.bfloat16
    _chk_act(g, "gn_bwd.g")
    assert g.shape == x.shape and (add is None or add.shape == x.shape) and (want32 or want16)
    if add is not None:
        add = _like(add.contiguous(), x.dtype == torch.bfloat16)
    B, H, W, Cc = x.shape
    HW = H * W
    dev = x.device
    coef = torch.empty((B, G, 2), device=dev, dtype=torch.float32)
    dx = torch.empty(x.shape, device=dev, dtype=torch.float32) if want32 else None
    dx16 = torch.empty(x.shape, device=dev, dtype=torch.bfloat16) if want16 else None
    s = _stream()
    fused = getattr(g, "_gnb", None)  # the dgrad that produced g left the first pass's sums (conv_dgrad(gnb=...))
    if fused is not None and fused[2] == _gnb_key(x, st, gamma, beta, silu):
        ws, nch = fused[0], fused[1]
    else:
        nch = _gn_nchunk(B, HW, Cc)
        ws = torch.empty((B, nch, Cc, 2), device=dev, dtype=torch.float32)
        lib.call("vae_gn_bwd_partial", _p(x), _b16(x), _p(g), _p(st.mean), _p(st.rstd), _p(gamma), _p(beta), B, HW, Cc, G, nch,
                 int(silu), int(g16), _p(ws), s)
    lib.call("vae_gn_bwd_final", _p(ws), _p(st.rstd), _p(gamma), B, HW, Cc, G, nch, _p(dgamma), _p(dbeta), _p(coef), s)
    lib.call("vae_gn_bwd_apply", _p(x), _b16(x), _p(g), _p(st.mean), _p(st.rstd), _p(gamma), _p(beta), _p(coef), _p(add), B, HW,
             Cc, G, int(silu), int(g16), _p(dx), _p(dx16), s)
    if dx is None:
        return dx16
    if dx16 is not None:
        dx._b16 = dx16
    return dx


# ------------------------------------------------------------------ batched GEMMs (attention)
def _gemm_rows(A, Bm, out, M, N, K, sn, sk, alpha, z, sAb, sWb, sCb):
    a = IgemmArgs()
    a.A, a.W, a.C = _p(A), _p(Bm), _p(out)
    a.g = ConvGeom(1, 1, M, K, 1, M, 1, 1, 0, 0, MODE_FWD)
    a.M, a.N, a.K, a.ldc = M, N, K, N
    a.sn, a.sk, a.st = sn, sk, 0
    a.batch, a.sAb, a.sWb, a.sCb = z, sAb, sWb, sCb
    a.xf, a.alpha, a.prec = XF_NONE, alpha, PRECISION  # bf16 mode: scores / context / their gradients on the bf16 MFMA too
    _launch_igemm(a)


def gemm_nt(A: torch.Tensor, Bm: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """out[z] = alpha * A[z] @ Bm[z]^T ; A [z,M,K], Bm [z,N,K]."""
    z, M, K = A.shape
    N = Bm.shape[1]
    out = torch.empty((z, M, N), device=A.device, dtype=torch.float32)
    _gemm_rows(A, Bm, out, M, N, K, K, 1, alpha, z, M * K, N * K, M * N)
    return out


def gemm_nn(A: torch.Tensor, Bm: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """out[z] = alpha * A[z] @ Bm[z] ; A [z,M,K], Bm [z,K,N]."""
    z, M, K = A.shape
    N = Bm.shape[2]
    out = torch.empty((z, M, N), device=A.device, dtype=torch.float32)
    _gemm_rows(A, Bm, out, M, N, K, 1, N, alpha, z, M * K, K * N, M * N)
    return out


def gemm_tn(A: torch.Tensor, Bm: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """out[z] = alpha * A[z]^T @ Bm[z] ; A [z,K,M], Bm [z,K,N]."""
    z, K, M = A.shape
    N = Bm.shape[2]
    out = torch.empty((z, M, N), device=A.device, dtype=torch.float32)
    a = WgradArgs()
    a.dY, a.X, a.out = _p(A), _p(Bm), _p(out)
    a.g = ConvGeom(1, 1, K, N, 1, K, 1, 1, 0, 0, MODE_FWD)
    a.M, a.N, a.ldy, a.npix, a.nsplit = M, N, M, K, 1
    a.batch, a.sYb, a.sXb, a.sOb = z, K * M, K * N, M * N
    a.xf, a.alpha, a.prec = XF_NONE, alpha, PRECISION  # bf16 mode: scores / context / their gradients on the bf16 MFMA too
    _launch_wgrad(a)
    return out


# ------------------------------------------------------------------ blockwise attention (no T x T tensor)
# sequences of at least this many tokens run the blockwise kernels (R >= 512: T >= 4096); shorter ones keep the
# materialised scores, which are small there (T = 1024: 4 MB per image).  Tests lower it to compare the two paths.
ATTN_BLOCKWISE_MIN_T = 4096
ATTN_CALLS = {"blockwise_fwd": 0, "blockwise_bwd": 0, "materialised_fwd": 0}


def attn_blockwise_ok(T: int, Cc: int) -> bool:
    return T >= ATTN_BLOCKWISE_MIN_T and bool(lib.query("vae_attn_supported", int(T), int(Cc)))


def _attn_operand(t: torch.Tensor) -> torch.Tensor:
    """operand of the attention kernels in the arithmetic's precision: the fp32 tensor itself, or its bf16 image"""
    if PRECISION != PREC_BF16:
        return t
    return pack_bf16(t, torch.empty(t.shape, device=t.device, dtype=torch.bfloat16))


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float):
    """q, k, v [B,T,C] fp32 -> (o [B,T,C] fp32, saved = operands + row log-sum-exp for attn_bwd)"""
    B, T, Cc = q.shape
    for t in (q, k, v):
        _chk_c(t, "attn_fwd")
    qe, ke, ve = _attn_operand(q), _attn_operand(k), _attn_operand(v)
    o = torch.empty_like(q)
    lse = torch.empty((B, T), device=q.device, dtype=torch.float32)
    fl = 4.0 * B * T * T * Cc  # Q K^T and P V
    _timed(f"attn_fwd_kernel<{'bf16' if PRECISION == PREC_BF16 else 'f32'}>", fl, fl, "vae_attn_fwd", _p(qe), _p(ke), _p(ve), B, T, Cc,
           float(scale), PRECISION, _p(o), _p(lse), _stream())
    ATTN_CALLS["blockwise_fwd"] += 1
    return o, (qe, ke, ve, lse, PRECISION)


def attn_bwd(saved, o: torch.Tensor, do: torch.Tensor, scale: float):
    """-> dq, dk, dv [B,T,C] fp32; P is recomputed blockwise from the saved log-sum-exp"""
    qe, ke, ve, lse, prec = saved
    B, T, Cc = o.shape
    _chk_c(do, "attn_bwd.do")
    with precision(prec):
        doe = _attn_operand(do)
    dq, dk, dv = torch.empty_like(o), torch.empty_like(o), torch.empty_like(o)
    dsum = torch.empty((B, T), device=o.device, dtype=torch.float32)
    # algorithmic: dP, dV, dQ, dK and one recomputation of S = 10 T^2 d; executed: S is recomputed in each of the 3 launches
    _timed(f"attn_bwd_kernels<{'bf16' if prec == PREC_BF16 else 'f32'}>", 10.0 * B * T * T * Cc, 16.0 * B * T * T * Cc, "vae_attn_bwd",
           _p(qe), _p(ke), _p(ve), _p(doe), _p(o), _p(do), _p(lse), B, T, Cc, float(scale), prec,
           _p(dq), _p(dk), _p(dv), _p(dsum), _stream())
    ATTN_CALLS["blockwise_bwd"] += 1
    return dq, dk, dv


def softmax_rows_(S: torch.Tensor):
    cols = S.shape[-1]
    lib.call("vae_softmax_rows", _p(S), S.numel() // cols, cols, _stream())
    return S


def softmax_bwd_rows_(P: torch.Tensor, dP: torch.Tensor):
    cols = P.shape[-1]
    lib.call("vae_softmax_bwd_rows", _p(P), _p(dP), P.numel() // cols, cols, _stream())
    return dP


# ------------------------------------------------------------------ loss / sample / layout / optimizer
def nchw_to_nhwc(x: torch.Tensor, cpad: Optional[int] = None) -> torch.Tensor:
    _chk_c(x, "nchw_to_nhwc")
    B, Cc, H, W = x.shape
    cp = cpad or Cc
    out = torch.empty((B, H, W, cp), device=x.device, dtype=torch.float32)
    lib.call("vae_nchw_to_nhwc", _p(x), B, Cc, H * W, cp, _p(out), _stream())
    return out


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    _chk_c(x, "nhwc_to_nchw")
    B, H, W, Cc = x.shape
    out = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32)
    lib.call("vae_nhwc_to_nchw", _p(x), B, Cc, H * W, _p(out), _stream())
    return out


def sample_kl(moments: torch.Tensor, eps: Optional[torch.Tensor]):
    """moments [B,h,w,2L], eps [B,h,w,L] or None (mode) -> z [B,h,w,L], kl_partial [B,nblk]."""
    _chk_c(moments, "sample_kl.moments")
    B, h, w, L2 = moments.shape
    L = L2 // 2
    nblk = (h * w * L + 255) // 256
    z = torch.empty((B, h, w, L), device=moments.device, dtype=torch.float32)
    klp = torch.empty((B, nblk), device=moments.device, dtype=torch.float32)
    lib.call("vae_sample_kl", _p(moments), _p(eps), B, h * w, L, _p(z), _p(klp), _stream())
    return z, klp


def sample_kl_bwd(moments, eps, dz, kl_weight: float):
    B, h, w, L2 = moments.shape
    L = L2 // 2
    dm = torch.empty_like(moments)
    lib.call("vae_sample_kl_bwd", _p(moments), _p(eps), _p(dz), B, h * w, L, float(kl_weight), _p(dm), _stream())
    return dm


def mse_kl_loss(recon: torch.Tensor, target: torch.Tensor, kl_partial: torch.Tensor, kl_weight: float) -> torch.Tensor:
    """-> device scalars [mse_mean, kl_mean, total] (train.py:289-291)."""
    n = recon.numel()
    nblk = max(1, min(1024, (n + 4095) // 4096))
    ws = torch.empty((nblk,), device=recon.device, dtype=torch.float32)
    sc = torch.empty((3,), device=recon.device, dtype=torch.float32)
    s = _stream()
    lib.call("vae_mse_partial", _p(recon), _p(target), n, _p(ws), nblk, s)
    B, kb = kl_partial.shape
    lib.call("vae_loss_final", _p(ws), nblk, n, _p(kl_partial), B, kb, float(kl_weight), _p(sc), s)
    return sc


def mse_bwd(recon: torch.Tensor, target: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    d = torch.empty_like(recon)
    lib.call("vae_mse_bwd", _p(recon), _p(target), recon.numel(), float(scale), _p(d), _stream())
    return d


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a + b; two bf16 tensors (or a mixed pair, brought to bf16) give a bf16 sum rounded once from the fp32 sum"""
    if a.dtype == torch.bfloat16 or b.dtype == torch.bfloat16:
        a, b = to_bf16(a), to_bf16(b)
        o = torch.empty_like(a)
        lib.call("vae_add_bf16", _p(a), _p(b), a.numel(), _p(o), _stream())
        return o
    o = torch.empty_like(a)
    lib.call("vae_add", _p(a), _p(b), a.numel(), _p(o), _stream())
    return o


def sqnorm(g: torch.Tensor, out: torch.Tensor, ws: Optional[torch.Tensor] = None):
    n = g.numel()
    nblk = 2048
    if ws is None:
        ws = torch.empty((nblk,), device=g.device, dtype=torch.float32)
    lib.call("vae_sqnorm", _p(g), n, _p(ws), nblk, _p(out), _stream())
    return out


def adamw(p, g, m, v, sqn: Optional[torch.Tensor], max_norm: float, lr: float, beta1: float, beta2: float, eps: float,
          wd: float, step: int):
    lib.call("vae_adamw", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(sqn), float(max_norm), float(lr), float(beta1),
             float(beta2), float(eps), float(wd), int(step), _stream())
