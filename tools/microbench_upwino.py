"""fp32 forward / dgrad / wgrad of the three upsampler convolutions conv3x3(nearest_upsample_2x(x)) at BASELINE configs[1]:
the 9-position Winograd-type kernels (csrc/conv3_upwino.hip, wgrad3_upwino.hip) vs the four phase convolutions on the direct
halo-tile kernels (library option "no_wino").  HIP-event timing; the wgrad time includes its reductions.
TFLOP/s are ALGORITHMIC (direct-convolution FLOPs at high resolution / time).  usage: python tools/microbench_upwino.py [u512 u256 u512s]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch  # noqa: E402
from vaehip import ops  # noqa: E402

SHAPES = {"u512": (16, 64, 512, 512), "u256": (16, 128, 256, 256), "u512s": (16, 32, 512, 512)}  # B, low-resolution side, Cin, Cout


def timeit(fn, n=40):  # (enough launches for the clock to settle)
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for nm in (sys.argv[1:] or list(SHAPES)):
    B, H, Ci, Co = SHAPES[nm]
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((B, H, H, Ci), device="cuda", generator=g)
    dy = torch.randn((B, 2 * H, 2 * H, Co), device="cuda", generator=g)
    bias = torch.randn(Co, device="cuda", generator=g)
    w = (torch.randn((Co, 3, 3, Ci), device="cuda", generator=g) / math.sqrt(9 * Ci)).permute(0, 3, 1, 2)
    gw, gb = torch.empty((Co, 3, 3, Ci), device="cuda").permute(0, 3, 1, 2), torch.empty(Co, device="cuda")
    fl = 2.0 * B * 4 * H * H * Ci * Co * 9
    runs = {"fwd": lambda: ops.conv_fwd(x, w, bias, "c3up"),
            "dgrad": lambda: ops.conv_dgrad(dy, w, "c3up", (H, H)),
            "wgrad": lambda: ops.conv_wgrad(dy, x, "c3up", gw, gb)}
    for variant in ("wino9", "phase"):
        ops.lib.call("vae_set_option", b"no_wino", 1 if variant == "phase" else 0)
        for k, fn in runs.items():
            ms = timeit(fn)
            print(f"{nm:6s} {variant:6s} {k:6s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s (algorithmic)", flush=True)
    ops.lib.call("vae_set_option", b"no_wino", 0)
