"""fp32 forward / dgrad / wgrad of the heaviest 3x3 layers: Winograd F(2x2,3x3) / F(3x3,2x2) vs the direct halo-tile kernels
(HIP-event timing; the wgrad time includes the split reduction).
TFLOP/s are ALGORITHMIC (direct-convolution FLOPs / time).  usage: python tools/microbench_wino.py [c128 c256 c512 c512s]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch  # noqa: E402
from vaehip import ops  # noqa: E402

SHAPES = {"c128": (16, 256, 128, 128), "c256": (16, 128, 256, 256), "c512": (16, 64, 512, 512), "c512s": (16, 32, 512, 512)}


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
    dy = torch.randn((B, H, H, Co), device="cuda", generator=g)
    res = torch.randn((B, H, H, Co), device="cuda", generator=g)
    bias = torch.randn(Co, device="cuda", generator=g)
    w = (torch.randn((Co, 3, 3, Ci), device="cuda", generator=g) / math.sqrt(9 * Ci)).permute(0, 3, 1, 2)
    gw, gb = torch.empty((Co, 3, 3, Ci), device="cuda").permute(0, 3, 1, 2), torch.empty(Co, device="cuda")
    st = ops.gn_stats(x, torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda"))
    fl = 2.0 * B * H * H * Ci * Co * 9
    runs = {"fwd": lambda: ops.conv_fwd(x, w, None, "c3"),
            "fwd_gnsilu_res": lambda: ops.conv_fwd(x, w, bias, "c3", xf=ops.XF_AFFINE_SILU, stats=st, res=res),
            "dgrad": lambda: ops.conv_dgrad(dy, w, "c3", (H, H)),
            "wgrad": lambda: ops.conv_wgrad(dy, x, "c3", gw, gb),
            "wgrad_gnsilu": lambda: ops.conv_wgrad(dy, x, "c3", gw, gb, xf=ops.XF_AFFINE_SILU, stats=st)}
    if os.environ.get("MB_ONLY"):
        runs = {k: v for k, v in runs.items() if k.startswith(os.environ["MB_ONLY"])}
    for variant in ("wino", "direct"):
        ops.lib.call("vae_set_option", b"no_wino", 1 if variant == "direct" else 0)
        for k, fn in runs.items():
            ms = timeit(fn)
            print(f"{nm:6s} {variant:7s} {k:15s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s (algorithmic)", flush=True)
    ops.lib.call("vae_set_option", b"no_wino", 0)
