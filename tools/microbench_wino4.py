"""fp32 forward / dgrad of the 8 plain 3x3 stride-1 layer shapes of the 256x256 batch-16 step: Winograd F(4x4,3x3) (conv3_wino4.hip)
vs F(2x2,3x3) (conv3_wino.hip, library option "no_wino4").  HIP-event timing; `exe` = executed MFMA work / 157.3 TFLOP/s.
usage: python tools/microbench_wino4.py [shape ...]   (MB_ONLY=fwd|dgrad restricts the runs, MB_ALGO=f4|f2 the algorithms)"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch  # noqa: E402
from vaehip import ops  # noqa: E402

# name: (B, H, Cin, Cout, launches of this shape per step: forward + dgrad)
SHAPES = {"c128": (16, 256, 128, 128), "c256": (16, 128, 256, 256), "c512": (16, 64, 512, 512), "c512s": (16, 32, 512, 512),
          "c512_256": (16, 128, 512, 256), "c256_128": (16, 256, 256, 128), "c128_256": (16, 128, 128, 256), "c256_512": (16, 64, 256, 512)}


def timeit(fn, n=30):
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
    gam, bet = torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    st = ops.gn_stats(x, gam, bet)
    ctx = ops.GnCtx(x, st, gam, bet, True, 32)
    fl = 2.0 * B * H * H * Ci * Co * 9
    runs = {"fwd": lambda: ops.conv_fwd(x, w, None, "c3"),
            "fwd_bias_res_gstat": lambda: ops.conv_fwd(x, w, bias, "c3", res=res, gstat_groups=32),
            "fwd_gnsilu": lambda: ops.conv_fwd(x, w, bias, "c3", xf=ops.XF_AFFINE_SILU, stats=st),
            "dgrad": lambda: ops.conv_dgrad(dy, w, "c3", (H, H)),
            "dgrad_gnb": lambda: ops.conv_dgrad(dy, w, "c3", (H, H), gnb=ctx)}
    if os.environ.get("MB_ONLY"):
        runs = {k: v for k, v in runs.items() if k.startswith(os.environ["MB_ONLY"])}
    for algo in (os.environ.get("MB_ALGO", "f4,f2").split(",")):
        ops.lib.call("vae_set_option", b"no_wino4", 0 if algo == "f4" else 1)
        frac = 0.25 if algo == "f4" else 16.0 / 36.0
        for k, fn in runs.items():
            ms = timeit(fn)
            print(f"{nm:9s} {algo:3s} {k:20s} {ms:8.3f} ms  alg {fl / ms / 1e9:7.1f} TFLOP/s  exe {fl * frac / ms / 1e9 / 157.3:5.3f} of peak", flush=True)
    ops.lib.call("vae_set_option", b"no_wino4", 0)
