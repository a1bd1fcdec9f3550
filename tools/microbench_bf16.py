"""Micro-benchmark of the bf16-image contraction kernels (wide tile vs 128-pixel tile, wgrad) on the heaviest SDXL-VAE conv
shapes, HIP-event timing.  usage: python tools/microbench_bf16.py [c128 c256 c512 c512s]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch  # noqa: E402
from vaehip import ops  # noqa: E402

SHAPES = {"c128": (16, 256, 128, 128), "c256": (16, 128, 256, 256), "c512": (16, 64, 512, 512), "c512s": (16, 32, 512, 512)}


def timeit(fn, n=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    names = sys.argv[1:] or list(SHAPES)
    ops.PRECISION = ops.PREC_BF16
    for nm in names:
        B, H, Ci, Co = SHAPES[nm]
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn((B, H, H, Ci), device="cuda", generator=g)
        dy = torch.randn((B, H, H, Co), device="cuda", generator=g)
        res = torch.randn((B, H, H, Co), device="cuda", generator=g)
        bias = torch.randn(Co, device="cuda", generator=g)
        wbuf = torch.randn((Co, 3, 3, Ci), device="cuda", generator=g) / math.sqrt(9 * Ci)
        w = wbuf.permute(0, 3, 1, 2)
        img = torch.empty(wbuf.numel(), device="cuda", dtype=torch.bfloat16)
        ops.pack_bf16(wbuf, img)
        ops.WEIGHTS16 = (wbuf.data_ptr(), wbuf.numel() * 4, img.data_ptr())
        gw = torch.empty_like(wbuf).permute(0, 3, 1, 2)
        gb = torch.empty(Co, device="cuda")
        x16 = x.bfloat16()  # round 3: activations and gradients are stored as bf16 (ops.ACT_BF16)
        st = ops.gn_stats(x16, torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda"))
        a16 = ops.gn_apply_bf16(x16, st, ops.XF_AFFINE_SILU)
        dy16 = dy.bfloat16()
        res16 = res.bfloat16()
        fl = 2.0 * B * H * H * Ci * Co * 9
        runs = {
            "fwd_img": lambda: ops.conv_fwd(x16, w, None, "c3", xf=ops.XF_AFFINE_SILU, stats=st, a16=a16),
            "fwd_img_bias_gstat": lambda: ops.conv_fwd(x16, w, bias, "c3", xf=ops.XF_AFFINE_SILU, stats=st, a16=a16, gstat_groups=32),
            "fwd_img_full": lambda: ops.conv_fwd(x16, w, bias, "c3", xf=ops.XF_AFFINE_SILU, stats=st, a16=a16, res=res16, gstat_groups=32),
            "dgrad_img": lambda: ops.conv_dgrad(dy16, w, "c3", (H, H)),
            "wgrad_img": lambda: ops.conv_wgrad(dy16, x16, "c3", gw, gb, x16=a16),
        }
        for variant in ("wide", "tile128"):
            ops.lib.call("vae_set_option", b"no_wide", 1 if variant == "tile128" else 0)
            for k, fn in runs.items():
                if variant == "tile128" and k.startswith("wgrad"):
                    continue
                ms = timeit(fn)
                print(f"{nm:6s} {variant:8s} {k:20s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
        ops.lib.call("vae_set_option", b"no_wide", 0)


if __name__ == "__main__":
    main()
