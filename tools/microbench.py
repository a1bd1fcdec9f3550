"""Micro-benchmark of the contraction kernels on the heaviest SDXL-VAE conv shapes (HIP-event timing).
usage: python tools/microbench.py [shape ...]   shapes: c128 (128->128 @256^2), c256 (256->256 @128^2), c512 (512->512 @64^2), c512s (512->512 @32^2)"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch  # noqa: E402
from vaehip import ops  # noqa: E402

SHAPES = {"c128": (16, 256, 128, 128), "c256": (16, 128, 256, 256), "c512": (16, 64, 512, 512), "c512s": (16, 32, 512, 512)}


def timeit(fn, n=5):
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
    only = os.environ.get("MB_ONLY", "")
    if os.environ.get("MB_PREC", "f32") == "bf16":
        ops.PRECISION = ops.PREC_BF16
    for nm in names:
        B, H, Ci, Co = SHAPES[nm]
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn((B, H, H, Ci), device="cuda", generator=g)
        dy = torch.randn((B, H, H, Co), device="cuda", generator=g)
        wbuf = torch.randn((Co, 3, 3, Ci), device="cuda", generator=g) / math.sqrt(9 * Ci)
        w = wbuf.permute(0, 3, 1, 2)
        if os.environ.get("MB_PACK") and hasattr(ops, "pack_bf16"):  # bf16 weight image, as the engine keeps it
            img = torch.empty(wbuf.numel(), device="cuda", dtype=torch.bfloat16)
            ops.pack_bf16(wbuf, img)
            ops.WEIGHTS16 = (wbuf.data_ptr(), wbuf.numel() * 4, img.data_ptr())
        gw = torch.empty_like(wbuf).permute(0, 3, 1, 2)
        gb = torch.empty(Co, device="cuda")
        gamma = torch.ones(Ci, device="cuda")
        beta = torch.zeros(Ci, device="cuda")
        st = ops.gn_stats(x, gamma, beta)
        fl = 2.0 * B * H * H * Ci * Co * 9
        runs = {
            "fwd": lambda: ops.conv_fwd(x, w, None, "c3"),
            "fwd_gnsilu": lambda: ops.conv_fwd(x, w, None, "c3", xf=ops.XF_AFFINE_SILU, stats=st),
            "dgrad": lambda: ops.conv_dgrad(dy, w, "c3", (H, H)),
            "wgrad": lambda: ops.conv_wgrad(dy, x, "c3", gw, gb),
            "wgrad_gnsilu": lambda: ops.conv_wgrad(dy, x, "c3", gw, gb, xf=ops.XF_AFFINE_SILU, stats=st),
        }
        for k, fn in runs.items():
            if only and k != only:
                continue
            ms = timeit(fn)
            print(f"{nm:6s} {k:13s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
