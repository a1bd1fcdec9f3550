#!/usr/bin/env python
"""GPU: error of the three fp32 3x3 paths (direct halo-tile, Winograd F(2x2,3x3), Winograd F(4x4,3x3)) against a float64 CPU
convolution on the same operands: forward and dgrad, random-normal test data and post-SiLU-like data.  Prints one JSON line per
case; tools/wino_f4_error_study.py is the CPU emulation the figures are compared with."""
import json
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
from vaehip import ops  # noqa: E402


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


def ohwi(w):
    return w.cuda().contiguous(memory_format=torch.channels_last)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def relmax(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def main():
    cases = [(2, 32, 32, 128, 128, "normal"), (2, 32, 32, 128, 128, "silu"), (1, 64, 64, 256, 256, "silu"), (2, 16, 32, 512, 512, "silu"),
             (2, 32, 32, 128, 128, "ones")]
    for B, H, W, Ci, Co, kind in cases:
        gen = torch.Generator().manual_seed(41 + Ci + Co + H)
        x = torch.randn(B, Ci, H, W, generator=gen) * 1.3 + 0.2
        if kind == "silu":
            x = F.silu(x)
        if kind == "ones":
            x = torch.ones_like(x)
        w = torch.randn(Co, Ci, 3, 3, generator=gen) / math.sqrt(9 * Ci)
        dy = torch.randn(B, Co, H, W, generator=gen)
        x64 = x.double().requires_grad_(True)
        y64 = F.conv2d(x64, w.double(), None, 1, 1)
        (gx64,) = torch.autograd.grad(y64, x64, dy.double())
        xd, wd, dyd = nhwc(x), ohwi(w), nhwc(dy)
        out = {"case": [B, H, W, Ci, Co, kind]}
        for name, opts in (("direct", {"no_wino": 1}), ("f2", {"no_wino4": 1}), ("f4", {})):
            ctx = [ops.option(k, v) for k, v in opts.items()]
            for c in ctx:
                c.__enter__()
            try:
                y = ops.conv_fwd(xd, wd, None, "c3")
                dx = ops.conv_dgrad(dyd, wd, "c3", (H, W))
            finally:
                for c in ctx:
                    c.__exit__(None, None, None)
            out[name] = {"y_rms": rel(nchw(y), y64.detach()), "y_max": relmax(nchw(y), y64.detach()),
                         "dx_rms": rel(nchw(dx), gx64), "dx_max": relmax(nchw(dx), gx64)}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
