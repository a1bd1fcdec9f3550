"""Does a weight-gradient kernel (matrix-bound) overlap with the GroupNorm backward + the next dgrad (the chain it does not
depend on) when the two run on different HIP streams?  Times the pair serially on one stream and concurrently on two.
usage: python tools/overlap_bench.py [f32|bf16] [c128 c256 c512]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch  # noqa: E402
from vaehip import ops  # noqa: E402

SHAPES = {"c128": (16, 256, 128), "c256": (16, 128, 256), "c512": (16, 64, 512)}
mode = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("f32", "bf16") else "f32"
names = [a for a in sys.argv[1:] if a in SHAPES] or list(SHAPES)
if mode == "bf16":
    ops.PRECISION = ops.PREC_BF16
side = torch.cuda.Stream()


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for nm in names:
    B, H, Cc = SHAPES[nm]
    g = torch.Generator(device="cuda").manual_seed(0)
    dt = torch.float32 if mode == "f32" else torch.bfloat16
    x = torch.randn((B, H, H, Cc), device="cuda", generator=g).to(dt)
    a_img = torch.randn((B, H, H, Cc), device="cuda", generator=g).to(dt)  # the transformed conv input (wgrad operand)
    dy = torch.randn((B, H, H, Cc), device="cuda", generator=g).to(dt)
    wbuf = torch.randn((Cc, 3, 3, Cc), device="cuda", generator=g) / math.sqrt(9 * Cc)
    w = wbuf.permute(0, 3, 1, 2)
    if mode == "bf16":
        img = torch.empty(wbuf.numel(), device="cuda", dtype=torch.bfloat16)
        ops.pack_bf16(wbuf, img)
        ops.WEIGHTS16 = (wbuf.data_ptr(), wbuf.numel() * 4, img.data_ptr())
    gw, gb = torch.empty((Cc, 3, 3, Cc), device="cuda").permute(0, 3, 1, 2), torch.empty(Cc, device="cuda")
    gamma, beta = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.1
    dg, db = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    st = ops.gn_stats(x, gamma, beta)

    def wgrad():
        ops.conv_wgrad(dy, a_img, "c3", gw, gb)

    def chain():  # what the backward pass does next on the critical path: dgrad of this layer, GroupNorm backward of the one before
        dA = ops.conv_dgrad(dy, w, "c3", (H, H))
        return ops.gn_bwd(x, dA, st, gamma, beta, True, None, dg, db, want32=mode == "f32", want16=mode == "bf16")

    def serial():
        wgrad()
        chain()

    def overlapped():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            wgrad()
        chain()
        torch.cuda.current_stream().wait_stream(side)

    tw, tc, ts, to = timed(wgrad), timed(chain), timed(serial), timed(overlapped)
    print(f"{mode} {nm}: wgrad {tw:.3f} ms, dgrad + gn_bwd {tc:.3f} ms, one stream {ts:.3f} ms, two streams {to:.3f} ms ({100 * (1 - to / ts):.1f} % saved)", flush=True)
