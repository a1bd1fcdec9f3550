"""The streaming GroupNorm passes at the model's shapes, fp32 and bf16 storage: microseconds and achieved GB/s of algorithmic bytes
(gn_apply: read + write; gn_bwd = partial + final + apply: x and dy read twice, dx written).
usage: python tools/gn_stream_bench.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch  # noqa: E402
from vaehip import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for (H, C) in [(256, 128), (128, 256), (64, 512), (32, 512)]:
    x32 = torch.randn(B, H, H, C, device="cuda")
    g32 = torch.randn_like(x32)
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    for store in ("fp32", "bf16"):
        x = x32 if store == "fp32" else x32.bfloat16()
        g = g32 if store == "fp32" else g32.bfloat16()
        es = 4 if store == "fp32" else 2
        st = ops.gn_stats(x, gamma, beta)
        n = x.numel()
        t_apply = timeit(lambda: ops.gn_apply(x, st, ops.XF_AFFINE_SILU))
        t_img = timeit(lambda: ops.gn_apply_bf16(x, st, ops.XF_AFFINE_SILU))
        t_bwd = timeit(lambda: ops.gn_bwd(x, g, st, gamma, beta, True, None, dg, db, want32=store == "fp32", want16=store == "bf16"))
        t_bwda = timeit(lambda: ops.gn_bwd(x, g, st, gamma, beta, True, x, dg, db, want32=store == "fp32", want16=store == "bf16"))
        print(f"B{B} {H}x{H} C{C} {store}: apply->fp32 {t_apply:7.1f} us {n*(es+4)/t_apply/1e3:6.0f} GB/s | apply->bf16 {t_img:7.1f} us {n*(es+2)/t_img/1e3:6.0f} GB/s | "
              f"bwd {t_bwd:7.1f} us {n*5*es/t_bwd/1e3:6.0f} GB/s | bwd+add {t_bwda:7.1f} us {n*6*es/t_bwda/1e3:6.0f} GB/s", flush=True)
