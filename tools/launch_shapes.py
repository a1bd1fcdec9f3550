"""Per-launch table of one train step: kernel, GEMM shape, milliseconds, TFLOP/s -- which LAUNCHES an instantiation's total hides
(the bench's kernel table sums per instantiation).  usage: python tools/launch_shapes.py [--dtype bf16] [--batch 32] [--res 256] [--match igemm_rows_bf16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch  # noqa: E402
from vaehip import ops  # noqa: E402
from vaehip.trainer import HipTrainer  # noqa: E402
from models.sdxl_vae_wrapper import SDXLVAEWrapper  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--match", default="", help="only kernels whose name contains this")
ap.add_argument("--min-ms", type=float, default=0.0)
args = ap.parse_args()

_name = ops._kernel_name


def _name_with_shape(fn, a):
    s = _name(fn, a)
    g = a.g
    if hasattr(a, "K"):
        extra = f" M{a.M} N{a.N} K{a.K} taps{g.taps} s{g.stride} mode{g.mode} batch{a.batch}"
    else:  # weight gradient: [M = Cout] x [N = Cin] per tap over npix pixels
        extra = f" Cout{a.M} Cin{a.N} npix{a.npix} taps{g.taps} s{g.stride} mode{g.mode} batch{a.batch} nsplit{a.nsplit}"
    return s + extra


ops._kernel_name = _name_with_shape
dev = torch.device("cuda", 0)
torch.manual_seed(42)
w = SDXLVAEWrapper("synthetic:42", device=dev)
tr = HipTrainer(w, lr=1e-5, max_grad_norm=1.0, kl_weight=1e-6, lr_warmup_steps=100, max_train_steps=10000,
                mixed_precision="bf16" if args.dtype == "bf16" else "no")
g = torch.Generator(device=dev).manual_seed(42)
x = torch.rand((args.batch, 3, args.res, args.res), device=dev, generator=g) * 2 - 1
eps = torch.randn((args.batch, 4, args.res // 8, args.res // 8), device=dev, generator=g)
for _ in range(3):
    tr.train_step(x, eps)
prof = ops.LaunchProfiler()
ops.PROFILER = prof
tr.train_step(x, eps)
ops.PROFILER = None
torch.cuda.synchronize()
tot = 0.0
rows = []
for key, flops, executed, e0, e1 in prof.records:
    ms = e0.elapsed_time(e1)
    tot += ms
    if args.match in key and ms >= args.min_ms:
        rows.append((key, ms, executed / ms / 1e9 if ms > 0 else 0.0))
print(f"{len(prof.records)} timed launches, {tot:.2f} ms in them; listed: {sum(r[1] for r in rows):.2f} ms")
for key, ms, tf in rows:
    print(f"{ms:8.3f} ms {tf:8.1f} TFLOP/s  {key}")
