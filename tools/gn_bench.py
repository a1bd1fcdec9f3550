"""GroupNorm pass timings at the config-2 shapes for different chunk counts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch
from vaehip import ops

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (B, H, C) in [(16, 256, 128), (16, 128, 256), (16, 64, 512), (16, 32, 512)]:
    x = torch.randn(B, H, H, C, device="cuda"); g = torch.randn_like(x)
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
    for target in (4096, 2048, 1024, 512, 256):
        ops._GN_TARGET = target
        st = ops.gn_stats(x, gamma, beta)
        t1 = timeit(lambda: ops.gn_stats(x, gamma, beta))
        t2 = timeit(lambda: ops.gn_bwd(x, g, st, gamma, beta, True, None, dg, db))
        gb = x.numel() * 4 / 1e9
        print(f"B{B} H{H} C{C} target {target:5d} nchunk {ops._gn_nchunk(B, H*H, C):4d}: stats {t1:7.1f} us ({gb/t1*1e6:6.0f} GB/s)  bwd {t2:7.1f} us ({gb*5/t2*1e6:6.0f} GB/s)", flush=True)
