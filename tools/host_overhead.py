"""How long does the host take to ENQUEUE one train step (Python + ctypes + allocator), vs the GPU time?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch
from models.sdxl_vae_wrapper import SDXLVAEWrapper
from vaehip.trainer import HipTrainer

for dtype in ("no", "bf16"):
    w = SDXLVAEWrapper("synthetic:42", device=torch.device("cuda"))
    tr = HipTrainer(w, mixed_precision=dtype)
    B = int(os.environ.get("B", 16))
    x = torch.rand(B, 3, 256, 256, device="cuda") * 2 - 1
    eps = torch.randn(B, 4, 32, 32, device="cuda")
    for _ in range(2):
        tr.train_step(x, eps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        tr.train_step(x, eps)
    t_enq = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"mixed_precision={dtype}: enqueue {t_enq*1e3:.1f} ms/step, wall {t_all*1e3:.1f} ms/step", flush=True)
    del w, tr
    torch.cuda.empty_cache()
