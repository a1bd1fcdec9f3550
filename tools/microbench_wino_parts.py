import math, os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/vae-channel-dynamics_amd") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from vaehip import ops
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for nm,(B,H,Ci,Co) in {"c128": (16, 256, 128, 128), "c512": (16, 64, 512, 512)}.items():
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((B, H, H, Ci), device="cuda", generator=g)
    res = torch.randn((B, H, H, Co), device="cuda", generator=g)
    bias = torch.randn(Co, device="cuda", generator=g)
    w = (torch.randn((Co, 3, 3, Ci), device="cuda", generator=g) / math.sqrt(9 * Ci)).permute(0, 3, 1, 2)
    st = ops.gn_stats(x, torch.ones(Ci, device="cuda"), torch.zeros(Ci, device="cuda"))
    runs = {"plain": lambda: ops.conv_fwd(x, w, None, "c3"),
            "bias": lambda: ops.conv_fwd(x, w, bias, "c3"),
            "res": lambda: ops.conv_fwd(x, w, bias, "c3", res=res),
            "gstat": lambda: ops.conv_fwd(x, w, bias, "c3", gstat_groups=32),
            "xf": lambda: ops.conv_fwd(x, w, bias, "c3", xf=ops.XF_AFFINE_SILU, stats=st),
            "xf_res_gstat": lambda: ops.conv_fwd(x, w, bias, "c3", xf=ops.XF_AFFINE_SILU, stats=st, res=res, gstat_groups=32)}
    for k, fn in runs.items():
        print(f"{nm} {k:14s} {timeit(fn):7.3f} ms", flush=True)
