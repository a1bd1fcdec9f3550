"""bitwise-repeatability stress of the split flat kernels (run several copies at once to share the GPU)"""
import math, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch
from vaehip import ops
g = torch.Generator(device="cuda").manual_seed(1)
bad = {}
N_IT = int(os.environ.get("N_IT", "300"))
cases = [(2, 16, 256, 128, "c1"), (2, 32, 128, 128, "c3s2"), (2, 4, 512, 512, "c1"), (2, 8, 512, 512, "c3"), (2, 4, 512, 512, "c3"),
         (2, 8, 256, 512, "c3"), (2, 16, 128, 256, "c1")]
for (B, H, Ci, Co, kind) in cases:
    x = torch.randn((B, H, H, Ci), device="cuda", generator=g)
    k = 1 if kind == "c1" else 3
    w = (torch.randn((Co, k, k, Ci), device="cuda", generator=g) / math.sqrt(k * k * Ci)).permute(0, 3, 1, 2)
    y0 = ops.conv_fwd(x, w, None, kind)
    dy = torch.randn_like(y0)
    d0 = ops.conv_dgrad(dy, w, kind, (H, H))
    for it in range(N_IT):
        y = ops.conv_fwd(x, w, None, kind)
        d = ops.conv_dgrad(dy, w, kind, (H, H))
        if not torch.equal(y, y0):
            bad[(kind, H, Ci, Co, "fwd")] = bad.get((kind, H, Ci, Co, "fwd"), 0) + 1
        if not torch.equal(d, d0):
            bad[(kind, H, Ci, Co, "dgrad")] = bad.get((kind, H, Ci, Co, "dgrad"), 0) + 1
    torch.cuda.synchronize()
q = torch.randn((2, 16, 512), device="cuda", generator=g)
kk = torch.randn((2, 16, 512), device="cuda", generator=g)
p0 = ops.gemm_nt(q, kk, 0.044)
for it in range(N_IT):
    if not torch.equal(ops.gemm_nt(q, kk, 0.044), p0):
        bad[("gemm_nt",)] = bad.get(("gemm_nt",), 0) + 1
print("mismatches", bad if bad else 0)
