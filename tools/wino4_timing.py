"""Where a workgroup of the fp32 Winograd F(4x4,3x3) forward (csrc/conv3_wino4.hip) spends its time: shader-clock stamps of waves
0, 4, 8 (the three waves of SIMD 0; wave 4 multiplies first, the others stage first) of workgroups 0..7, instrumented build
(`make -C csrc timing`, loaded through VAEHIP_LIB).  Per step 24 MFMAs of 64 cycles per wave = 1536 cycles of matrix work per
wave, 4608 per SIMD.  The stamps are intrusive: read the phases as a picture, not as the production kernel's times.
usage: VAEHIP_LIB=vae-channel-dynamics_amd/csrc/libvaehip_timing.so python tools/wino4_timing.py [c128 c256 c512]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from vaehip import ops  # noqa: E402

SHAPES = {"c128": (16, 256, 128, 128), "c256": (16, 128, 256, 256), "c512": (16, 64, 512, 512)}
for nm in (sys.argv[1:] or list(SHAPES)):
    B, H, Ci, Co = SHAPES[nm]
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((B, H, H, Ci), device="cuda", generator=g)
    bias = torch.randn(Co, device="cuda", generator=g)
    w = (torch.randn((Co, 3, 3, Ci), device="cuda", generator=g) / math.sqrt(9 * Ci)).permute(0, 3, 1, 2)
    stamps = torch.zeros(max(8 * 3 * 42, ((B * H * H + 127) // 128) * Co // 2 + 1), device="cuda", dtype=torch.int64)
    tr = stamps.view(torch.float32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(40):
        stamps.zero_()
        e0.record()
        ops.conv_fwd(x, w, bias, "c3", track=tr)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    t = stamps[:8 * 3 * 42].cpu().numpy().reshape(8, 3, 42).astype("float64")
    print(f"{nm}: {ms:.3f} ms incl. the weight transform; {Ci // 8} steps per workgroup", flush=True)
    for wg in range(8):
        for wv in range(3):
            v = t[wg, wv, :10]
            if v[0] == 0:
                continue
            st = t[wg, wv, 10:].reshape(8, 4)
            ph1, ph2, bar = np.mean(st[:, 1] - st[:, 0]), np.mean(st[:, 2] - st[:, 1]), np.mean(st[:, 3] - st[:, 2])
            per = (st[-1, 3] - st[0, 0]) / 8
            names = ("stage", "A+mfma") if wv != 1 else ("A+mfma", "stage")
            print(f"  wg {wg} wave {4 * wv}: prologue {v[1]-v[0]:6.0f}  main loop {v[2]-v[1]:7.0f} ({(v[2]-v[1]) / (Ci // 8):5.0f}/step)  block 0: to LDS {v[3]-v[2]:5.0f} "
                  f"transform+store {v[4]-v[3]:6.0f}  block 1: to LDS {v[5]-v[4]:5.0f} transform+store {v[6]-v[5]:6.0f}  tail {v[7]-v[6]:5.0f}  total {v[7]-v[0]:7.0f}  "
                  f"clock {(v[7]-v[0])/max(v[9]-v[8],1)*0.1:5.2f} GHz | steps 2..9: {per:5.0f}/step: {names[0]} {ph1:5.0f} {names[1]} {ph2:5.0f} barrier {bar:5.0f}", flush=True)
