"""Where a step (8 input channels) of the fp32 Winograd forward spends its time: shader-clock stamps of waves 0 and 4 (the two
waves of workgroup w on SIMD 0: wave 0 stages then multiplies, wave 4 multiplies then stages) of workgroups 0..7, instrumented
build (`make -C csrc timing`, loaded through VAEHIP_LIB).  Per step: 16 MFMAs of 64 cycles per wave = 1024 cycles of matrix work
per wave, 4 waves per SIMD (two workgroups per CU).  The stamps are intrusive (each one waits for the wave's LDS operations and
the build spills more registers): read the phases as a picture, not as the production kernel's times.
usage: VAEHIP_LIB=vae-channel-dynamics_amd/csrc/libvaehip_timing.so python tools/wino_timing.py [c128 c256 c512]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from vaehip import ops  # noqa: E402

SHAPES = {"c128": (16, 256, 128, 128), "c256": (16, 128, 256, 256), "c512": (16, 64, 512, 512)}
ops.PRECISION = ops.PREC_F32
for nm in (sys.argv[1:] or list(SHAPES)):
    B, H, Ci, Co = SHAPES[nm]
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn((B, H, H, Ci), device="cuda", generator=g)
    bias = torch.randn(Co, device="cuda", generator=g)
    wbuf = torch.randn((Co, 3, 3, Ci), device="cuda", generator=g) / math.sqrt(9 * Ci)
    w = wbuf.permute(0, 3, 1, 2)
    stamps = torch.zeros(max(8 * 2 * 64 * 6 + 8 * 2 * 10, ((B * H * H + 127) // 128) * Co // 2 + 1), device="cuda", dtype=torch.int64)
    tr = stamps.view(torch.float32)  # (the kernel reinterprets the pointer)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(60):  # (the clock settles over the first launches)
        stamps.zero_()
        e0.record()
        ops.conv_fwd(x, w, bias, "c3", track=tr)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    t = stamps[:8 * 2 * 64 * 6].cpu().numpy().reshape(8, 2, 64, 6).astype("float64")
    nst = min(Ci // 8, 64)
    print(f"{nm}: {ms:.3f} ms incl. the weight transform; {nst} steps per workgroup", flush=True)
    for wg in range(8):
        a, b = t[wg, 0, :nst], t[wg, 1, :nst]
        if a[0, 0] == 0:
            continue
        sa, sb = a[1:-1], b[1:-1]  # steady state: without the first and the last step
        per = (a[-1, 5] - a[0, 0]) / nst
        print(f"  wg {wg}: step {per:7.0f} ticks | wave 0: stage {np.mean(sa[:,2]-sa[:,1]):5.0f} A-wait {np.mean(sa[:,3]-sa[:,2]):5.0f} mfma-issue {np.mean(sa[:,4]-sa[:,3]):5.0f} "
              f"barrier {np.mean(sa[:,5]-sa[:,4]):5.0f} | wave 4: A-wait {np.mean(sb[:,2]-sb[:,1]):5.0f} mfma-issue {np.mean(sb[:,3]-sb[:,2]):5.0f} "
              f"stage {np.mean(sb[:,4]-sb[:,3]):5.0f} barrier {np.mean(sb[:,5]-sb[:,4]):5.0f}", flush=True)
    e = stamps[8 * 2 * 64 * 6:8 * 2 * 64 * 6 + 8 * 2 * 10].cpu().numpy().reshape(8, 2, 10).astype("float64")
    for wg in range(8):
        v = e[wg, 0]
        if v[0] == 0:
            continue
        print(f"  wg {wg} wave 0: prologue {v[1]-v[0]:6.0f}  main loop {v[2]-v[1]:7.0f}  block 0: to LDS {v[3]-v[2]:6.0f} transform+store {v[4]-v[3]:6.0f}  "
              f"block 1: to LDS {v[5]-v[4]:6.0f} transform+store {v[6]-v[5]:6.0f}  tail {v[7]-v[6]:6.0f}  total {v[7]-v[0]:7.0f}  clock {(v[7]-v[0])/max(v[9]-v[8],1)*0.1:5.2f} GHz", flush=True)
    a = t[0, 0, :nst]
    print("  wg 0 wave 0, first 6 steps (ticks since its first stamp):")
    for s in range(min(6, nst)):
        print("    " + " ".join(f"{v - a[0, 0]:8.0f}" for v in a[s]))
