"""Where a tile of the wide-tile bf16 kernel spends its time: shader-clock stamps of workgroup 0 (instrumented build, `make -C
csrc timing`, loaded through VAEHIP_LIB).  Per tile: main loop, output epilogue, statistics epilogue.
usage: VAEHIP_LIB=vae-channel-dynamics_amd/csrc/libvaehip_timing.so python tools/wide_timing.py [c128 c256 c512]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch  # noqa: E402
from vaehip import ops  # noqa: E402

SHAPES = {"c128": (16, 256, 128, 128), "c256": (16, 128, 256, 256), "c512": (16, 64, 512, 512)}
ops.PRECISION = ops.PREC_BF16
for nm in (sys.argv[1:] or list(SHAPES)):
    B, H, Ci, Co = SHAPES[nm]
    g = torch.Generator(device="cuda").manual_seed(0)
    x16 = torch.randn((B, H, H, Ci), device="cuda", generator=g).bfloat16()
    res16 = torch.randn((B, H, H, Co), device="cuda", generator=g).bfloat16()
    bias = torch.randn(Co, device="cuda", generator=g)
    wbuf = torch.randn((Co, 3, 3, Ci), device="cuda", generator=g) / math.sqrt(9 * Ci)
    w = wbuf.permute(0, 3, 1, 2)
    img = torch.empty(wbuf.numel(), device="cuda", dtype=torch.bfloat16)
    ops.pack_bf16(wbuf, img)
    ops.WEIGHTS16 = (wbuf.data_ptr(), wbuf.numel() * 4, img.data_ptr())
    for label, kw in (("plain", {}), ("bias+gstat", {"gstat_groups": 32, "b": True}), ("bias+gstat+res", {"gstat_groups": 32, "b": True, "res": res16})):
        stamps = torch.zeros(max(64 * 4, ((B * H * H + 127) // 128) * Co // 2 + 1), device="cuda", dtype=torch.int64)
        tr = stamps.view(torch.float32)  # (the kernel reinterprets the pointer)
        for _ in range(2):
            stamps.zero_()
            ops.conv_fwd(x16, w, bias if kw.get("b") else None, "c3", a16=x16, res=kw.get("res"), gstat_groups=kw.get("gstat_groups"), track=tr)
        torch.cuda.synchronize()
        t = stamps[:256].cpu().view(64, 4).numpy()
        n = int((t[:, 0] > 0).sum())
        rows = t[:n].astype("float64")
        main, epi, gst = rows[:, 1] - rows[:, 0], rows[:, 2] - rows[:, 1], rows[:, 3] - rows[:, 2]
        gap = rows[1:, 0] - rows[:-1, 3] if n > 1 else [0]
        print(f"{nm} {label:16s} tiles/WG {n:3d}  main {main.mean():9.0f}  epilogue {epi.mean():8.0f}  stats {gst.mean():7.0f}  between {sum(gap)/max(len(gap),1):6.0f}  "
              f"(shader-clock ticks; total per tile {(main+epi+gst).mean():9.0f})", flush=True)
