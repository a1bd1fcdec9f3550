"""Dead-weight scan (vae_dead_scan): time of one DeadNeuronTracker pass over the 83.65 M parameters and the HBM
rate it implies (4 B per parameter per pass; the 'threshold' mode is one pass, 'percent_of_mean' / 'both' two).
usage: python tools/dead_scan_bench.py   (prints one JSON line; run under rocprofv3 --kernel-trace --stats for the kernel rows)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
import torch  # noqa: E402
from models.sdxl_vae_wrapper import SDXLVAEWrapper  # noqa: E402
from tracking.deadneuron import DeadNeuronTracker  # noqa: E402

TARGET = (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear, torch.nn.GroupNorm)


def main():
    w = SDXLVAEWrapper("synthetic:42", device=torch.device("cuda"))
    n = sum(p.numel() for p in w.vae.parameters())
    out = {"parameters": n}
    for mode, passes in (("threshold", 1), ("both", 2)):
        t = DeadNeuronTracker(TARGET, [], 1e-5, 0.1, mode)
        t.track_dead_neurons(w, 0)  # plan + warm-up
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        t0 = time.perf_counter()
        e0.record()
        for i in range(reps):
            t.track_dead_neurons(w, i + 1)
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        out[mode] = {"wall_ms_per_call_incl_host": round(wall * 1e3, 3), "passes_over_the_arena": passes,
                     "algorithmic_bytes": 4 * n * passes}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
