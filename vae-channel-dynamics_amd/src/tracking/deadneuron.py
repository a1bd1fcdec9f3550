"""DeadNeuronTracker: percentage of near-zero weights per parameter
(reference src/tracking/deadneuron.py:11-115; same constructor, modes and history layout).

The reference does three torch ops and a `.item()` host sync per parameter (248 syncs per call).
When the parameters live in the HIP arena this tracker runs ONE multi-segment scan kernel over
the flat 335 MB arena (counts of |w| < thr and sum|w| per parameter) and reads the small result
vectors back once; other models fall back to the reference's per-parameter formulas.
"""
import ctypes as C
import logging
from collections import defaultdict
from typing import List, Tuple, Type

import numpy as np
import torch
import torch.nn as nn

logger = logging.getLogger(__name__)


class DeadNeuronTracker:
    def __init__(self, target_layer_classes: Tuple[Type[nn.Module], ...], target_layer_names_for_raw_weights: List[str],
                 threshold: float, mean_percentage: float, dead_type: str = "threshold"):
        self.threshold = threshold
        self.mean_percentage = mean_percentage
        self.target_layer_classes = target_layer_classes
        self.target_layer_names_for_raw_weights = target_layer_names_for_raw_weights
        self.dead_type = dead_type
        modes = {"threshold": self.smaller_than_threshold, "percent_of_mean": self.percent_of_mean, "both": self.both}
        if dead_type not in modes:
            logger.warning(f"Unknown dead_type: {dead_type}. Defaulting to no-op for percentage calculation.")
        self.get_percentage = modes.get(dead_type, self.noop)
        self.weights_history = defaultdict(list)
        self.percent_history = defaultdict(list)
        self._plan = None

    # ------------------------------------------------------------------ selection (deadneuron.py:51-62)
    def _eligible(self, model: nn.Module):
        out = []
        for name, param in model.named_parameters():
            if not param.requires_grad:
                continue
            if name in self.target_layer_names_for_raw_weights:
                self.weights_history[name] = [param.detach().cpu().numpy()]
            if "weight" in name or "bias" in name:
                try:
                    module = model.get_submodule(".".join(name.split(".")[:-1]))
                except AttributeError:
                    continue
                if isinstance(module, self.target_layer_classes):
                    out.append((name, param))
        return out

    def track_dead_neurons(self, model_wrapper: nn.Module, global_step: int):
        if hasattr(model_wrapper, "vae") and model_wrapper.vae is not None:
            model = model_wrapper.vae
        elif isinstance(model_wrapper, nn.Module):
            model = model_wrapper
        else:
            logger.error("DeadNeuronTracker: model_wrapper is not an nn.Module or has no .vae attribute.")
            return
        items = self._eligible(model)
        arena = getattr(model, "arena", None)
        if arena is not None and arena.flat.is_cuda and self.get_percentage != self.noop and items:
            pcts = self._scan_arena(arena, items)
        else:
            pcts = [self.get_percentage(p) for _, p in items]
        for (name, _), pct in zip(items, pcts):
            self.percent_history[name].append((global_step, pct))

    # ------------------------------------------------------------------ fused arena scan
    def _scan_arena(self, arena, items):
        from vaehip.lib import lib
        dev = arena.flat.device
        key = (id(arena), tuple(n for n, _ in items))
        if self._plan is None or self._plan[0] != key:
            offs = []
            for _, p in items:
                o = arena.offset_of[id(p)]
                offs.append((o, o + p.numel()))
            seg = torch.tensor([[b, e] for b, e in offs], dtype=torch.int64)
            ch = lib.query("vae_dead_scan_chunk")  # elements per workgroup
            c0 = np.concatenate([[0], np.cumsum([max(1, -(-(e - b) // ch)) for b, e in offs])]).astype(np.int32)
            self._plan = (key, seg.to(dev).contiguous().view(-1), torch.tensor([e - b for b, e in offs], dtype=torch.float64),
                          torch.from_numpy(c0).to(dev), int(c0[-1]))
        _, seg_dev, numel, chunk0, nchunk = self._plan  # seg_dev: [n][2] = {begin, end} offsets into the arena
        n = numel.shape[0]
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        counts = torch.zeros(n, dtype=torch.int64, device=dev)
        abssum = torch.zeros(n, dtype=torch.float64, device=dev)
        pcnt = torch.empty(nchunk, dtype=torch.int64, device=dev)
        psum = torch.empty(nchunk, dtype=torch.float64, device=dev)
        base = arena.flat
        lib.call("vae_dead_scan", C.c_void_p(base.data_ptr()), C.c_void_p(seg_dev.data_ptr()), C.c_void_p(chunk0.data_ptr()), n, nchunk,
                 float(np.float32(self.threshold)), C.c_void_p(pcnt.data_ptr()), C.c_void_p(psum.data_ptr()),
                 C.c_void_p(counts.data_ptr()), C.c_void_p(abssum.data_ptr()), stream)
        if self.dead_type == "threshold":
            c = counts.cpu().numpy().astype(np.float64)
            return [float(ci / ni * 100.0) for ci, ni in zip(c, numel.numpy())]
        mean_abs = (abssum.cpu() / numel).to(torch.float32).numpy()  # torch: param_abs.mean().item() (fp32)
        degenerate = np.abs(mean_abs.astype(np.float64)) < 1e-9
        athr = np.where(degenerate, 1e-9, self.mean_percentage * mean_abs.astype(np.float64)).astype(np.float32)
        athr_dev = torch.from_numpy(athr).to(dev)
        use_fixed = 1 if self.dead_type == "both" else 0
        lib.call("vae_dead_scan_adaptive", C.c_void_p(base.data_ptr()), C.c_void_p(seg_dev.data_ptr()), C.c_void_p(chunk0.data_ptr()), n,
                 nchunk, float(np.float32(self.threshold)), use_fixed, C.c_void_p(athr_dev.data_ptr()),
                 C.c_void_p(pcnt.data_ptr()), C.c_void_p(counts.data_ptr()), stream)
        c = counts.cpu().numpy().astype(np.float64)
        out = []
        for ci, ni, deg in zip(c, numel.numpy(), degenerate):
            if self.dead_type == "percent_of_mean" and deg:
                out.append(100.0 if ci == ni else 0.0)  # deadneuron.py:86-88
            else:
                out.append(float(ci / ni * 100.0))
        return out

    # ------------------------------------------------------------------ reference formulas (any tensor, any device)
    def noop(self, param):
        return 0.0

    def smaller_than_threshold(self, param: torch.Tensor) -> float:
        n = param.numel()
        if n == 0:
            return 0.0
        return ((param.abs() < self.threshold).sum().item() / n) * 100.0

    def percent_of_mean(self, param: torch.Tensor) -> float:
        if param.numel() == 0:
            return 0.0
        a = param.abs()
        mean_abs = a.mean().item()
        if abs(mean_abs) < 1e-9:
            return 100.0 if (a < 1e-9).all().item() else 0.0
        return ((a < self.mean_percentage * mean_abs).sum().item() / param.numel()) * 100.0

    def both(self, param: torch.Tensor) -> float:
        n = param.numel()
        if n == 0:
            return 0.0
        a = param.abs()
        fixed = a < self.threshold
        mean_abs = a.mean().item()
        adaptive = (a < 1e-9) if abs(mean_abs) < 1e-9 else (a < self.mean_percentage * mean_abs)
        return ((fixed & adaptive).sum().item() / n) * 100.0
