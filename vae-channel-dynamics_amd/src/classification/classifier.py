"""RegionClassifier: flags GroupNorm channels whose tracked statistic is below a threshold
(reference src/classification/classifier.py:10-151; same config keys, same output dict).

The result is an integer mask, so it must be bit-identical to the reference:
the comparison is done exactly as NumPy 2 does `float32_array < python_float`
(NEP 50: the threshold is rounded to float32 first), classifier.py:135.
"""
import logging
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

logger = logging.getLogger(__name__)


class RegionClassifier:
    def __init__(self, model: Optional[torch.nn.Module], config: Dict[str, Any]):
        self.config = config
        self.method = config.get("method", "threshold_groupnorm_activity")
        self.threshold = float(config.get("threshold", 1e-3))
        self.target_metric_key = config.get("target_metric_key", "mean_abs_activation_per_channel")
        self.layers_to_classify: List[str] = config.get("layers_to_classify", [])
        self._layer_to_param_map: Dict[str, Tuple[str, int]] = {}
        if model is not None:
            self._build_groupnorm_map(model)
        else:
            logger.warning("RegionClassifier initialised without a model - parameter mapping will be heuristic only.")
        logger.info(f"RegionClassifier initialised (method={self.method}, thr={self.threshold}, "
                    f"metric={self.target_metric_key}, map_size={len(self._layer_to_param_map)})")
        if not self._layer_to_param_map:
            logger.warning("RegionClassifier: no GroupNorm layers found / mapped.")

    def _build_groupnorm_map(self, model: torch.nn.Module):
        """`<gn>.output` and `vae.<gn>.output` -> (`<gn>.weight`, C) for every affine nn.GroupNorm (classifier.py:43-81)."""
        for mod_name, mod in model.named_modules():
            if not isinstance(mod, torch.nn.GroupNorm):
                continue
            if not isinstance(getattr(mod, "weight", None), torch.nn.Parameter):
                continue
            entry = (f"{mod_name}.weight", mod.num_channels)
            key = f"{mod_name}.output"
            self._layer_to_param_map[key] = entry
            if not mod_name.startswith("vae."):
                self._layer_to_param_map[f"vae.{key}"] = entry

    def _lookup_param_info(self, layer_id: str) -> Optional[Tuple[str, int]]:
        hit = self._layer_to_param_map.get(layer_id)
        if hit is None and "." in layer_id:
            hit = self._layer_to_param_map.get(layer_id.split(".", 1)[1])
        return hit

    def classify(self, tracked_data_for_step: Dict[str, Any], global_step: int) -> Dict[str, Any]:
        if not self.config.get("enabled", False):
            return {}
        results: Dict[str, Any] = {}
        if self.method != "threshold_groupnorm_activity":
            logger.warning(f"Unknown classification method: {self.method}")
            return results
        if not tracked_data_for_step:
            return results
        for layer_id, metrics in tracked_data_for_step.items():
            if self.layers_to_classify and layer_id not in self.layers_to_classify:
                continue
            vals = metrics.get(self.target_metric_key)
            if not (isinstance(vals, np.ndarray) and vals.ndim == 1):
                continue
            info = self._lookup_param_info(layer_id)
            if info is None:
                continue
            pname, num_ch = info
            if vals.shape[0] != num_ch:
                logger.warning(f"{layer_id}: channel mismatch ({vals.shape[0]} vs {num_ch}) - skipped.")
                continue
            idx = np.where(vals < self.threshold)[0]
            if idx.size == 0:
                continue
            results[layer_id] = {
                "param_name_scale": pname,
                "inactive_channel_indices": idx.tolist(),
                "metric_used": self.target_metric_key,
                "threshold_value": self.threshold,
                "values_of_inactive_channels": vals[idx].tolist(),
            }
            logger.info(f"Step {global_step}: {layer_id} -> {len(idx)} inactive channels (param {pname})")
        logger.info(f"Classification complete - {len(results)} layer(s) flagged.")
        return results
