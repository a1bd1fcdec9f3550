"""InterventionHandler: nudges GroupNorm scale parameters of channels the classifier flagged
(reference src/intervention/nudger.py:10-172; same config keys, gating and arithmetic).

The arithmetic is the reference's, bit for bit: the fp32 scale is read as a Python float
(float64), multiplied by `nudge_factor` in float64, capped with min(., max_scale_value) and
stored back with one rounding to fp32 (nudger.py:130-140).  Instead of one `.item()` host sync
per channel the flagged entries are gathered in one transfer, updated in float64 on the host and
scattered back in one; with duplicate indices the sequential loop is used so that repeated
nudges compound exactly as in the reference.
The parameters are edited in place in the live arena the HIP kernels read every step.
"""
import logging
from typing import Any, Dict, Optional

import numpy as np
import torch
import torch.nn as nn

logger = logging.getLogger(__name__)


class InterventionHandler:
    def __init__(self, model: nn.Module, config: Dict[str, Any]):
        self.model = model
        self.config = config
        self.strategy = config.get("strategy", "none")
        self.nudge_factor = float(config.get("nudge_factor", 1.1))
        self.nudge_value_add = float(config.get("nudge_value_add", 0.01))
        self.max_scale_value = float(config.get("max_scale_value", 2.0))
        self.num_nudges_applied = 0
        logger.info(f"InterventionHandler initialized (strategy: {self.strategy}, model type: {type(model)})")

    def _get_parameter(self, param_name: str) -> Optional[nn.Parameter]:
        obj = self.model
        for part in param_name.split("."):
            if not hasattr(obj, part):
                logger.error(f"Model does not have attribute '{part}' in path '{param_name}'.")
                return None
            obj = getattr(obj, part)
        if isinstance(obj, nn.Parameter):
            return obj
        logger.error(f"Attribute '{param_name}' is not a Parameter, but {type(obj)}.")
        return None

    def _apply(self, p: nn.Parameter, indices, fn) -> int:
        n = p.data.numel()
        valid = [int(i) for i in indices if 0 <= int(i) < n]
        for i in indices:
            if not (0 <= int(i) < n):
                logger.warning(f"Inactive index {i} out of bounds (size: {n})")
        if not valid:
            return 0
        with torch.no_grad():
            if len(set(valid)) != len(valid):
                for i in valid:  # duplicates: sequential, exactly as the reference loop
                    p.data[i] = fn(np.float64(p.data[i].item())).item()
            else:
                idx = torch.tensor(valid, dtype=torch.long, device=p.device)
                cur = p.data.reshape(-1)[idx].to("cpu", torch.float64).numpy()
                new = torch.from_numpy(np.asarray(fn(cur), dtype=np.float64)).to(torch.float32)
                p.data.reshape(-1)[idx] = new.to(p.device)
        return len(valid)

    def intervene(self, classification_results: Dict[str, Any], global_step: int):
        if not self.config.get("enabled", False) or self.strategy == "none":
            return
        interval = self.config.get("intervention_interval", 200)
        if global_step == 0 or global_step % interval != 0:
            if not (interval == 1 and global_step > 0):
                return
        if not classification_results:
            logger.info(f"Step {global_step}: No regions classified by RegionClassifier, skipping intervention.")
            return
        self.num_nudges_applied = 0
        if self.strategy == "gentle_nudge_groupnorm_scale":
            fn = lambda v: np.minimum(v * self.nudge_factor, self.max_scale_value)  # noqa: E731
        elif self.strategy == "reset_groupnorm_scale":
            fn = lambda v: np.ones_like(v)  # noqa: E731
        else:
            logger.warning(f"Unknown intervention strategy: {self.strategy}")
            return
        for layer_key, data in classification_results.items():
            pname = data.get("param_name_scale")
            idx = data.get("inactive_channel_indices")
            if not pname or idx is None:
                logger.warning(f"Missing 'param_name_scale' or 'inactive_channel_indices' for layer_key '{layer_key}'. Skipping.")
                continue
            p = self._get_parameter(pname)
            if p is None:
                logger.warning(f"Could not retrieve scale parameter '{pname}' for layer_key '{layer_key}'. Skipping.")
                continue
            self.num_nudges_applied += self._apply(p, idx, fn)
        if self.num_nudges_applied > 0:
            logger.info(f"Applied '{self.strategy}' to {self.num_nudges_applied} channel scales at step {global_step}.")
