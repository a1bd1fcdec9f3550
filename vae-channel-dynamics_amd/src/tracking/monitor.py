"""ActivityMonitor with the reference's API and aggregation semantics
(reference src/tracking/monitor.py:11-274), served by fused device-side reductions.

Same constructor, layer ids (`f"{name}.{capture_point}"`), `step()` keys, `get_data_for_step`,
`export_all_processed_data_to_records`, `remove_hooks`.  What changes is WHERE the numbers come
from: when a target layer belongs to the HIP engine and asks only for
`mean_abs_activation_per_channel`, no torch hook is registered; the engine emits the per-channel
vector from a fused reduction (no activation tensor, no device->host sync per forward -- the
reference does `.cpu().numpy()` inside every hook, monitor.py:67).  Vectors stay on the device
until `step()`, which performs ONE transfer per layer.  Any other metric / any foreign model
falls back to ordinary forward hooks with the reference's formulas.

Reference quirks kept on purpose: values are appended on EVERY forward, train or
eval (validation pollutes the buffer, monitor.py:98-101); `step()` takes the UNWEIGHTED mean over
the buffered forwards (monitor.py:181-182); `full_activation_map` keeps only the first
(monitor.py:166-167).
Deliberate fix: under data parallelism the buffered vectors are averaged across ranks in step()
(the reference classifies rank-0-local statistics, train.py:311).
"""
import logging
from collections import defaultdict
from typing import Any, Callable, Dict, List, Optional

import numpy as np
import torch

logger = logging.getLogger(__name__)

FUSED_METRIC = "mean_abs_activation_per_channel"
KNOWN_METRICS = (FUSED_METRIC, "full_activation_map", "mean_activation", "std_activation")


def _resolve(root: torch.nn.Module, dotted: str) -> torch.nn.Module:
    """getattr chain with a `.module` fallback at every level (DDP-wrapped roots), monitor.py:41-54."""
    cur = root
    for part in dotted.split("."):
        if hasattr(cur, part):
            cur = getattr(cur, part)
        elif hasattr(cur, "module") and hasattr(cur.module, part):
            cur = getattr(cur.module, part)
        else:
            raise AttributeError(f"Model (or its .module) does not have a layer named '{dotted}' (path: {part})")
    return cur


def compute_metrics(tensor: torch.Tensor, metrics: List[str]) -> Dict[str, Any]:
    """slow-path formulas, identical to monitor.py:56-80."""
    out: Dict[str, Any] = {}
    if not isinstance(tensor, torch.Tensor):
        logger.warning(f"Cannot calculate metrics for non-tensor type: {type(tensor)}")
        return out
    for name in metrics:
        try:
            if name == FUSED_METRIC:
                if tensor.ndim >= 2:
                    dims = [0] + list(range(2, tensor.ndim))
                    out[name] = tensor.abs().mean(dim=dims).detach().cpu().numpy()
                else:
                    out[name] = tensor.abs().mean().detach().cpu().numpy()
            elif name == "full_activation_map":
                out[name] = tensor.detach().clone().cpu()
            elif name == "mean_activation":
                out[name] = tensor.mean().detach().cpu().numpy()
            elif name == "std_activation":
                out[name] = tensor.std().detach().cpu().numpy()
            else:
                logger.warning(f"Unknown metric '{name}' requested.")
        except Exception as e:  # metric failures are swallowed, monitor.py:78-79
            logger.error(f"Error calculating metric '{name}': {e}", exc_info=True)
    return out


class ActivityMonitor:
    def __init__(self, model: torch.nn.Module, tracking_config: Dict[str, Any]):
        self.model = model
        self.config = tracking_config
        self.target_layers_config: List[Dict[str, Any]] = self.config.get("target_layers", [])
        self.hook_collected_buffer = defaultdict(lambda: defaultdict(list))
        self.processed_data_by_step = defaultdict(dict)
        self.hooks: list = []
        self.fused_layers: List[str] = []
        self.sync_across_ranks = bool(self.config.get("sync_across_ranks", True))
        if self.config.get("enabled", False):
            self._register_hooks()
            logger.info(f"ActivityMonitor initialized for {len(self.target_layers_config)} target(s) "
                        f"({len(self.fused_layers)} fused on device).")
        else:
            logger.info("ActivityMonitor is disabled in config.")

    # ------------------------------------------------------------------ registration
    def _sink(self, layer_id: str) -> Callable[[torch.Tensor], None]:
        def sink(vec: torch.Tensor):
            self.hook_collected_buffer[layer_id][FUSED_METRIC].append(vec)
        return sink

    def _hook(self, layer_id: str, metrics: List[str], point: str) -> Callable:
        def fn(module, hook_input, hook_output=None):
            if point == "input":
                t = hook_input[0] if isinstance(hook_input, tuple) and hook_input else hook_input
            else:
                t = hook_output
            if isinstance(t, torch.Tensor):
                for k, v in compute_metrics(t, metrics).items():
                    self.hook_collected_buffer[layer_id][k].append(v)
        return fn

    def _register_hooks(self):
        self.remove_hooks()
        self.hook_collected_buffer.clear()
        for conf in self.target_layers_config:
            name = conf.get("name")
            point = conf.get("capture_point", "output")
            if not name:
                logger.warning("Skipping a target_layer entry with no name.")
                continue
            layer_id = f"{name}.{point}"
            metrics = conf.get("metrics", [FUSED_METRIC])
            try:
                layer = _resolve(self.model, name)
                if point not in ("input", "output"):
                    logger.warning(f"Unknown capture_point '{point}' for layer {name}. Skipping.")
                    continue
                engine_ref = getattr(layer, "_vae_engine", None)
                engine = engine_ref() if engine_ref is not None else None
                if engine is not None and list(metrics) == [FUSED_METRIC]:
                    self.hooks.append(engine.add_tracker(layer, point, self._sink(layer_id)))
                    self.fused_layers.append(layer_id)
                    logger.info(f"Registered FUSED device tracker for layer: {name} ({point})")
                elif point == "input":
                    self.hooks.append(layer.register_forward_pre_hook(self._hook(layer_id, metrics, point)))
                    logger.info(f"Registered FORWARD PRE-HOOK for layer: {name} (input)")
                else:
                    self.hooks.append(layer.register_forward_hook(self._hook(layer_id, metrics, point)))
                    logger.info(f"Registered FORWARD HOOK for layer: {name} (output)")
            except AttributeError as e:
                logger.error(f"Could not register hook for {layer_id} (AttributeError): {e}")
            except Exception as e:
                logger.error(f"Unexpected error registering hook for {layer_id}: {e}", exc_info=True)

    def remove_hooks(self):
        for h in self.hooks:
            h.remove()
        self.hooks = []
        self.fused_layers = []

    # ------------------------------------------------------------------ aggregation
    def _to_host(self, values: list) -> list:
        """device vectors of the fused path -> list of np.float32 arrays with ONE transfer."""
        if values and all(isinstance(v, torch.Tensor) and v.is_cuda for v in values):
            stacked = torch.stack(values)
            if self.sync_across_ranks and torch.distributed.is_available() and torch.distributed.is_initialized() \
                    and torch.distributed.get_world_size() > 1:
                # ranks must have buffered the same number of forwards (train.py makes them skip batches together)
                n = torch.tensor([stacked.shape[0], -stacked.shape[0]], device=stacked.device, dtype=torch.int64)
                torch.distributed.all_reduce(n, op=torch.distributed.ReduceOp.MAX)
                if int(n[0]) != -int(n[1]):
                    raise RuntimeError(f"ActivityMonitor: ranks buffered different numbers of forwards "
                                       f"({stacked.shape[0]} here, {int(n[0])} max, {-int(n[1])} min); the per-rank "
                                       f"tracker vectors cannot be averaged")
                torch.distributed.all_reduce(stacked)
                stacked = stacked / torch.distributed.get_world_size()
            host = stacked.cpu().numpy()
            return [host[i] for i in range(host.shape[0])]
        return values

    def step(self, global_step: int) -> Dict[str, Any]:
        if not self.config.get("enabled", False):
            return {}
        if global_step % self.config.get("track_interval", 100) != 0:
            return {}
        log: Dict[str, Any] = {}
        processed: Dict[str, Dict[str, Any]] = {}
        for layer_id, metric_data in self.hook_collected_buffer.items():
            processed[layer_id] = {}
            for metric, values in metric_data.items():
                if not values:
                    continue
                agg = None
                key = f"tracking/{layer_id}/{metric}"
                try:
                    if metric == "full_activation_map":
                        agg = values[0]
                        arr = agg.numpy() if isinstance(agg, torch.Tensor) else agg
                        if isinstance(arr, np.ndarray):
                            log[key + "_mean"] = np.mean(arr.astype(np.float32))
                            log[key + "_std"] = np.std(arr.astype(np.float32))
                    elif FUSED_METRIC in metric:
                        vals = self._to_host(values)
                        if all(isinstance(v, np.ndarray) for v in vals):
                            agg = np.mean(np.stack(vals), axis=0)
                            log[key + "_overall_mean"] = np.mean(agg)
                            log[key + "_overall_std"] = np.std(agg)
                        else:
                            agg = vals[0]
                            if isinstance(agg, np.ndarray):
                                log[key + "_overall_mean"] = np.mean(agg)
                                log[key + "_overall_std"] = np.std(agg)
                            elif agg is not None:
                                log[key + "_overall_mean"] = float(agg)
                    else:
                        agg = np.mean([v.item() if hasattr(v, "item") else float(v) for v in values])
                        log[key] = agg
                except Exception as e:
                    logger.error(f"Error aggregating metric {metric} for {layer_id} in step(): {e}", exc_info=True)
                    agg = values[0]
                if agg is not None:
                    processed[layer_id][metric] = agg
        if processed:
            self.processed_data_by_step[global_step] = processed
            logger.info(f"ActivityMonitor collected and processed data for step {global_step}.")
        self.hook_collected_buffer.clear()
        return log

    def get_data_for_step(self, global_step: int) -> Dict[str, Any]:
        return self.processed_data_by_step.get(global_step, {})

    def export_all_processed_data_to_records(self) -> List[Dict[str, Any]]:
        recs: List[Dict[str, Any]] = []
        for gs, step_data in self.processed_data_by_step.items():
            for layer_id, metrics in step_data.items():
                for metric, value in metrics.items():
                    base = {"global_step": gs, "layer_identifier": layer_id, "original_metric_name": metric}

                    def add(kind, val):
                        recs.append({**base, "metric_type": kind, "metric_value": val})

                    if isinstance(value, torch.Tensor):
                        arr = value.numpy()
                    elif isinstance(value, np.ndarray):
                        arr = value
                    else:
                        add("scalar", float(value))
                        continue
                    if arr.ndim == 0:
                        add("scalar", float(arr.item()))
                    elif metric == "full_activation_map":
                        a32 = arr.astype(np.float32)
                        add("full_map_shape", str(arr.shape))
                        add("full_map_mean", float(np.mean(a32)))
                        add("full_map_std", float(np.std(a32)))
                        add("full_map_min", float(np.min(a32)))
                        add("full_map_max", float(np.max(a32)))
                    elif FUSED_METRIC in metric:
                        add("per_channel_overall_mean", float(np.mean(arr)))
                        add("per_channel_overall_std", float(np.std(arr)))
                        add("per_channel_overall_min", float(np.min(arr)))
                        add("per_channel_overall_max", float(np.max(arr)))
                    else:
                        a32 = arr.astype(np.float32)
                        add("array_mean", float(np.mean(a32)))
                        add("array_std", float(np.std(a32)))
        return recs

    def __del__(self):
        try:
            self.remove_hooks()
        except Exception:
            pass
