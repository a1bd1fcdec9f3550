"""YAML config loading with the reference's single-level `defaults: [base]` inheritance
(reference src/utils/config_utils.py:9-65).  The merge is SHALLOW on purpose: a nested dict in
the experiment file replaces the base's dict wholesale (config_utils.py:52-55), which is why
experiment configs silently fall back to the code-side defaults of train.py."""
import logging
import os
from typing import Any, Dict

import yaml

logger = logging.getLogger(__name__)


def _read(path: str) -> Dict[str, Any]:
    with open(path, "r") as f:
        data = yaml.safe_load(f)
    return data or {}


def load_config(config_path: str) -> Dict[str, Any]:
    if not os.path.exists(config_path):
        raise FileNotFoundError(f"Configuration file not found: {config_path}")
    cfg = _read(config_path)
    merged: Dict[str, Any] = {}
    defaults = cfg.get("defaults")
    if isinstance(defaults, list) and "defaults" in cfg:
        base_path = os.path.join(os.path.dirname(config_path), f"{defaults[0]}.yaml")
        logger.info(f"Loading base configuration from: {base_path}")
        if not os.path.exists(base_path):
            raise FileNotFoundError(f"Base configuration file not found: {base_path}")
        merged.update(_read(base_path))
        del cfg["defaults"]
    merged.update(cfg)
    logger.info(f"Successfully loaded configuration from {config_path}")
    return merged
