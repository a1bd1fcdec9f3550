"""stdout (+ optional file) logging setup, as reference src/utils/logging_utils.py:6-25."""
import logging
import sys
from typing import Optional


def setup_logging(log_level: int = logging.INFO, log_file: Optional[str] = None):
    handlers = [logging.StreamHandler(sys.stdout)]
    if log_file:
        handlers.append(logging.FileHandler(log_file))
    logging.basicConfig(level=log_level, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s",
                        datefmt="%Y-%m-%d %H:%M:%S", handlers=handlers)
    logging.getLogger(__name__).info(f"Logging setup complete. Level: {logging.getLevelName(log_level)}")
