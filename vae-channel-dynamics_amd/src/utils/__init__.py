from .config_utils import load_config  # noqa: F401
from .logging_utils import setup_logging  # noqa: F401
