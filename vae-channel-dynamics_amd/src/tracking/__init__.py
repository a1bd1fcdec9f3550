from .monitor import ActivityMonitor  # noqa: F401
from .deadneuron import DeadNeuronTracker  # noqa: F401
