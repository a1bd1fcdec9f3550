from .nudger import InterventionHandler  # noqa: F401
