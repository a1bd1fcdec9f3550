from .classifier import RegionClassifier  # noqa: F401
