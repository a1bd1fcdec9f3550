"""MI355X-native kernels of the SDXL-VAE train step (ctypes binding of libvaehip.so).

Fails loudly if the HIP library is missing: there is no CPU fallback in the product path.
"""
from .lib import lib, VaeHipError, LIB_PATH  # noqa: F401
