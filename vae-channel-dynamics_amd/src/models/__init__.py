from .sdxl_vae_wrapper import SDXLVAEWrapper  # noqa: F401
