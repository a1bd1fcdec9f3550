"""Drop-in for the reference's model boundary (reference src/models/sdxl_vae_wrapper.py:10-179),
running on the MI355X HIP engine instead of diffusers.AutoencoderKL.

Same constructor, attributes (`.vae`, `.scaling_factor`), forward contract and hook helpers.
`pretrained_model_name_or_path` is a local directory (config.json + safetensors, the layout
written by `vae.save_pretrained`, reference train.py:412 / evaluate.py:91-102) or
`synthetic[:seed]`; hub names raise (no network), as a failed load does in the reference
(sdxl_vae_wrapper.py:38-40).
"""
import logging
from typing import Callable, Dict, List, Optional, Union

import torch

from vaehip.autoencoder import AutoencoderKLHip

logger = logging.getLogger(__name__)


class SDXLVAEWrapper(torch.nn.Module):
    def __init__(self, pretrained_model_name_or_path: str = "stabilityai/sdxl-vae",
                 torch_dtype: Optional[Union[str, torch.dtype]] = None, device: Optional[torch.device] = None):
        super().__init__()
        self.pretrained_model_name_or_path = pretrained_model_name_or_path
        self.torch_dtype = torch_dtype
        self._init_device = device
        self.vae = self._load_vae()
        self.scaling_factor = self.vae.config.scaling_factor
        self._hook_handles: List[torch.utils.hooks.RemovableHandle] = []
        self._captured_activations: Dict[str, torch.Tensor] = {}

    def _load_vae(self) -> AutoencoderKLHip:
        logger.info(f"Loading VAE model from: {self.pretrained_model_name_or_path}")
        try:
            vae = AutoencoderKLHip.from_pretrained(self.pretrained_model_name_or_path, torch_dtype=self.torch_dtype,
                                                   device=self._init_device)
        except Exception as e:
            logger.error(f"Failed to load VAE model from {self.pretrained_model_name_or_path}: {e}")
            raise
        logger.info(f"VAE model loaded successfully. VAE scaling factor: {vae.config.scaling_factor}")
        return vae

    # reference forward contract: sdxl_vae_wrapper.py:42-77 (latents are NOT scaled in training)
    def forward(self, pixel_values: torch.Tensor, sample_posterior: bool = True):
        latent_dist = self.vae.encode(pixel_values).latent_dist
        latents = latent_dist.sample() if sample_posterior else latent_dist.mode()
        reconstruction = self.vae.decode(latents).sample
        return {"reconstruction": reconstruction, "latent_dist": latent_dist, "latents_sampled": latents}

    # ad-hoc activation capture used by evaluate.py (reference :79-143)
    def _capture(self, name: str) -> Callable:
        def hook(module, input_data, output_data):
            self._captured_activations[name] = output_data.detach().cpu()
        return hook

    def add_hooks(self, layer_names: List[str]):
        self.remove_hooks()
        found = False
        for name, module in self.vae.named_modules():
            if name in layer_names:
                self._hook_handles.append(module.register_forward_hook(self._capture(name)))
                logger.info(f"Registered activation hook for VAE layer: '{name}'")
                found = True
        if not found and layer_names:
            logger.warning(f"No hooks registered. Ensure layer names {layer_names} are correct and exist in the VAE.")

    def remove_hooks(self):
        if not self._hook_handles:
            return
        for h in self._hook_handles:
            h.remove()
        self._hook_handles.clear()
        self._captured_activations.clear()
        logger.info("Cleared all VAE model hooks and captured activations.")

    def get_captured_activations(self) -> Dict[str, torch.Tensor]:
        return self._captured_activations

    def clear_captured_activations(self):
        self._captured_activations.clear()

    @torch.no_grad()
    def encode(self, pixel_values: torch.Tensor) -> torch.Tensor:
        self.vae.eval()
        d = self.vae.encode(pixel_values.to(self.vae.device, dtype=self.vae.dtype)).latent_dist
        return d.sample() * self.scaling_factor

    @torch.no_grad()
    def decode(self, latents: torch.Tensor) -> torch.Tensor:
        self.vae.eval()
        latents = latents / self.scaling_factor
        return self.vae.decode(latents.to(self.vae.device, dtype=self.vae.dtype)).sample.clamp(-1, 1)
