"""Evaluation of a trained VAE checkpoint (reference src/evaluate.py:78-328): same CLI flags, loads
`<checkpoint_path>/vae`, deterministic reconstruction (latent mode), Average MSE / KL / PSNR / SSIM,
sample PNGs and `eval_metrics.txt` in the reference's format.

The forward runs on the HIP engine.  PSNR (data_range 1.0 on [0,1]-clamped images) and SSIM (11x11
gaussian, sigma 1.5) are computed here directly (the reference needs torchmetrics); they are
evaluation-only torch ops, not part of the train-step hot path.  Logit-lens visualisation is out of
scope; `--enable_logit_lens` is accepted and captures the requested layers' activations through the
hook protocol (as evaluate.py:207-211 does) so downstream tooling can use them.
"""
import argparse
import logging
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

from utils.config_utils import load_config
from utils.logging_utils import setup_logging
from data_utils import load_and_preprocess_dataset, create_dataloader
from models.sdxl_vae_wrapper import SDXLVAEWrapper

setup_logging()
logger = logging.getLogger(__name__)


def parse_args():
    p = argparse.ArgumentParser(description="Evaluate a trained SDXL VAE model.")
    p.add_argument("--config_path", type=str, required=True)
    p.add_argument("--checkpoint_path", type=str, required=True)
    p.add_argument("--eval_split", type=str, default="test")
    p.add_argument("--output_dir", type=str, default=None)
    p.add_argument("--num_samples_to_save", type=int, default=16)
    p.add_argument("--batch_size", type=int, default=None)
    p.add_argument("--enable_logit_lens", default=True, type=lambda x: (str(x).lower() == "true"))
    p.add_argument("--logit_lens_layers", type=str, nargs="+",
                   default=["encoder.down_blocks.0.resnets.0.norm1", "encoder.down_blocks.1.resnets.0.conv_shortcut"])
    p.add_argument("--logit_lens_num_samples", type=int, default=1)
    p.add_argument("--logit_lens_projection_type", type=str, default="mini_decoder_single_channel",
                   choices=["mini_decoder_single_channel", "mini_decoder_full_map"])
    p.add_argument("--logit_lens_mini_decoder_input_channels", type=int, default=None)
    return p.parse_args()


def to_unit(t: torch.Tensor) -> torch.Tensor:
    return torch.clamp((t + 1.0) / 2.0, 0.0, 1.0)


def psnr_sums(pred: torch.Tensor, target: torch.Tensor):
    """torchmetrics PeakSignalNoiseRatio(data_range=1.0) accumulates sum of squared error and count."""
    return torch.sum((pred - target) ** 2).double(), pred.numel()


def ssim_per_image(pred: torch.Tensor, target: torch.Tensor, sigma: float = 1.5, ksize: int = 11) -> torch.Tensor:
    """SSIM with a gaussian window (data_range 1.0, k1=0.01, k2=0.03), reflect-padded and border-cropped like torchmetrics.
    The window statistics are taken in float64: var = E[x^2] - mu^2 cancels catastrophically in fp32 on flat regions
    (errors of 1e-4 in the index against the 9e-4 stabiliser c2); torchmetrics itself works in the input dtype."""
    out_dtype = pred.dtype
    pred, target = pred.double(), target.double()
    c = pred.shape[1]
    ax = torch.arange(ksize, dtype=pred.dtype, device=pred.device) - (ksize - 1) / 2
    g = torch.exp(-(ax / sigma) ** 2 / 2)
    g = (g / g.sum()).unsqueeze(0)
    win = (g.t() @ g).expand(c, 1, ksize, ksize).contiguous()
    pad = (ksize - 1) // 2
    p = F.pad(pred, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat([p, t, p * p, t * t, p * t])
    out = F.conv2d(stack, win, groups=c)
    mu_p, mu_t, e_pp, e_tt, e_pt = out.split(pred.shape[0])
    s_pp, s_tt, s_pt = e_pp - mu_p ** 2, e_tt - mu_t ** 2, e_pt - mu_p * mu_t
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    ssim = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p ** 2 + mu_t ** 2 + c1) * (s_pp + s_tt + c2))
    return ssim[..., pad:-pad, pad:-pad].reshape(pred.shape[0], -1).mean(-1).to(out_dtype)


def save_png(t: torch.Tensor, path: str):
    from PIL import Image
    a = (to_unit(t.float().cpu()) * 255.0).round().byte().permute(1, 2, 0).numpy()
    Image.fromarray(a).save(path)


def main():
    args = parse_args()
    config = load_config(args.config_path)
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: evaluation runs on the HIP engine (no CPU fallback)")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    if args.output_dir is None:
        args.output_dir = os.path.join(args.checkpoint_path, f"eval_results_{args.eval_split}")
    os.makedirs(args.output_dir, exist_ok=True)
    model_path = os.path.join(args.checkpoint_path, "vae")
    if not os.path.isdir(model_path):
        logger.error(f"VAE model directory not found at: {model_path}")
        sys.exit(1)
    w = SDXLVAEWrapper(pretrained_model_name_or_path=model_path, device=device)
    w.vae.eval()
    data_cfg = config.get("data", {})
    bs = args.batch_size or data_cfg.get("validation_batch_size", data_cfg.get("batch_size", 4))
    ds = load_and_preprocess_dataset(
        dataset_name=data_cfg.get("validation_dataset_name", data_cfg.get("dataset_name")),
        dataset_config_name=data_cfg.get("validation_dataset_config_name", data_cfg.get("dataset_config_name", None)),
        image_column=data_cfg.get("image_column", "image"), resolution=data_cfg.get("resolution", 256),
        max_samples=data_cfg.get("validation_max_samples", None), split=args.eval_split)
    dl = create_dataloader(ds, batch_size=bs, num_workers=data_cfg.get("num_workers", 0), shuffle=False)

    total_mse = total_kl = 0.0
    n = saved = 0
    sse = torch.zeros((), dtype=torch.float64, device=device)
    sse_count = 0
    ssim_sum = torch.zeros((), dtype=torch.float64, device=device)
    with torch.no_grad():
        for step, batch in enumerate(dl):
            if step == 0 and args.enable_logit_lens:
                w.add_hooks(args.logit_lens_layers)
            pv = batch.get("pixel_values") if batch else None
            if pv is None:
                continue
            pv = pv.to(device, dtype=torch.float32)
            out = w(pv, sample_posterior=False)
            rec = out["reconstruction"]
            kl = out["latent_dist"].kl()
            b = pv.shape[0]
            total_mse += F.mse_loss(rec.float(), pv.float(), reduction="mean").item() * b
            total_kl += kl.mean().item() * b
            n += b
            r01, o01 = to_unit(rec).contiguous(), to_unit(pv)
            s, c = psnr_sums(r01, o01)
            sse += s
            sse_count += c
            ssim_sum += ssim_per_image(r01, o01).double().sum()
            while saved < args.num_samples_to_save and saved - (n - b) < b:
                i = saved - (n - b)
                save_png(pv[i], os.path.join(args.output_dir, f"sample_{saved}_orig.png"))
                save_png(rec[i], os.path.join(args.output_dir, f"sample_{saved}_recon.png"))
                saved += 1
            if step == 0 and args.enable_logit_lens:
                acts = w.get_captured_activations()
                torch.save({k: v for k, v in acts.items()}, os.path.join(args.output_dir, "first_batch_activations.pt"))
                w.remove_hooks()
    avg_mse = total_mse / n if n else 0
    avg_kl = total_kl / n if n else 0
    psnr = float(10.0 * torch.log10(1.0 / (sse / max(sse_count, 1)))) if n else float("nan")
    ssim = float(ssim_sum / n) if n else float("nan")
    logger.info("***** Evaluation Results *****")
    logger.info(f"  Dataset split: {args.eval_split}; samples: {n}")
    logger.info(f"  Average MSE Loss: {avg_mse:.6f}  Average KL Divergence: {avg_kl:.6f}  PSNR: {psnr:.4f} dB  SSIM: {ssim:.4f}")
    with open(os.path.join(args.output_dir, "eval_metrics.txt"), "w") as f:
        f.write(f"Evaluation Split: {args.eval_split}\n")
        f.write(f"Checkpoint Path: {args.checkpoint_path}\n")
        f.write(f"Number of Samples Processed: {n}\n")
        f.write(f"Average MSE: {avg_mse}\n")
        f.write(f"Average KL: {avg_kl}\n")
        f.write(f"Average PSNR: {psnr}\n")
        f.write(f"Average SSIM: {ssim}\n")


if __name__ == "__main__":
    try:
        main()
    except Exception as e:
        logging.getLogger(__name__).error(f"Unhandled exception in main: {e}", exc_info=True)
        sys.exit(1)
