"""Input pipeline with the reference's function surface (reference src/data_utils.py:13-225):
get_transform / load_and_preprocess_dataset / create_dataloader, items are {"pixel_values": [3,R,R] in [-1,1]}.

The reference needs `torchvision` + a Hub download.  Here:
  dataset_name "synthetic[:N]"      deterministic uniform [-1,1) images (benchmarks, plumbing, tests)
  dataset_name = local directory    image files below it (optionally <dir>/<split>/...), PIL decode,
                                    Resize(shorter side, bilinear) -> CenterCrop -> RGB -> [0,1] -> (x-0.5)/0.5
                                    -- the same transform chain as data_utils.py:24-30 without torchvision
  anything else                     handed to `datasets.load_dataset` (works only with a populated HF cache)

`data.gpu_preprocess: true` (a key the reference does not have) moves Resize / CenterCrop / ToTensor / Normalize to the
GPU: workers only decode, batches of raw uint8 images go to the device and `vaehip.preprocess.GpuPreprocessor` produces
the same tensors bit for bit (the per-item CPU resize is what limits an 8-GPU node at >= 200 images/s per GPU).
"""
import logging
import os
from typing import Any, Dict, List, Optional

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

logger = logging.getLogger(__name__)
IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp", ".tif", ".tiff")


def get_transform(resolution: int):
    from PIL import Image

    def tf(img) -> torch.Tensor:
        w, h = img.size
        # torchvision's Resize(int) truncates the long side; CenterCrop rounds the half margin (half to even)
        short, long = (w, h) if w <= h else (h, w)
        new_long = int(resolution * long / short)
        nw, nh = (resolution, new_long) if w <= h else (new_long, resolution)
        if (nw, nh) != (w, h):
            img = img.resize((nw, nh), Image.BILINEAR)
        left, top = int(round((nw - resolution) / 2.0)), int(round((nh - resolution) / 2.0))
        img = img.crop((left, top, left + resolution, top + resolution))
        if img.mode != "RGB":
            img = img.convert("RGB")
        a = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
        return a.sub_(0.5).div_(0.5)
    return tf


class SyntheticImageDataset(Dataset):
    """uniform [-1,1) images, a pure function of (seed, index): identical on every rank and run."""

    def __init__(self, n: int, resolution: int, seed: int = 42):
        self.n, self.res, self.seed = int(n), int(resolution), int(seed)

    def __len__(self):
        return self.n

    def __getitem__(self, i) -> Dict[str, Any]:
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(i))
        return {"pixel_values": torch.rand((3, self.res, self.res), generator=g) * 2 - 1}


def raw_item(img, tf) -> Dict[str, Any]:
    """item for the GPU transform: the decoded bytes ("RGB" / "L" images, which the kernel resamples like Pillow); any
    other mode (palette, alpha, CMYK ...) goes through the CPU chain, whose mode-specific resampling is not restated"""
    if img.mode in ("RGB", "L"):
        return {"pixel_u8": torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())}
    return {"pixel_values": tf(img)}


class ImageFolderDataset(Dataset):
    def __init__(self, root: str, resolution: int, max_samples: Optional[int] = None, raw: bool = False):
        self.raw = bool(raw)
        files: List[str] = []
        for d, _, fs in sorted(os.walk(root)):
            files += [os.path.join(d, f) for f in sorted(fs) if f.lower().endswith(IMG_EXT)]
        if not files:
            raise ValueError(f"no image files under {root}")
        self.files = files[:max_samples] if max_samples else files
        self.tf = get_transform(resolution)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i) -> Dict[str, Any]:
        from PIL import Image
        try:
            with Image.open(self.files[i]) as im:
                return raw_item(im, self.tf) if self.raw else {"pixel_values": self.tf(im)}
        except Exception as e:  # mirrors transform_images' per-item error handling (data_utils.py:121-152)
            logger.error(f"Failed to load {self.files[i]}: {e}")
            return {"pixel_values": None}


def load_and_preprocess_dataset(dataset_name: str, dataset_config_name: Optional[str] = None, image_column: str = "image",
                                resolution: int = 256, max_samples: Optional[int] = None, split: str = "train",
                                streaming: bool = False, cache_dir: Optional[str] = None, gpu_preprocess: bool = False):
    if dataset_name is None:
        raise ValueError("data.dataset_name is required")
    if dataset_name.startswith("synthetic"):
        n = int(dataset_name.split(":", 1)[1]) if ":" in dataset_name else 1024
        if max_samples:
            n = min(n, int(max_samples))
        seed = 42 if split == "train" else 4242
        logger.info(f"Synthetic dataset: {n} images at {resolution}x{resolution} (split {split})")
        return SyntheticImageDataset(n, resolution, seed)
    if os.path.isdir(dataset_name):
        root = os.path.join(dataset_name, split) if os.path.isdir(os.path.join(dataset_name, split)) else dataset_name
        logger.info(f"Local image folder dataset: {root}")
        return ImageFolderDataset(root, resolution, max_samples, raw=gpu_preprocess)
    from datasets import load_dataset  # needs a populated local HF cache: there is no network in this build
    logger.info(f"Loading dataset '{dataset_name}' (config: {dataset_config_name}, split: {split}) via datasets")
    ds = load_dataset(dataset_name, dataset_config_name, split=split, streaming=streaming, cache_dir=cache_dir)
    if image_column not in ds.column_names:
        raise ValueError(f"Image column '{image_column}' not found. Available: {ds.column_names}")
    if max_samples is not None and not streaming:
        ds = ds.select(range(min(int(max_samples), len(ds))))
    tf = get_transform(resolution)

    def transform_images(examples):
        out = []
        for im in examples[image_column]:
            try:
                out.append(tf(im))
            except Exception as e:
                logger.error(f"transform failed: {e}")
                out.append(None)
        return {"pixel_values": out}

    def raw_images(examples):
        items = []
        for im in examples[image_column]:
            try:
                items.append(raw_item(im, tf))
            except Exception as e:
                logger.error(f"decode failed: {e}")
                items.append({})
        return {"pixel_u8": [it.get("pixel_u8") for it in items], "pixel_values": [it.get("pixel_values") for it in items]}
    return ds.with_transform(raw_images if gpu_preprocess else transform_images)


def safe_collate(batch):
    """drops items whose transform failed; returns {"pixel_values": tensor} or None (data_utils.py:197-215)."""
    good = [b["pixel_values"] for b in batch if b.get("pixel_values") is not None and len(b["pixel_values"]) > 0]
    if len(good) < len(batch):
        logger.warning(f"Collate function filtered {len(batch) - len(good)} items due to missing/empty 'pixel_values'.")
    if not good:
        return None
    try:
        return {"pixel_values": torch.utils.data.default_collate(good)}
    except Exception as e:
        logger.error(f"Error during collate_fn: {e}. Skipping batch.")
        return None


def raw_collate(batch):
    """collate for the GPU transform: images differ in size, so the batch is a list (uint8 [H][W](x3) tensors, or fp32
    [3][R][R] tensors of items the CPU chain handled); failed items are dropped like safe_collate does"""
    items = []
    for b in batch:
        t = b.get("pixel_u8")
        if t is None:
            t = b.get("pixel_values")
        if t is not None and len(t) > 0:
            items.append(t)
    if len(items) < len(batch):
        logger.warning(f"Collate function filtered {len(batch) - len(items)} items due to missing/empty images.")
    return {"items": items} if items else None


class GpuTransformLoader:
    """wraps a DataLoader of raw batches: yields {"pixel_values": fp32 CUDA tensor [N][3][R][R]} like the CPU pipeline"""

    def __init__(self, loader: DataLoader, preprocessor):
        self.loader, self.pre = loader, preprocessor
        self.sampler, self.dataset, self.batch_size = loader.sampler, loader.dataset, loader.batch_size

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for batch in self.loader:
            if not batch:
                yield None
                continue
            items = batch["items"]
            u8 = [i for i, t in enumerate(items) if t.dtype == torch.uint8]
            if len(u8) == len(items):
                yield {"pixel_values": self.pre(items)}
                continue
            out = torch.empty((len(items), 3, self.pre.res, self.pre.res), device=self.pre.device, dtype=torch.float32)
            if u8:
                out[torch.tensor(u8, device=out.device)] = self.pre([items[i] for i in u8])
            for i, t in enumerate(items):
                if t.dtype != torch.uint8:
                    out[i] = t.to(out.device, dtype=torch.float32)
            yield {"pixel_values": out}


def create_dataloader(dataset, batch_size: int, num_workers: int = 0, shuffle: bool = True, pin_memory: bool = True,
                      collate_fn=None, rank: int = 0, world_size: int = 1, seed: int = 42, gpu_preprocess=None):
    """per-rank shard when world_size > 1 (what accelerate.prepare does to the reference's dataloader, train.py:205-210).
    gpu_preprocess: a vaehip.preprocess.GpuPreprocessor for datasets loaded with gpu_preprocess=True."""
    if gpu_preprocess is not None and not isinstance(dataset, SyntheticImageDataset):
        inner = create_dataloader(dataset, batch_size, num_workers, shuffle, pin_memory, raw_collate, rank, world_size, seed)
        return GpuTransformLoader(inner, gpu_preprocess)
    is_iterable = isinstance(dataset, torch.utils.data.IterableDataset)
    sampler = None
    if world_size > 1 and not is_iterable:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world_size, rank=rank,
                                                                  shuffle=shuffle, seed=seed, drop_last=False)
    return DataLoader(dataset, batch_size=batch_size, shuffle=(shuffle and not is_iterable and sampler is None),
                      sampler=sampler, num_workers=num_workers, pin_memory=pin_memory and torch.cuda.is_available(),
                      collate_fn=collate_fn or safe_collate)
