"""Input pipeline with the reference's function surface (reference src/data_utils.py:13-225):
get_transform / load_and_preprocess_dataset / create_dataloader, items are {"pixel_values": [3,R,R] in [-1,1]}.

The reference needs `torchvision` + a Hub download.  Here:
  dataset_name "synthetic[:N]"      deterministic uniform [-1,1) images (benchmarks, plumbing, tests)
  dataset_name = local directory    image files below it (optionally <dir>/<split>/...), PIL decode,
                                    Resize(shorter side, bilinear) -> CenterCrop -> RGB -> [0,1] -> (x-0.5)/0.5
                                    -- the same transform chain as data_utils.py:24-30 without torchvision
  anything else                     handed to `datasets.load_dataset` (works only with a populated HF cache)
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
        s = resolution / min(w, h)
        nw, nh = max(resolution, int(round(w * s))), max(resolution, int(round(h * s)))
        if (nw, nh) != (w, h):
            img = img.resize((nw, nh), Image.BILINEAR)
        left, top = (nw - resolution) // 2, (nh - resolution) // 2
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


class ImageFolderDataset(Dataset):
    def __init__(self, root: str, resolution: int, max_samples: Optional[int] = None):
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
                return {"pixel_values": self.tf(im)}
        except Exception as e:  # mirrors transform_images' per-item error handling (data_utils.py:121-152)
            logger.error(f"Failed to load {self.files[i]}: {e}")
            return {"pixel_values": None}


def load_and_preprocess_dataset(dataset_name: str, dataset_config_name: Optional[str] = None, image_column: str = "image",
                                resolution: int = 256, max_samples: Optional[int] = None, split: str = "train",
                                streaming: bool = False, cache_dir: Optional[str] = None):
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
        return ImageFolderDataset(root, resolution, max_samples)
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
    return ds.with_transform(transform_images)


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


def create_dataloader(dataset, batch_size: int, num_workers: int = 0, shuffle: bool = True, pin_memory: bool = True,
                      collate_fn=None, rank: int = 0, world_size: int = 1, seed: int = 42) -> DataLoader:
    """per-rank shard when world_size > 1 (what accelerate.prepare does to the reference's dataloader, train.py:205-210)."""
    is_iterable = isinstance(dataset, torch.utils.data.IterableDataset)
    sampler = None
    if world_size > 1 and not is_iterable:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world_size, rank=rank,
                                                                  shuffle=shuffle, seed=seed, drop_last=False)
    return DataLoader(dataset, batch_size=batch_size, shuffle=(shuffle and not is_iterable and sampler is None),
                      sampler=sampler, num_workers=num_workers, pin_memory=pin_memory and torch.cuda.is_available(),
                      collate_fn=collate_fn or safe_collate)
