"""GPU-side input transform: Resize(bilinear) -> CenterCrop -> RGB -> ToTensor -> Normalize(0.5, 0.5) of the reference
(src/data_utils.py:24-30) on uint8 image batches, through the C ABI (vae_preprocess_u8, csrc/preprocess.hip).

The host only decodes (PIL) and uploads the bytes; the coefficient tables of Pillow's resampler depend on the source /
target sizes alone and are cached per size.  The arithmetic is Pillow's 8-bit fixed point, so the tensor equals what
`data_utils.get_transform` produces on the CPU bit for bit (tests/test_preprocess.py)."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from .lib import lib

PRECISION_BITS = 32 - 8 - 2  # Pillow Resample.c


def resized_size(w: int, h: int, resolution: int) -> Tuple[int, int]:
    """torchvision's Resize(int) (reference src/data_utils.py:25): the shorter side becomes `resolution`, the longer one
    int(resolution * long / short) -- TRUNCATED, not rounded (640x427 -> 383x256).  -> (new_w, new_h)"""
    short, long = (w, h) if w <= h else (h, w)
    new_long = int(resolution * long / short)
    return (resolution, new_long) if w <= h else (new_long, resolution)


def crop_offset(size: int, resolution: int) -> int:
    """torchvision's CenterCrop (reference src/data_utils.py:26): int(round((size - crop) / 2.0)) with Python's
    round-half-to-even (margin 1 -> 0, margin 3 -> 2, margin 5 -> 2)"""
    return int(round((size - resolution) / 2.0))


def bilinear_coeffs(in_size: int, out_size: int, first: int, count: int):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (triangle filter, support 1 x max(scale, 1)) for output
    positions first .. first+count-1 of a resize in_size -> out_size: (bounds [count][2] int32, kk [count][ksize] int32)"""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(first, first + count, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    t = np.arange(ksize, dtype=np.float64)[None, :]
    arg = np.abs((t + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(arg < 1.0, 1.0 - arg, 0.0)
    w = np.where(t < xmax[:, None], w, 0.0)
    # the reference loop sums the weights left to right in double precision; cumulative sum keeps that order
    ww = np.cumsum(w, axis=1)[:, -1:]
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    kk = (0.5 + w * float(1 << PRECISION_BITS)).astype(np.int64).astype(np.int32)  # weights are >= 0 for this filter
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, kk


class GpuPreprocessor:
    """callable: list of uint8 images ([H][W][3] RGB or [H][W] grey; numpy or torch, any sizes) -> fp32 CUDA tensor
    [N][3][R][R] in [-1, 1].  Images of equal size go through one launch pair."""

    def __init__(self, resolution: int, device):
        self.res = int(resolution)
        self.device = torch.device(device)
        self._tables: Dict[Tuple[int, int], tuple] = {}

    def _plan(self, h: int, w: int):
        key = (h, w)
        if key not in self._tables:
            R = self.res
            nw, nh = resized_size(w, h, R)
            left, top = crop_offset(nw, R), crop_offset(nh, R)
            bx, kx = bilinear_coeffs(w, nw, left, R)
            by, ky = bilinear_coeffs(h, nh, top, R)
            row0 = int(by[:, 0].min())
            nrows = int((by[:, 0] + by[:, 1]).max()) - row0
            dev = self.device
            self._tables[key] = (torch.from_numpy(bx).to(dev), torch.from_numpy(kx).to(dev), kx.shape[1],
                                 torch.from_numpy(by).to(dev), torch.from_numpy(ky).to(dev), ky.shape[1], row0, nrows)
        return self._tables[key]

    def __call__(self, images: Sequence) -> torch.Tensor:
        R = self.res
        out = torch.empty((len(images), 3, R, R), device=self.device, dtype=torch.float32)
        groups: Dict[Tuple[int, int, int], List[int]] = {}
        arrs = []
        for i, im in enumerate(images):
            a = im if isinstance(im, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(im))
            if a.dtype != torch.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] not in (1, 3)):
                raise ValueError(f"GpuPreprocessor: image {i}: expected uint8 [H][W] or [H][W][3], got {a.dtype} {tuple(a.shape)}")
            c = 1 if a.ndim == 2 or a.shape[2] == 1 else 3
            arrs.append(a.contiguous())
            groups.setdefault((a.shape[0], a.shape[1], c), []).append(i)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        for (h, w, c), idx in groups.items():
            bx, kx, ksx, by, ky, ksy, row0, nrows = self._plan(h, w)
            src = torch.stack([arrs[i].reshape(h, w, c) for i in idx]).to(self.device, non_blocking=True)
            n = len(idx)
            tmp = torch.empty((n, nrows, R, 3), device=self.device, dtype=torch.uint8)
            dst = out if n == len(images) else torch.empty((n, 3, R, R), device=self.device, dtype=torch.float32)
            lib.call("vae_preprocess_u8", C.c_void_p(src.data_ptr()), n, h, w, c, R, C.c_void_p(bx.data_ptr()),
                     C.c_void_p(kx.data_ptr()), ksx, C.c_void_p(by.data_ptr()), C.c_void_p(ky.data_ptr()), ksy, row0, nrows,
                     C.c_void_p(tmp.data_ptr()), C.c_void_p(dst.data_ptr()), stream)
            if dst is not out:
                out[torch.tensor(idx, device=self.device)] = dst
        return out
