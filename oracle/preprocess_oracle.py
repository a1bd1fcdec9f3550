"""TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else): CPU restatement of the reference's image transform
chain, reference src/data_utils.py:24-30

    transforms.Resize(resolution, BILINEAR) -> CenterCrop(resolution) -> convert("RGB") -> ToTensor() -> Normalize([0.5], [0.5])

torchvision is not installed here, so the chain is restated on what torchvision calls for PIL inputs:
Pillow's resampler (src/libImaging/Resample.c, third-party, Pillow 12.2.0 in this image): separable, horizontal pass
then vertical pass, triangle filter whose support grows with the down-scale factor (antialiasing), coefficients
normalised and quantised to 22 fractional bits, 8-bit intermediate between the passes.  Written in numpy integer
arithmetic; PINNED by tests/test_preprocess.py against Pillow itself (bit-exact uint8) on seeded images, and against the
committed fixture tests/golden/preprocess.npz.  Parity against torchvision proper: unpinned (absent)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resized_size(w: int, h: int, resolution: int):
    """Resize(int): the shorter side becomes `resolution`, the other keeps the aspect ratio (data_utils.py:25;
    this repo's src/data_utils.py:get_transform computes the same).
    Restated from torchvision's published `_compute_resized_output_size` (torchvision is not installed: parity with it is
    unpinned): new_short = resolution, new_long = int(resolution * long / short) -- truncation."""
    if w <= h:
        return resolution, int(resolution * h / w)
    return int(resolution * w / h), resolution


def crop_offset(size: int, resolution: int) -> int:
    """torchvision CenterCrop: crop_top = int(round((image_height - crop_height) / 2.0)) (Python round: half to even)"""
    return int(round((size - resolution) / 2.0))


def coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter.
    -> bounds [out][2] = (xmin, count), kk [out][ksize] int32"""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(xmax, np.float64)
        ww = 0.0
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, bounds, kk, axis: int) -> np.ndarray:
    """one 8-bit resampling pass along `axis` of an [H][W][C] uint8 image (ImagingResampleHorizontal/Vertical_8bpc)"""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for i in range(bounds.shape[0]):
        lo, n = int(bounds[i, 0]), int(bounds[i, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for t in range(n):
            acc += src[lo + t] * int(kk[i, t])
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def transform_u8(img: np.ndarray, resolution: int) -> np.ndarray:
    """[H][W][3] (or [H][W]) uint8 -> resized + centre-cropped [R][R][3] uint8"""
    if img.ndim == 2:
        img = np.repeat(img[:, :, None], 3, axis=2)  # convert("RGB") of an "L" image replicates the channel
    h, w = img.shape[:2]
    nw, nh = resized_size(w, h, resolution)
    if (nw, nh) != (w, h):
        bx, kx = coeffs(w, nw)
        by, ky = coeffs(h, nh)
        img = _pass(_pass(img, bx, kx, 1), by, ky, 0)
    left, top = crop_offset(nw, resolution), crop_offset(nh, resolution)
    return img[top:top + resolution, left:left + resolution]


def transform(img: np.ndarray, resolution: int) -> np.ndarray:
    """-> float32 [3][R][R] in [-1, 1]: ToTensor (x / 255 in fp32) then Normalize ((x - 0.5) / 0.5 in fp32)"""
    a = transform_u8(img, resolution).transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    return (a - np.float32(0.5)) / np.float32(0.5)
