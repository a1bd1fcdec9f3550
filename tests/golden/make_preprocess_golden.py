"""Generates tests/golden/preprocess.npz: seeded uint8 images and what the reference's transform chain
(reference src/data_utils.py:24-30: Resize(bilinear) -> CenterCrop -> RGB -> ToTensor -> Normalize(0.5, 0.5)) makes of
them.  torchvision is not installed in this image, so the chain is evaluated with Pillow directly -- the library
torchvision's Resize / CenterCrop call for PIL inputs -- through this repo's data_utils.get_transform (Pillow 12.2.0).
PARITY STATUS: Pillow's 8-bit resampler IS the arithmetic torchvision runs for PIL inputs, so the pixel values are pinned
by Pillow itself.  The Resize / CenterCrop GEOMETRY (truncated long side, half-to-even crop margin) comes from this repo's
own data_utils, restated from torchvision's published formulas and hand-checked on the cases marked below: against
torchvision itself it is **parity unpinned** (torchvision is not importable here and the reference holds no fixture).
Run from the repo root:  python tests/golden/make_preprocess_golden.py"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-channel-dynamics_amd", "src"))
from data_utils import get_transform  # noqa: E402

CASES = [  # (height, width, channels, resolution)
    (30, 40, 3, 16), (375, 500, 3, 64), (20, 20, 3, 32), (97, 61, 3, 32), (64, 64, 3, 64), (50, 70, 1, 32), (33, 100, 3, 48),
    (427, 640, 3, 256),  # long side 383.7 -> 383 (torchvision truncates), margin 127 -> offset 64 (63.5 rounds to even)
    (41, 30, 3, 16),     # portrait: 16 x 21 (21.87 truncated), margin 5 -> offset 2 (2.5 rounds to even, not 3)
]


def main():
    rng = np.random.default_rng(20240601)
    out = {}
    for i, (h, w, c, r) in enumerate(CASES):
        a = rng.integers(0, 256, (h, w, c) if c == 3 else (h, w), dtype=np.uint8)
        # smooth half of the cases a little (real photographs are not white noise): running mean along x
        if i % 2 == 1:
            a = (np.cumsum(a.astype(np.int64), axis=1) // (np.arange(a.shape[1]) + 1).reshape((1, -1) + (1,) * (a.ndim - 2))).astype(np.uint8)
        out[f"in{i}"] = a
        out[f"res{i}"] = np.int32(r)
        out[f"out{i}"] = get_transform(r)(Image.fromarray(a)).numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "preprocess.npz"), **out)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
