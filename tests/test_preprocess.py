"""Input transform (reference src/data_utils.py:24-30).  CPU: the numpy restatement of Pillow's resampler
(oracle/preprocess_oracle.py) against Pillow itself and the committed fixture, the host-side coefficient tables, and the
raw-batch loader plumbing.  GPU: vae_preprocess_u8 through vaehip.preprocess.GpuPreprocessor, bit-identical to the CPU
chain (integer arithmetic), on the fixture, on live seeded images of many sizes, and end to end through the dataloader."""
import os

import numpy as np
import pytest
import torch

import preprocess_oracle as po

G = os.path.join(os.path.dirname(__file__), "golden", "preprocess.npz")
SIZES = [(30, 40, 16), (375, 500, 64), (20, 20, 32), (97, 61, 32), (64, 64, 64), (256, 341, 256), (33, 100, 48), (513, 300, 128),
         (16, 16, 16), (17, 300, 16)]


def _fixture():
    z = np.load(G)
    n = len([k for k in z.files if k.startswith("in")])
    return [(z[f"in{i}"], int(z[f"res{i}"]), z[f"out{i}"]) for i in range(n)]


def test_oracle_matches_pillow_and_fixture():
    from PIL import Image
    from data_utils import get_transform
    for a, r, want in _fixture():
        assert np.array_equal(po.transform(a, r), want)          # pinned by the committed vectors
    rng = np.random.default_rng(1)
    for h, w, r in SIZES:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(po.transform(a, r), get_transform(r)(Image.fromarray(a)).numpy()), (h, w, r)  # pinned by Pillow
    g = rng.integers(0, 256, (41, 59), dtype=np.uint8)
    assert np.array_equal(po.transform(g, 24), get_transform(24)(Image.fromarray(g)).numpy())


def test_host_coefficient_tables_match_the_restatement():
    from vaehip.preprocess import bilinear_coeffs, resized_size
    for i, o in [(40, 21), (500, 85), (20, 32), (97, 51), (64, 64), (341, 341), (100, 145), (3000, 256), (17, 1000)]:
        b, k = po.coeffs(i, o)
        b2, k2 = bilinear_coeffs(i, o, 0, o)
        assert np.array_equal(b, b2) and np.array_equal(k, k2), (i, o)
        lo, n = o // 3, max(1, o // 2)
        b3, k3 = bilinear_coeffs(i, o, lo, n)   # only the crop window
        assert np.array_equal(b3, b[lo:lo + n]) and np.array_equal(k3, k[lo:lo + n])
        assert int(k.sum(axis=1).min()) >= (1 << 22) - k.shape[1] and int(k.sum(axis=1).max()) <= (1 << 22) + k.shape[1]
    assert resized_size(500, 375, 64) == po.resized_size(500, 375, 64) == (85, 64)
    # torchvision's geometry (reference src/data_utils.py:25-26): Resize(int) TRUNCATES the long side, CenterCrop rounds the half
    # margin half-to-even -- known answers worked by hand from torchvision's published formulas (parity with torchvision itself
    # is unpinned: not installed)
    from vaehip.preprocess import crop_offset
    assert resized_size(640, 427, 256) == po.resized_size(640, 427, 256) == (383, 256)   # 383.7 -> 383, not 384
    assert resized_size(30, 41, 16) == po.resized_size(30, 41, 16) == (16, 21)           # portrait: 21.87 -> 21
    assert [crop_offset(n, 16) for n in (16, 17, 19, 21, 23)] == [po.crop_offset(n, 16) for n in (16, 17, 19, 21, 23)] == [0, 0, 2, 2, 4]


def test_raw_batch_loader_plumbing(tmp_path):
    """workers only decode; mixed sizes / modes; a failed item is dropped; order is kept"""
    from PIL import Image
    from data_utils import create_dataloader, get_transform, load_and_preprocess_dataset
    d = tmp_path / "imgs" / "train"
    d.mkdir(parents=True)
    rng = np.random.default_rng(2)
    Image.fromarray(rng.integers(0, 256, (30, 40, 3), dtype=np.uint8)).save(d / "a.png")
    Image.fromarray(rng.integers(0, 256, (20, 20), dtype=np.uint8)).save(d / "b.png")              # grey
    Image.fromarray(rng.integers(0, 256, (25, 35, 4), dtype=np.uint8), "RGBA").save(d / "c.png")  # alpha: CPU chain
    (d / "d.png").write_bytes(b"not an image")
    ds = load_and_preprocess_dataset(str(tmp_path / "imgs"), resolution=16, split="train", gpu_preprocess=True)
    assert ds[0]["pixel_u8"].shape == (30, 40, 3) and ds[1]["pixel_u8"].shape == (20, 20) and ds[2]["pixel_values"].shape == (3, 16, 16)

    class CpuStandIn:  # the kernel's arithmetic, restated (tests only)
        res, device = 16, torch.device("cpu")

        def __call__(self, items):
            return torch.stack([torch.from_numpy(po.transform(t.numpy(), 16)) for t in items])
    got = next(iter(create_dataloader(ds, 4, shuffle=False, gpu_preprocess=CpuStandIn())))["pixel_values"]
    ref = next(iter(create_dataloader(load_and_preprocess_dataset(str(tmp_path / "imgs"), resolution=16, split="train"), 4, shuffle=False)))
    assert got.shape == (3, 3, 16, 16) and torch.equal(got, ref["pixel_values"])


@pytest.mark.gpu
def test_gpu_transform_is_bit_identical_to_the_cpu_chain(cuda):
    from PIL import Image
    from data_utils import get_transform
    from vaehip.preprocess import GpuPreprocessor
    for a, r, want in _fixture():
        got = GpuPreprocessor(r, cuda)([a])[0].cpu().numpy()
        assert np.array_equal(got, want), (a.shape, r)
    rng = np.random.default_rng(3)
    for r in (16, 64, 256):
        imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w, _ in SIZES if min(h, w) * 8 >= r]
        imgs += [rng.integers(0, 256, (77, 50), dtype=np.uint8), imgs[0].copy(), imgs[0][::-1].copy()]  # grey; two of one size
        got = GpuPreprocessor(r, cuda)(imgs).cpu()
        for i, a in enumerate(imgs):
            ref = get_transform(r)(Image.fromarray(a))
            assert torch.equal(got[i], ref), (r, a.shape)
    assert float(got.min()) >= -1.0 and float(got.max()) <= 1.0


@pytest.mark.gpu
def test_gpu_dataloader_equals_cpu_dataloader(cuda, tmp_path):
    from PIL import Image
    from data_utils import create_dataloader, load_and_preprocess_dataset
    from vaehip.preprocess import GpuPreprocessor
    d = tmp_path / "imgs" / "train"
    d.mkdir(parents=True)
    rng = np.random.default_rng(4)
    for i, (h, w) in enumerate([(300, 400), (256, 256), (300, 400), (500, 333), (90, 120)]):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(d / f"{i}.png")
    Image.fromarray(rng.integers(0, 256, (64, 80), dtype=np.uint8)).save(d / "g.png")
    root = str(tmp_path / "imgs")
    cpu = [b["pixel_values"] for b in create_dataloader(load_and_preprocess_dataset(root, resolution=64, split="train"), 4, shuffle=False)]
    gpu = [b["pixel_values"] for b in create_dataloader(load_and_preprocess_dataset(root, resolution=64, split="train", gpu_preprocess=True),
                                                        4, shuffle=False, gpu_preprocess=GpuPreprocessor(64, cuda))]
    assert len(cpu) == len(gpu) == 2
    for a, b in zip(cpu, gpu):
        assert b.is_cuda and torch.equal(a, b.cpu())
