"""evaluate.py's PSNR / SSIM (reference src/evaluate.py:172-183 uses torchmetrics PeakSignalNoiseRatio(data_range=1.0)
and StructuralSimilarityIndexMeasure(data_range=1.0, gaussian_kernel=True, sigma=1.5, kernel_size=11)).  torchmetrics is not
installed here and the reference holds no fixture for these numbers: **parity unpinned vs torchmetrics**.  What is checked is
the published definition, through closed forms derived independently of the implementation (1-D window sums in float64):
identical images, a constant offset, and a two-level image against a flat one."""
import math

import numpy as np
import pytest
import torch

from evaluate import psnr_sums, ssim_per_image, to_unit

C1, C2 = 0.01 ** 2, 0.03 ** 2


def _window(ksize=11, sigma=1.5):
    ax = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2
    g = np.exp(-(ax / sigma) ** 2 / 2)
    return g / g.sum()


def _psnr(pred, target):
    s, c = psnr_sums(pred, target)
    return 10.0 * math.log10(1.0 / (float(s) / c))


def test_identical_images():
    torch.manual_seed(0)
    x = torch.rand(3, 3, 40, 48)
    assert torch.allclose(ssim_per_image(x, x), torch.ones(3), atol=1e-6)
    s, c = psnr_sums(x, x)
    assert float(s) == 0.0 and c == x.numel()


@pytest.mark.parametrize("a,delta", [(0.25, 0.1), (0.6, -0.05), (0.0, 0.3)])
def test_constant_offset_closed_forms(a, delta):
    t = torch.full((2, 3, 32, 32), a)
    p = t + delta
    assert _psnr(p, t) == pytest.approx(10 * math.log10(1 / delta ** 2), rel=1e-5)
    want = (2 * a * (a + delta) + C1) / (a * a + (a + delta) ** 2 + C1)  # both variances and the covariance vanish
    assert torch.allclose(ssim_per_image(p, t), torch.full((2,), want), atol=2e-6)
    # a textured target with the same offset: sigma_p = sigma_t = cov, so SSIM = luminance term, evaluated per window
    torch.manual_seed(1)
    tt = torch.rand(1, 1, 24, 24) * 0.5 + 0.2
    g = _window()
    w2 = np.outer(g, g)
    x = tt[0, 0].double().numpy()
    pad = np.pad(x, 5, mode="reflect")
    lum, n = 0.0, 0
    for i in range(5, 24 - 5):        # torchmetrics crops the 5 border pixels after the reflect-padded convolution
        for j in range(5, 24 - 5):
            mu = float((pad[i:i + 11, j:j + 11] * w2).sum())
            lum += (2 * mu * (mu + delta) + C1) / (mu * mu + (mu + delta) ** 2 + C1)
            n += 1
    assert float(ssim_per_image(tt + delta, tt)) == pytest.approx(lum / n, abs=2e-6)


def test_two_level_image_against_flat_image():
    """target: left half a, right half b (constant along y); prediction: flat c.  The vertical direction of the window
    integrates out, so mu_t(x) and E[t^2](x) are 1-D sums of the gaussian window: an independent evaluation."""
    a, b, c, H, W = 0.2, 0.8, 0.5, 32, 32
    t = torch.empty(1, 1, H, W)
    t[..., : W // 2] = a
    t[..., W // 2:] = b
    p = torch.full_like(t, c)
    g = _window()
    row = np.where(np.arange(W) < W // 2, a, b).astype(np.float64)
    rp = np.pad(row, 5, mode="reflect")
    vals = []
    for x in range(5, W - 5):
        seg = rp[x:x + 11]
        mu = float((seg * g).sum())
        var = float((seg * seg * g).sum()) - mu * mu
        vals.append(((2 * mu * c + C1) * C2) / ((mu * mu + c * c + C1) * (var + C2)))  # sigma_p = cov = 0
    want = float(np.mean(vals))  # every row of the cropped map is the same
    assert float(ssim_per_image(p, t)) == pytest.approx(want, abs=2e-6)
    assert 0 < want < 1
    mse = ((a - c) ** 2 + (b - c) ** 2) / 2
    assert _psnr(p, t) == pytest.approx(10 * math.log10(1 / mse), rel=1e-5)


def test_unit_range_mapping_and_accumulation():
    """evaluate.py:196-198: [-1,1] -> clamp((x+1)/2, 0, 1); PSNR accumulates squared error and element counts over
    batches (torchmetrics' update/compute), SSIM averages per image."""
    x = torch.tensor([-3.0, -1.0, 0.0, 0.5, 1.0, 2.0])
    assert torch.equal(to_unit(x), torch.tensor([0.0, 0.0, 0.5, 0.75, 1.0, 1.0]))
    torch.manual_seed(2)
    t = torch.rand(4, 3, 20, 20)
    p = (t + 0.1 * torch.randn_like(t)).clamp(0, 1)
    s1, c1 = psnr_sums(p[:1], t[:1])
    s2, c2 = psnr_sums(p[1:], t[1:])
    sa, ca = psnr_sums(p, t)
    assert c1 + c2 == ca and float(s1 + s2) == pytest.approx(float(sa), rel=1e-12)
    per = ssim_per_image(p, t)
    assert per.shape == (4,) and torch.allclose(per, torch.cat([ssim_per_image(p[i:i + 1], t[i:i + 1]) for i in range(4)]), atol=1e-6)
    assert float(per.max()) < 1.0 and float(per.min()) > 0.0
