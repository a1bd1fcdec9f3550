"""BASELINE.json configs[1] at FULL size (256x256, batch 16) -- too large for the CPU oracle, so the step is checked through
size-independent properties: (1) determinism (bitwise), (2) the halo-tile kernels against the flat implicit-GEMM kernels
(two different algorithms, VAEHIP_FLAT_CONV=1 selects the second), (3) the batch mean: one step on 16 images equals the mean
of two steps on 8 -- different tile counts, split-K plans and GroupNorm chunkings on the same math.  The same in bf16 mode
(tile + activation-image kernels against the flat bf16 kernels): there a different summation order moves a conv output by
~1e-6, which flips the bf16 rounding of ~1e-3 of the next layer's operands by one ulp (0.4 %), so two correct
evaluations differ by ~2e-4 per layer and ~2e-3 in the gradient after 50 layers: the tolerances are the arithmetic's."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
TRACKED = ["encoder.conv_in", "encoder.down_blocks.0.resnets.0.norm1", "decoder.up_blocks.1.resnets.0.norm1"]


@pytest.fixture(scope="module")
def model(cuda):
    from models.sdxl_vae_wrapper import SDXLVAEWrapper
    return SDXLVAEWrapper("synthetic:42", device=cuda)


def _step(w, x, eps, klw=1e-6):
    eng = w.vae.engine
    got = {}
    hs = [eng.add_tracker(w.vae.get_submodule(n), "output", lambda v, n=n: got.__setitem__(n, v.clone())) for n in TRACKED]
    res = eng.forward_backward(x, eps, klw)
    for h in hs:
        h.remove()
    torch.cuda.synchronize()
    return res["scalars"].clone(), w.vae.arena.grad.clone(), got


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("mode,tol,gtol", [("no", 2e-5, 2e-4), ("bf16", 5e-4, 2e-2)])
def test_full_size_step_properties(model, mode, tol, gtol):
    import vae_oracle as vo
    w = model
    eng = w.vae.engine
    eng.set_precision(mode)
    try:
        B, R = 16, 256
        x, eps = vo.synthetic_pixels(B, R, 42).cuda(), vo.synthetic_eps(B, R, 42).cuda()
        s1, g1, t1 = _step(w, x, eps)
        s2, g2, t2 = _step(w, x, eps)
        assert torch.equal(s1, s2) and torch.equal(g1, g2), "the step is not deterministic"
        assert all(torch.equal(t1[n], t2[n]) for n in TRACKED)
        assert torch.isfinite(s1).all() and torch.isfinite(g1).all()

        # (2) a different algorithm for every 3x3 layer: flat implicit GEMM instead of the halo-tile kernels
        os.environ["VAEHIP_FLAT_CONV"] = "1"
        try:
            sf, gf, tf = _step(w, x, eps)
        finally:
            del os.environ["VAEHIP_FLAT_CONV"]
        assert float((sf - s1).abs().max() / s1.abs().max()) < tol
        assert _rel(gf, g1) < gtol, _rel(gf, g1)
        for n in TRACKED:
            assert float(((tf[n] - t1[n]).abs() / t1[n].abs()).max()) < gtol, n
        assert not torch.equal(gf, g1)  # the other kernels really ran

        # (3) batch mean: 16 images == mean of two steps on 8 (per-sample GroupNorm, mean-reduced losses)
        sa, ga, ta = _step(w, x[:8], eps[:8])
        sb, gb, tb = _step(w, x[8:], eps[8:])
        assert float((0.5 * (sa + sb) - s1).abs().max() / s1.abs().max()) < tol
        assert _rel(0.5 * (ga + gb), g1) < gtol
        for n in TRACKED:
            assert float(((0.5 * (ta[n] + tb[n]) - t1[n]).abs() / t1[n].abs()).max()) < gtol, n
    finally:
        eng.set_precision("no")
