"""End-to-end parity of the HIP engine against the CPU oracle (oracle/vae_oracle.py) on the same
seeded weights / pixels / eps.  Tolerance per BASELINE.json north_star: 1e-4 relative for fp32
outputs (reconstruction MSE, KL, per-channel stats); the inactivity mask must be identical."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TRACKED = ["encoder.conv_in", "encoder.down_blocks.0.resnets.0.norm1", "decoder.up_blocks.1.resnets.0.norm1"]


# 2x the worst per-tensor gradient error measured on MI355X (profiles/r03_parity_measured.json; round 4 with F(4x4,3x3) in the path:
# profiles/r04_parity_measured.json), worst per case over the three fp32 paths (Winograd + activation image, Winograd fused, direct):
# R=32 2.5e-5, R=64 2.3e-5 (3.9e-5 with F(4x4), inside the bar), R=40 3.6e-5, R=48 2.8e-5, R=128 4.6e-5 -- every tensor inside
# north_star's 1e-4
GRAD_TOL = {(32, 2): 5.0e-5, (64, 2): 4.6e-5, (40, 1): 7.2e-5, (48, 3): 5.6e-5, (128, 1): 9.3e-5}


def ops_mod():
    from vaehip import ops
    return ops


def _record(kind, key, value):
    """measured parity figures of this run -> gpurun_out/parity_measured.json (copied to profiles/ when tolerances are set)"""
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if not os.path.isdir(out):
        return
    path = os.path.join(out, "parity_measured.json")
    try:
        d = json.load(open(path))
    except Exception:
        d = {}
    d.setdefault(kind, {})[key] = value
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)


def _rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def pair(cuda):
    import vae_oracle as vo
    from models.sdxl_vae_wrapper import SDXLVAEWrapper
    o = vo.OracleWrapper(seed=42)
    w = SDXLVAEWrapper("synthetic:1")
    w.vae.load_state_dict(o.vae.state_dict())
    w.to(cuda)
    assert w.vae.arena.flat.is_cuda and w.vae.arena.owns(w.vae)
    return o, w


def _oracle_step(o, x, eps, klw):
    import vae_oracle as vo
    stats = {}
    hooks = []
    for name in TRACKED:
        mod = o.vae.get_submodule(name)
        hooks.append(mod.register_forward_hook(
            lambda m, i, out, name=name: stats.__setitem__(name, vo.mean_abs_per_channel(out))))
    for p in o.parameters():
        p.grad = None
    out = o(x, sample_posterior=True, eps=eps)
    rec, kl, total = vo.losses(out, x, klw)
    total.backward()
    for h in hooks:
        h.remove()
    return out, rec, kl, total, stats


# (32,2) and (64,2): halo-tile kernels on the wide layers; (40,1) and (48,3): ragged sizes -- widths 40/20/10/5 and
# 48/24/12/6 are no multiple of the 32-pixel tile, batch 1 / odd batch, attention over 25 / 36 tokens: flat kernels
# (128,1) (round 4): maps of 128 / 64 / 32 pixels are whole 16 x 32 tiles: every plain 3x3 layer down to the 32-pixel level runs the
# F(4x4,3x3) kernels, the 16-pixel level F(2x2,3x3)
@pytest.mark.parametrize("R,B,klw", [(32, 2, 1e-6), (64, 2, 1e-2), (40, 1, 1e-4), (48, 3, 1e-3), (128, 1, 1e-4)])
def test_forward_backward_matches_oracle(pair, R, B, klw):
    import vae_oracle as vo
    o, w = pair
    x, eps = vo.synthetic_pixels(B, R, 42), vo.synthetic_eps(B, R, 42)
    out, rec, kl, total, stats = _oracle_step(o, x, eps, klw)

    eng = w.vae.engine
    got = {}
    handles = [eng.add_tracker(w.vae.get_submodule(n), "output", lambda v, n=n: got.__setitem__(n, v)) for n in TRACKED]
    res = eng.forward_backward(x.cuda(), eps.cuda(), klw)
    for h in handles:
        h.remove()
    sc = res["scalars"].cpu()
    assert abs(sc[0] - rec.item()) / rec.item() < 1e-4
    assert abs(sc[1] - kl.item()) / abs(kl.item()) < 1e-4
    assert abs(sc[2] - total.item()) / abs(total.item()) < 1e-4
    assert _rel(res["reconstruction"].permute(0, 3, 1, 2), out["reconstruction"]) < 1e-4
    with torch.no_grad():
        mom_ref = o.vae.quant_conv(o.vae.encoder(x))
    assert _rel(res["moments"].permute(0, 3, 1, 2), mom_ref) < 1e-4
    assert _rel(res["latents"].permute(0, 3, 1, 2), out["latents_sampled"]) < 1e-4
    # per-channel tracker stats (monitor.py:66) within 1e-4 relative, and identical inactivity mask
    for n in TRACKED:
        ref = stats[n]
        v = got[n].cpu().numpy()
        assert np.max(np.abs(v - ref) / ref) < 1e-4, n
        thr = np.float32(np.median(ref))
        assert np.array_equal(v < thr, ref < thr), n
    # every gradient: max |error| of a tensor relative to that tensor's own max |gradient|.  Tensors whose reference gradient is
    # below 1e-5 of the largest gradient entry of the model are held to that global scale instead: attention to_k.bias has a
    # mathematically ZERO gradient (softmax shift invariance) -- its reference value is rounding noise.
    worst, worst_name, worst_small, worst_small_name = 0.0, None, 0.0, None
    oparams = dict(o.vae.named_parameters())
    gmax = max(float(p.grad.abs().max()) for p in o.vae.parameters())
    for name, p in w.vae.named_parameters():
        ref = oparams[name].grad.double()
        err = float((p.grad.detach().double().cpu() - ref).abs().max())
        rmax = float(ref.abs().max())
        if rmax < 1e-5 * gmax:
            if err / gmax > worst_small:
                worst_small, worst_small_name = err / gmax, name
        elif err / rmax > worst:
            worst, worst_name = err / rmax, name
    # tolerance = 2x the worst error MEASURED for this case on MI355X (profiles/r03_parity_measured.json), not a flat figure;
    # north_star's 1e-4 holds for the gradient NORM below
    assert worst <= GRAD_TOL[(R, B)], (worst_name, worst, GRAD_TOL[(R, B)])
    assert worst_small <= 1e-6, (worst_small_name, worst_small)
    gn_ref = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in o.vae.parameters()))
    gn = torch.sqrt((w.vae.arena.grad.double() ** 2).sum()).cpu()
    assert abs(gn - gn_ref) / gn_ref < 1e-4
    print(f"R={R} B={B} worst grad rel err {worst:.3e} ({worst_name})")
    _record("grad_worst", f"R={R},B={B},wino={ops_mod().WINOGRAD},act32_min={ops_mod().ACT_IMAGE32_MIN_CIN}",
            {"rel_to_own_max": worst, "tensor": worst_name, "near_zero_rel_to_global_max": worst_small, "near_zero_tensor": worst_small_name})


@pytest.mark.parametrize("path", ["winograd_fused_transform", "direct_kernels"])
@pytest.mark.parametrize("R,B,klw", [(32, 2, 1e-6), (64, 2, 1e-2)])
def test_other_fp32_conv_paths_match_oracle(pair, monkeypatch, R, B, klw, path):
    """the default fp32 path (Winograd kernels reading a materialised GroupNorm+SiLU tensor) is what the test above runs; the
    two alternatives -- Winograd with the transform fused into the halo staging, and the direct halo-tile kernels -- are held
    to the same oracle tolerances"""
    from vaehip import ops
    if path == "winograd_fused_transform":
        monkeypatch.setattr(ops, "ACT_IMAGE32_MIN_CIN", 10 ** 9)
    else:
        monkeypatch.setattr(ops, "WINOGRAD", False)
    prof = ops.LaunchProfiler()
    monkeypatch.setattr(ops, "PROFILER", prof)
    test_forward_backward_matches_oracle(pair, R, B, klw)
    monkeypatch.setattr(ops, "PROFILER", None)
    # which kernels served the step (a test that passes on the DEFAULT path would prove nothing about the alternative):
    names = {}
    for rec in prof.records:
        names[rec[0]] = names.get(rec[0], 0) + 1
    wino_fused = sum(n for k, n in names.items() if k.startswith(("conv3_wino_kernel<2,", "conv3_wino_kernel<1,", "wgrad3_wino_kernel<2>", "wgrad3_wino_kernel<1>")))
    wino_plain_fwd = sum(n for k, n in names.items() if k.startswith("conv3_wino_kernel<0,"))
    wino_any = sum(n for k, n in names.items() if "wino" in k and "reduce" not in k)
    direct = sum(n for k, n in names.items() if k.startswith(("conv3_tile_kernel", "wgrad3_tile_kernel")))
    if path == "winograd_fused_transform":
        # GroupNorm+SiLU applied inside the Winograd kernels' halo staging (xf template argument 2) in the forward AND the weight
        # gradient of every resnet convolution they take; the launches left on the untransformed instantiation <0,..> are the
        # dgrads (their operand is a gradient: nothing to transform)
        n_fwd_fused = sum(n for k, n in names.items() if k.startswith("conv3_wino_kernel<2,"))
        n_wg_fused = sum(n for k, n in names.items() if k.startswith("wgrad3_wino_kernel<2>"))
        assert n_fwd_fused > 0 and n_wg_fused > 0, names
        assert wino_plain_fwd <= n_fwd_fused, names   # at most one dgrad per fused forward: no forward ran on a materialised input
        assert sum(n for k, n in names.items() if k.startswith("wgrad3_wino_kernel<0>")) == 0, names
    else:
        assert wino_any == 0 and direct > 0, names
    _record("kernels_served", f"R={R},B={B},path={path}", names)


def test_autograd_path_equals_fast_path(pair):
    """SDXLVAEWrapper.forward + torch losses + .backward() (the reference's train.py:287-299 sequence)
    must give the same numbers as the fused path."""
    import vae_oracle as vo
    o, w = pair
    R, B, klw = 32, 2, 1e-3
    x, eps = vo.synthetic_pixels(B, R, 7).cuda(), vo.synthetic_eps(B, R, 7).cuda()
    res = w.vae.engine.forward_backward(x, eps, klw)
    fast = w.vae.arena.grad.clone()
    for p in w.parameters():
        p.grad = None
    d = w.vae.encode(x).latent_dist
    z = d.sample(eps)
    recon = w.vae.decode(z).sample
    rec = F.mse_loss(recon.float(), x.float(), reduction="mean")
    kl = d.kl().mean()
    (rec + klw * kl).backward()
    sc = res["scalars"].cpu()
    assert abs(sc[0] - rec.item()) / rec.item() < 1e-6
    assert abs(sc[1] - kl.item()) / abs(kl.item()) < 1e-5
    gmax = float(fast.abs().max())
    for name, p in w.vae.named_parameters():
        assert p.grad is not None, name
        ref = w.vae.arena.view_of(fast, p, w.vae.arena.offset_of[id(p)])
        err = float((p.grad - ref).abs().max())
        assert err <= 1e-5 * float(ref.abs().max()) + 1e-7 * gmax, (name, err)


def test_foreign_hooks_get_real_tensors(pair):
    """register_forward_hook / register_forward_pre_hook protocol (monitor.py:126-133) on fused layers."""
    import vae_oracle as vo
    o, w = pair
    x = vo.synthetic_pixels(2, 32, 3)
    names = ["encoder.down_blocks.0.resnets.0.norm1", "encoder.down_blocks.1.resnets.0.conv2",
             "decoder.mid_block.attentions.0.to_q", "decoder.up_blocks.0"]
    ref, got, handles = {}, {}, []
    for n in names:
        handles.append(o.vae.get_submodule(n).register_forward_hook(lambda m, i, out, n=n: ref.__setitem__(n, (i[0].detach(), out.detach()))))
        handles.append(w.vae.get_submodule(n).register_forward_hook(lambda m, i, out, n=n: got.__setitem__(n, (i[0].detach().cpu(), out.detach().cpu()))))
    pre = {}
    handles.append(w.vae.get_submodule(names[0]).register_forward_pre_hook(lambda m, i: pre.__setitem__("x", i[0].shape)))
    with torch.no_grad():
        o(x, sample_posterior=False)
        out = w(x.cuda(), sample_posterior=False)
    for h in handles:
        h.remove()
    assert out["reconstruction"].shape == (2, 3, 32, 32)
    assert pre["x"] == (2, 128, 32, 32)
    for n in names:
        assert got[n][0].shape == ref[n][0].shape and got[n][1].shape == ref[n][1].shape, n
        assert _rel(got[n][0], ref[n][0]) < 1e-4 and _rel(got[n][1], ref[n][1]) < 1e-4, n
    # hooks removed -> fused path again, same result (a hooked conv runs without the fused residual / statistics epilogue and
    # may land on another kernel than the fused one -- Winograd vs direct --, so equal to summation accuracy, not bitwise)
    with torch.no_grad():
        out2 = w(x.cuda(), sample_posterior=False)
    assert _rel(out2["reconstruction"], out["reconstruction"]) < 5e-5


def test_standalone_submodule_calls(pair):
    import vae_oracle as vo
    o, w = pair
    gen = torch.Generator().manual_seed(0)
    t = torch.randn(1, 128, 8, 8, generator=gen)
    with torch.no_grad():
        r = w.vae.encoder.down_blocks[0].resnets[0](t.cuda())
        ref = o.vae.encoder.down_blocks[0].resnets[0](t)
        assert _rel(r, ref) < 1e-4
        g = w.vae.encoder.down_blocks[0].resnets[0].norm1(t.cuda())
        assert _rel(g, o.vae.encoder.down_blocks[0].resnets[0].norm1(t)) < 1e-5


def test_bf16_compute_mode_tracks_fp32(pair):
    """training.mixed_precision: bf16 -- bf16 MFMA products with fp32 accumulation, fp32 tensors and statistics.
    Tolerance is the arithmetic's (bf16 has 8 significant bits): losses 2e-2, gradient direction cosine > 0.99."""
    import vae_oracle as vo
    o, w = pair
    R, B, klw = 64, 2, 1e-4
    x, eps = vo.synthetic_pixels(B, R, 42, 5).cuda(), vo.synthetic_eps(B, R, 42, 5).cuda()
    eng = w.vae.engine
    names = ["encoder.down_blocks.0.resnets.0.norm1", "decoder.up_blocks.1.resnets.0.norm1"]
    res, grads, stats = {}, {}, {}
    for mode in ("no", "bf16"):
        eng.set_precision(mode)
        got = {}
        hs = [eng.add_tracker(w.vae.get_submodule(n), "output", lambda v, n=n: got.__setitem__(n, v.cpu())) for n in names]
        res[mode] = eng.forward_backward(x, eps, klw)["scalars"].cpu()
        grads[mode] = w.vae.arena.grad.clone()
        stats[mode] = got
        for h in hs:
            h.remove()
    eng.set_precision("no")
    assert float((res["bf16"] - res["no"]).abs().max() / res["no"].abs().max()) < 2e-2
    assert not torch.equal(res["bf16"], res["no"])  # the bf16 kernels really ran
    cos = torch.nn.functional.cosine_similarity(grads["bf16"].double(), grads["no"].double(), dim=0)
    assert float(cos) > 0.99, float(cos)
    gn = grads["bf16"].norm() / grads["no"].norm()
    assert 0.97 < float(gn) < 1.03
    for n in names:
        assert float(((stats["bf16"][n] - stats["no"][n]).abs() / stats["no"][n]).max()) < 2e-2
    with pytest.raises(NotImplementedError):
        eng.set_precision("fp16")


def test_checkpointed_decoder_is_bitwise_identical(pair):
    """training.gradient_checkpointing: decoder (BASELINE config 5) -- every resnet / attention / sampler of the decoder
    keeps only its input and is re-run before its own backward.  Every kernel is deterministic, so losses and ALL
    gradients are bit-identical, and trackers / hooks fire once per step, not once per pass."""
    import vae_oracle as vo
    o, w = pair
    R, B, klw = 128, 2, 1e-3
    x, eps = vo.synthetic_pixels(B, R, 42, 9).cuda(), vo.synthetic_eps(B, R, 42, 9).cuda()
    eng = w.vae.engine
    name = "decoder.up_blocks.1.resnets.0.norm1"
    out = {}
    for ck in (False, True):
        eng.checkpoint_decoder = ck
        fired, hooked = [], []
        h1 = eng.add_tracker(w.vae.get_submodule(name), "output", lambda v: fired.append(v.clone()))
        h2 = w.vae.decoder.conv_out.register_forward_hook(lambda m, i, o_: hooked.append(tuple(o_.shape)))
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        res = eng.forward_backward(x, eps, klw)
        h1.remove(); h2.remove()
        peak = torch.cuda.max_memory_allocated() - base
        out[ck] = (res["scalars"].clone(), w.vae.arena.grad.clone(), fired, hooked, peak)
        del res
    eng.checkpoint_decoder = False
    assert torch.equal(out[True][0], out[False][0])
    assert torch.equal(out[True][1], out[False][1])
    assert len(out[True][2]) == 1 and torch.equal(out[True][2][0], out[False][2][0])
    assert out[True][3] == out[False][3] == [(B, 3, R, R)]
    assert out[True][4] < out[False][4]  # the point of it: lower peak memory
    print(f"peak memory {out[False][4] / 2**20:.0f} MiB -> {out[True][4] / 2**20:.0f} MiB with the decoder checkpointed")


def test_gradient_accumulation_equals_full_batch_step(pair, cuda):
    """training.gradient_accumulation_steps = 2 over two half batches == one step on the whole batch (GroupNorm is per
    sample, both losses are batch means): same clipped AdamW update, one optimizer / scheduler step per two calls."""
    import vae_oracle as vo
    from models.sdxl_vae_wrapper import SDXLVAEWrapper
    from vaehip.trainer import HipTrainer
    o, w0 = pair
    R, B = 32, 4
    x, eps = vo.synthetic_pixels(B, R, 42, 3).cuda(), vo.synthetic_eps(B, R, 42, 3).cuda()
    params, gnorm = {}, {}
    for accum in (1, 2):
        w = SDXLVAEWrapper("synthetic:1")
        w.vae.load_state_dict(o.vae.state_dict())
        w.to(cuda)
        tr = HipTrainer(w, lr=1e-3, kl_weight=1e-3, lr_warmup_steps=1, max_train_steps=10, gradient_accumulation_steps=accum)
        for it in range(2):  # two optimizer updates (the first has lr = 0, train.py:197-200)
            if accum == 1:
                tr.train_step(x, eps)
                assert tr.sync_gradients
            else:
                tr.train_step(x[:2], eps[:2])
                assert not tr.sync_gradients and tr.global_step == it
                tr.train_step(x[2:], eps[2:])
                assert tr.sync_gradients
            assert tr.global_step == it + 1
        params[accum] = w.vae.arena.flat.clone()
        gnorm[accum] = float(w.vae.arena.grad.double().norm())
    assert abs(gnorm[1] - gnorm[2]) / gnorm[1] < 1e-5
    moved = (params[1] - w0.vae.arena.flat).abs().max()
    assert float(moved) > 0
    assert float((params[1] - params[2]).abs().max()) < 2e-2 * float(moved)  # Adam's sign-like first step amplifies rounding
