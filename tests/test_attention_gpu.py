"""Blockwise (flash-style) attention kernels, C-ABI vae_attn_fwd / vae_attn_bwd (csrc/attention.hip): single head of
width 512, no T x T tensor.  Parity: fp32 against a float64 restatement of oracle/vae_oracle.py:Attention's score /
softmax / context arithmetic (1e-4 relative, north_star's fp32 bar) at T = 1024 and 4096, and against the oracle MODULE
itself (GroupNorm + linears + residual, forward and every gradient); bf16 against the same arithmetic on bf16-rounded
operands (P and dS are rounded to bf16 for the second product: 2^-9 relative per element); and the engine's two
attention paths (materialised scores vs blockwise) against each other."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, do, scale):
    q, k, v, do = (t.double().cpu().requires_grad_(True) if i < 3 else t.double().cpu() for i, t in enumerate((q, k, v, do)))
    s = torch.bmm(q, k.transpose(1, 2)) * scale
    p = torch.softmax(s, dim=-1)
    o = torch.bmm(p, v)
    o.backward(do)
    return o.detach(), torch.logsumexp(s, dim=-1).detach(), q.grad, k.grad, v.grad


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("B,T", [(2, 1024), (1, 4096), (3, 64)])
def test_attention_fp32_matches_float64_reference(cuda, B, T):
    from vaehip import ops
    torch.manual_seed(T)
    C = 512
    # scores with a spread of several units (softmax far from uniform), as after a trained GroupNorm + linear
    q, k, v, do = (torch.randn(B, T, C, device=cuda) * s for s in (2.0, 2.0, 1.0, 1.0))
    scale = C ** -0.5
    with ops.precision(ops.PREC_F32):
        o, saved = ops.attn_fwd(q, k, v, scale)
        dq, dk, dv = ops.attn_bwd(saved, o, do, scale)
    torch.cuda.synchronize()
    ro, rl, rq, rk, rv = _ref(q, k, v, do, scale)
    assert _rel(o, ro) < 1e-4 and _rel(saved[3], rl) < 1e-5
    assert _rel(dq, rq) < 1e-4 and _rel(dk, rk) < 1e-4 and _rel(dv, rv) < 1e-4
    # deterministic: the partial score tiles are summed in a fixed order
    with ops.precision(ops.PREC_F32):
        o2, saved2 = ops.attn_fwd(q, k, v, scale)
        dq2, dk2, dv2 = ops.attn_bwd(saved2, o2, do, scale)
    assert torch.equal(o, o2) and torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)


@pytest.mark.parametrize("B,T", [(2, 1024), (1, 4096)])
def test_attention_bf16_matches_reference_on_rounded_operands(cuda, B, T):
    from vaehip import ops
    torch.manual_seed(T + 1)
    C = 512
    q, k, v, do = (torch.randn(B, T, C, device=cuda) * s for s in (2.0, 2.0, 1.0, 1.0))
    scale = C ** -0.5
    with ops.precision(ops.PREC_BF16):
        o, saved = ops.attn_fwd(q, k, v, scale)
        dq, dk, dv = ops.attn_bwd(saved, o, do, scale)
    torch.cuda.synchronize()
    assert saved[0].dtype == torch.bfloat16
    rb = lambda t: t.bfloat16().float()
    ro, rl, rq, rk, rv = _ref(rb(q), rb(k), rb(v), rb(do), scale)
    assert _rel(saved[3], rl) < 1e-5          # scores: exact products of the rounded operands, fp32 sums
    assert _rel(o, ro) < 4e-3                  # P rounded to bf16 for P.V
    # D = rowsum(dO * O) uses the fp32 dO and the kernel's O, the reference the rounded dO: 1e-2 covers both roundings
    assert _rel(dv, rv) < 1e-2 and _rel(dq, rq) < 1e-2 and _rel(dk, rk) < 1e-2


@pytest.mark.parametrize("hw", [32, 64])
def test_attention_module_blockwise_matches_oracle_module(cuda, hw):
    """the mid-block Attention (GroupNorm, to_q/k/v, attention, to_out, residual) at 32x32 = 1024 and 64x64 = 4096 tokens,
    forced onto the blockwise kernels, against oracle/vae_oracle.py:Attention: output and every gradient."""
    import vae_oracle as vo
    from vaehip import autoencoder as A
    from vaehip import ops
    from models.sdxl_vae_wrapper import SDXLVAEWrapper
    o = vo.OracleWrapper(seed=42)
    w = SDXLVAEWrapper("synthetic:1")
    w.vae.load_state_dict(o.vae.state_dict())
    w.to(cuda)
    eng = w.vae.engine
    m = w.vae.decoder.mid_block.attentions[0]
    om = o.vae.decoder.mid_block.attentions[0]
    assert isinstance(m, A.Attention)
    torch.manual_seed(hw)
    B, C = 1, 512
    x = torch.randn(B, C, hw, hw) * 1.5
    with torch.no_grad():  # weights that give the softmax some contrast (the synthetic init is nearly uniform attention)
        for mod, omod in ((m.to_q, om.to_q), (m.to_k, om.to_k)):
            omod.weight.mul_(6.0)
            mod.weight.copy_(omod.weight.to(cuda))
    dy = torch.randn(B, C, hw, hw)
    xo = x.clone().requires_grad_(True)
    for p in om.parameters():
        p.grad = None
    yo = om(xo)
    yo.backward(dy)

    prev = ops.ATTN_BLOCKWISE_MIN_T
    ops.ATTN_BLOCKWISE_MIN_T = 64
    n0 = dict(ops.ATTN_CALLS)
    try:
        tape = []
        gbuf = torch.zeros_like(w.vae.arena.grad)
        xn = x.to(cuda).permute(0, 2, 3, 1).contiguous()
        with eng._mode():
            y = eng._attention(m, xn, tape, notify=False)
            dx = eng.run_tape(tape, dy.to(cuda).permute(0, 2, 3, 1).contiguous(), gbuf)
    finally:
        ops.ATTN_BLOCKWISE_MIN_T = prev
    torch.cuda.synchronize()
    assert ops.ATTN_CALLS["blockwise_fwd"] == n0["blockwise_fwd"] + 1 and ops.ATTN_CALLS["blockwise_bwd"] == n0["blockwise_bwd"] + 1
    assert _rel(y.permute(0, 3, 1, 2), yo.detach()) < 1e-4
    assert _rel(dx.permute(0, 3, 1, 2), xo.grad) < 1e-4
    for (name, p), (_, po) in zip(m.named_parameters(), om.named_parameters()):
        g = w.vae.arena.grad_view(p, gbuf)
        ref = po.grad
        if name == "to_k.bias":  # mathematically zero (softmax shift invariance): compare on the scale of to_q's gradient
            assert float(g.abs().max()) < 1e-4 * float(om.to_q.bias.grad.abs().max()) + 1e-7
            continue
        assert _rel(g, ref) < 2e-4, name
    with torch.no_grad():
        for mod, omod in ((m.to_q, om.to_q), (m.to_k, om.to_k)):
            omod.weight.div_(6.0)


@pytest.mark.parametrize("mode,tol", [("no", 2e-5), ("bf16", 1e-2)])
def test_engine_blockwise_path_equals_materialised_path(cuda, mode, tol):
    """one train step at 256x256 (T = 1024 in both mid blocks) with the attention on the blockwise kernels against the
    same step with the materialised T x T scores: two algorithms, same losses and gradients."""
    import vae_oracle as vo
    from vaehip import ops
    from models.sdxl_vae_wrapper import SDXLVAEWrapper
    w = SDXLVAEWrapper("synthetic:3", device=cuda)
    eng = w.vae.engine
    eng.set_precision(mode)
    x, eps = vo.synthetic_pixels(2, 256, 7).cuda(), vo.synthetic_eps(2, 256, 7).cuda()
    res = eng.forward_backward(x, eps, 1e-6)
    s0, g0 = res["scalars"].clone(), w.vae.arena.grad.clone()
    prev = ops.ATTN_BLOCKWISE_MIN_T
    ops.ATTN_BLOCKWISE_MIN_T = 64
    n0 = dict(ops.ATTN_CALLS)
    try:
        res = eng.forward_backward(x, eps, 1e-6)
    finally:
        ops.ATTN_BLOCKWISE_MIN_T = prev
        eng.set_precision("no")
    s1, g1 = res["scalars"].clone(), w.vae.arena.grad.clone()
    assert ops.ATTN_CALLS["blockwise_fwd"] == n0["blockwise_fwd"] + 2 and ops.ATTN_CALLS["blockwise_bwd"] == n0["blockwise_bwd"] + 2
    assert float((s1 - s0).abs().max() / s0.abs().max()) < tol
    assert float((g1.double() - g0.double()).norm() / g0.double().norm()) < 10 * tol
    assert not torch.equal(g0, g1)
