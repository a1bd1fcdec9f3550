"""Per-kernel parity: every HIP entry point (through the C ABI) against a plain PyTorch fp32
CPU reference of the same op, on seeded inputs.  Tolerances: fp32 MFMA is an exact-product
fp32 fmaf chain, so differences are summation-order only (1e-5 relative to the operand scale)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.double().cpu()
    b = b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _nhwc(x):  # NCHW cpu -> NHWC cuda
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(x):  # NHWC cuda -> NCHW cpu
    return x.permute(0, 3, 1, 2).cpu()


def _ref_conv(x, w, b, kind):
    if kind == "c3":
        return F.conv2d(x, w, b, 1, 1)
    if kind == "c1":
        return F.conv2d(x, w, b)
    if kind == "c3s2":
        return F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, 2, 0)
    if kind == "c3up":
        return F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, 1, 1)


def _to_dev_ohwi(w):
    co, ci, kh, kw = w.shape
    buf = w.permute(0, 2, 3, 1).contiguous().cuda()  # OHWI memory
    return buf.permute(0, 3, 1, 2)  # logical OIHW view


CONV_CASES = [
    # kind, B, H, W, Ci, Co
    ("c3", 2, 8, 8, 128, 128),
    ("c3", 1, 12, 20, 128, 256),
    ("c3", 2, 5, 7, 256, 128),      # ragged M
    ("c3", 2, 16, 16, 3, 128),      # conv_in (scalar path)
    ("c3", 2, 16, 16, 128, 3),      # conv_out (skinny N)
    ("c3", 2, 4, 4, 512, 8),        # encoder.conv_out
    ("c3", 2, 4, 4, 4, 512),        # decoder.conv_in
    ("c1", 2, 8, 8, 128, 256),      # shortcut
    ("c1", 2, 4, 4, 8, 8),          # quant_conv
    ("c1", 2, 4, 4, 4, 4),          # post_quant_conv
    ("c3s2", 2, 16, 16, 128, 128),  # downsampler
    ("c3s2", 1, 10, 14, 128, 128),
    ("c3s2", 2, 32, 32, 128, 128),  # B*H*W/4 % 128 == 0: parity-class-major dgrad (VAE_MODE_DGRAD_S2)
    ("c3s2", 1, 16, 64, 256, 256),
    ("c3up", 2, 8, 8, 128, 128),    # upsampler
    ("c3up", 1, 5, 6, 256, 256),
    # shapes served by the LDS halo-tile kernel (W % 32 == 0, H % 4 == 0, Cout > 32)
    ("c3", 2, 8, 32, 128, 128),
    ("c3", 1, 4, 64, 256, 512),
    ("c3", 2, 12, 32, 512, 256),
    ("c3up", 2, 4, 16, 128, 256),
    ("c3up", 1, 6, 32, 256, 256),
    # low-resolution W % 32 == 0 and H % 4 == 0: four phase convolutions with 2x2 effective kernels (forward and dgrad)
    ("c3up", 2, 4, 32, 128, 256),
    ("c3up", 1, 8, 64, 256, 128),
]


@pytest.mark.parametrize("kind,B,H,W,Ci,Co", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(cuda, kind, B, H, W, Ci, Co):
    from vaehip import ops
    gen = torch.Generator().manual_seed(1234 + Ci + Co + H)
    k = 1 if kind == "c1" else 3
    x = torch.randn(B, Ci, H, W, generator=gen)
    w = torch.randn(Co, Ci, k, k, generator=gen) / math.sqrt(Ci * k * k)
    b = torch.randn(Co, generator=gen)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y_ref = _ref_conv(xr, wr, br, kind)
    dy = torch.randn(y_ref.shape, generator=gen)
    y_ref.backward(dy)

    cpad = 4 if Ci == 3 else Ci
    xd = torch.zeros(B, H, W, cpad)
    xd[..., :Ci] = x.permute(0, 2, 3, 1)
    xd = xd.cuda()
    wd = _to_dev_ohwi(w)
    y = ops.conv_fwd(xd, wd, b.cuda(), kind)
    assert _rel(_nchw(y), y_ref.detach()) < 2e-5

    dyd = _nhwc(dy)
    if Ci != 3:
        dx = ops.conv_dgrad(dyd, wd, kind, (H, W))
        assert _rel(_nchw(dx), xr.grad) < 2e-5
    gw = torch.full_like(wd.permute(0, 2, 3, 1).contiguous(), float("nan")).permute(0, 3, 1, 2)
    gb = torch.full((Co,), float("nan"), device="cuda")
    ops.conv_wgrad(dyd, xd, kind, gw, gb)
    assert _rel(gw.cpu(), wr.grad) < 3e-5
    assert _rel(gb.cpu(), br.grad) < 3e-5


@pytest.mark.parametrize("C,H,W,silu", [(128, 16, 16, True), (256, 8, 8, True), (512, 4, 4, True), (512, 6, 10, False),
                                         (128, 8, 32, True), (512, 4, 32, True), (256, 4, 64, False)])  # last 3: halo-tile kernel
def test_gn_fused_conv_and_backward(cuda, C, H, W, silu):
    """GroupNorm(+SiLU) fused into the conv operand load; GN backward; tracker reduction."""
    from vaehip import ops
    gen = torch.Generator().manual_seed(7 + C)
    B, Co = 2, 128
    x = torch.randn(B, C, H, W, generator=gen) * 1.7 + 0.3
    gamma = 1 + 0.3 * torch.randn(C, generator=gen)
    beta = 0.2 * torch.randn(C, generator=gen)
    w = torch.randn(Co, C, 3, 3, generator=gen) / math.sqrt(9 * C)
    res = torch.randn(B, Co, H, W, generator=gen)
    xr, gr, br, wr = (t.clone().requires_grad_(True) for t in (x, gamma, beta, w))
    n = F.group_norm(xr, 32, gr, br, 1e-6)
    act = F.silu(n) if silu else n
    y_ref = F.conv2d(act, wr, None, 1, 1) + res
    dy = torch.randn(y_ref.shape, generator=gen)
    y_ref.backward(dy)

    xd, gd, bd, wd = _nhwc(x), gamma.cuda(), beta.cuda(), _to_dev_ohwi(w)
    st = ops.gn_stats(xd, gd, bd)
    mean_ref = x.view(B, 32, -1).mean(-1)
    var_ref = x.view(B, 32, -1).var(-1, unbiased=False)
    assert _rel(st.mean, mean_ref) < 1e-5
    assert _rel(st.rstd, 1 / torch.sqrt(var_ref + 1e-6)) < 1e-5
    xf = ops.XF_AFFINE_SILU if silu else ops.XF_AFFINE
    y = ops.conv_fwd(xd, wd, None, "c3", xf=xf, stats=st, res=_nhwc(res))
    assert _rel(_nchw(y), y_ref.detach()) < 3e-5
    # materialised GN output (hook slow path) and the fused tracker metric (monitor.py:66)
    nd = ops.gn_apply(xd, st, ops.XF_AFFINE)
    assert _rel(_nchw(nd), n.detach()) < 1e-5
    tr = ops.gn_track(xd, st)
    tr_ref = n.detach().abs().mean(dim=[0, 2, 3])
    assert float(((tr.cpu() - tr_ref).abs() / tr_ref).max()) < 1e-5
    # backward
    dyd = _nhwc(dy)
    gw = torch.empty_like(wd.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
    ops.conv_wgrad(dyd, xd, "c3", gw, None, xf=xf, stats=st)
    assert _rel(gw.cpu(), wr.grad) < 5e-5
    g = ops.conv_dgrad(dyd, wd, "c3", (H, W))
    dgam = torch.empty(C, device="cuda")
    dbet = torch.empty(C, device="cuda")
    addt = torch.randn(B, H, W, C, generator=gen)
    dx = ops.gn_bwd(xd, g, st, gd, bd, silu, addt.cuda(), dgam, dbet)
    assert _rel(_nchw(dx) - addt.permute(0, 3, 1, 2), xr.grad) < 5e-5
    assert _rel(dgam, gr.grad) < 5e-5
    assert _rel(dbet, br.grad) < 5e-5


def test_conv_track_epilogue(cuda):
    from vaehip import ops
    gen = torch.Generator().manual_seed(5)
    B, H, W, Ci, Co = 2, 16, 16, 3, 128
    x = torch.rand(B, Ci, H, W, generator=gen) * 2 - 1
    w = torch.randn(Co, Ci, 3, 3, generator=gen) / math.sqrt(27)
    b = torch.randn(Co, generator=gen) * 0.1
    xd = torch.zeros(B, H, W, 4)
    xd[..., :3] = x.permute(0, 2, 3, 1)
    xd = xd.cuda()
    ws = ops.conv_track_buffer(B * H * W, Co, "cuda")
    y = ops.conv_fwd(xd, _to_dev_ohwi(w), b.cuda(), "c3", track=ws)
    tr = ops.track_final(ws, B * H * W)
    ref = F.conv2d(x, w, b, 1, 1).abs().mean(dim=[0, 2, 3])
    assert float(((tr.cpu() - ref).abs() / ref).max()) < 1e-5


@pytest.mark.parametrize("z,M,N,K", [(2, 64, 64, 512), (3, 100, 36, 40), (1, 256, 512, 256)])
def test_batched_gemms(cuda, z, M, N, K):
    from vaehip import ops
    gen = torch.Generator().manual_seed(z * 100 + M)
    A = torch.randn(z, M, K, generator=gen)
    Bt = torch.randn(z, N, K, generator=gen)
    assert _rel(ops.gemm_nt(A.cuda(), Bt.cuda(), 0.5), 0.5 * A @ Bt.transpose(1, 2)) < 2e-5
    Bn = torch.randn(z, K, N, generator=gen)
    assert _rel(ops.gemm_nn(A.cuda(), Bn.cuda()), A @ Bn) < 2e-5
    At = torch.randn(z, K, M, generator=gen)
    assert _rel(ops.gemm_tn(At.cuda(), Bn.cuda(), 2.0), 2.0 * At.transpose(1, 2) @ Bn) < 2e-5


def test_softmax_fwd_bwd(cuda):
    from vaehip import ops
    gen = torch.Generator().manual_seed(3)
    S = (torch.randn(2, 70, 300, generator=gen) * 3).requires_grad_(True)
    P_ref = torch.softmax(S, -1)
    dP = torch.randn(P_ref.shape, generator=gen)
    P_ref.backward(dP)
    P = ops.softmax_rows_(S.detach().clone().cuda())
    assert _rel(P, P_ref.detach()) < 1e-5
    dS = ops.softmax_bwd_rows_(P, dP.clone().cuda())
    assert _rel(dS, S.grad) < 2e-5


def test_sample_kl_mse_and_backward(cuda):
    from vaehip import ops
    gen = torch.Generator().manual_seed(11)
    B, h, w, L, R = 3, 4, 4, 4, 32
    mom = torch.randn(B, 2 * L, h, w, generator=gen) * 2
    mom[0, L, 0, 0] = 25.0   # exercises clamp(max=20): zero gradient
    mom[1, L + 1, 1, 1] = -40.0
    eps = torch.randn(B, L, h, w, generator=gen)
    target = torch.rand(B, 3, R, R, generator=gen) * 2 - 1
    recon = torch.randn(B, 3, R, R, generator=gen)
    klw = 1e-3
    mr = mom.clone().requires_grad_(True)
    rr = recon.clone().requires_grad_(True)
    mean, lv = torch.chunk(mr, 2, 1)
    lvc = lv.clamp(-30, 20)
    z_ref = mean + torch.exp(0.5 * lvc) * eps
    kl = 0.5 * torch.sum(mean ** 2 + lvc.exp() - 1 - lvc, dim=[1, 2, 3])
    mse = F.mse_loss(rr, target)
    dz = torch.randn(z_ref.shape, generator=gen)
    total = mse + klw * kl.mean() + (z_ref * dz).sum()
    total.backward()

    md, ed = _nhwc(mom), _nhwc(eps)
    z, klp = ops.sample_kl(md, ed)
    assert _rel(_nchw(z), z_ref.detach()) < 1e-6
    assert _rel(klp.sum(1), kl.detach()) < 1e-5
    rd, td = _nhwc(recon), _nhwc(target)
    sc = ops.mse_kl_loss(rd, td, klp, klw).cpu()
    assert abs(sc[0] - mse.item()) / mse.item() < 1e-6
    assert abs(sc[1] - kl.mean().item()) / abs(kl.mean().item()) < 1e-5
    assert abs(sc[2] - (mse + klw * kl.mean()).item()) < 1e-5
    assert _rel(_nchw(ops.mse_bwd(rd, td)), rr.grad) < 1e-6
    dm = ops.sample_kl_bwd(md, ed, _nhwc(dz), klw)
    assert _rel(_nchw(dm), mr.grad) < 1e-5
    # mode(): eps None
    z0, _ = ops.sample_kl(md, None)
    assert _rel(_nchw(z0), mom[:, :L]) == 0.0


def test_layout_and_pool(cuda):
    from vaehip import ops
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(2, 3, 6, 10, generator=gen)
    y = ops.nchw_to_nhwc(x.cuda(), 4).cpu()
    assert torch.equal(y[..., :3], x.permute(0, 2, 3, 1)) and float(y[..., 3].abs().max()) == 0.0
    assert torch.equal(ops.nhwc_to_nchw(ops.nchw_to_nhwc(x.cuda())).cpu(), x)
    s = torch.randn(2, 8, 12, 128, generator=gen)
    from vaehip.lib import lib
    out = torch.empty(2, 4, 6, 128, device="cuda")
    lib.call("vae_sumpool2x2", ops._p(s.cuda()), 2, 4, 6, 128, ops._p(out), ops._stream())
    ref = s.view(2, 4, 2, 6, 2, 128).sum(dim=(2, 4))
    assert _rel(out, ref) < 1e-6


def test_fused_clip_adamw_matches_torch(cuda):
    """train.py:184-187,301-302: clip_grad_norm_(1.0) then torch.optim.AdamW, 3 steps incl. lr=0 first step."""
    from vaehip import ops
    gen = torch.Generator().manual_seed(21)
    n = 100003
    p0 = torch.randn(n, generator=gen)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p_ref], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    p = p0.clone().cuda()
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    sq = torch.zeros(1, device="cuda")
    for step, lr in enumerate([0.0, 1e-3, 5e-4], start=1):
        g = torch.randn(n, generator=gen) * (3.0 if step != 2 else 1e-3)
        p_ref.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
        for grp in opt.param_groups:
            grp["lr"] = lr
        opt.step()
        gd = g.cuda()
        ops.sqnorm(gd, sq)
        assert abs(math.sqrt(sq.item()) - tn.item()) / tn.item() < 1e-5
        ops.adamw(p, gd, m, v, sq, 1.0, lr, 0.9, 0.999, 1e-8, 1e-2, step)
        assert _rel(p, p_ref.detach()) < 1e-6
    st = opt.state[p_ref]
    # v carries clip^2: twice the fp32 summation-order difference of the two norm computations
    assert _rel(m, st["exp_avg"]) < 1e-5 and _rel(v, st["exp_avg_sq"]) < 5e-5


def test_errors_are_loud(cuda):
    from vaehip import ops, VaeHipError
    x = torch.zeros(1, 4, 4, 128, device="cuda")
    w = torch.zeros(128, 128, 3, 3, device="cuda")  # NOT channels_last memory
    with pytest.raises(ValueError):
        ops.conv_fwd(x, w, None, "c3")
    from vaehip.lib import lib
    with pytest.raises(VaeHipError):
        lib.call("vae_add", None, None, 0, None, None)


# ---------------------------------------------------------------------------------------------------------
# bf16 compute mode (training.mixed_precision: bf16): operands rounded to bf16 at LDS staging, fp32 accumulate.
# bf16 x bf16 products are exact in fp32, so against a CPU fp32 conv of the bf16-ROUNDED operands the kernel
# must agree to summation-order accuracy -- a tight check of every index / transposing-read path.
# ---------------------------------------------------------------------------------------------------------
def _r16(t):
    return t.bfloat16().float()


@pytest.fixture
def bf16_mode():
    """bf16 arithmetic with fp32 STORAGE of the activations (ops.ACT_BF16 = False): the kernels' results can then be held to
    summation-order accuracy against fp32 references.  bf16 storage has its own tests (tests/test_act16_gpu.py)."""
    from vaehip import ops
    ops.PRECISION, keep = ops.PREC_BF16, ops.ACT_BF16
    ops.ACT_BF16 = False
    yield
    ops.PRECISION, ops.ACT_BF16 = ops.PREC_F32, keep


@pytest.fixture
def packed_weights():
    """hands every conv weight to the bf16 kernels as a pre-rounded bf16 image too (what the engine does per step)"""
    from vaehip import ops
    keep = []

    def pack(wd):
        buf = wd.permute(0, 2, 3, 1)  # the OHWI memory behind the logical OIHW view
        assert buf.is_contiguous()
        img = torch.empty(buf.numel(), device="cuda", dtype=torch.bfloat16)
        ops.pack_bf16(buf, img)
        keep.append(img)
        ops.WEIGHTS16 = (buf.data_ptr(), buf.numel() * 4, img.data_ptr())
        return wd
    yield pack
    ops.WEIGHTS16 = None


# (5,128,128,128,128): 640 tiles on 512 persistent workgroups -- some run two tiles through the cross-tile pipeline
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("kind,B,H,W,Ci,Co", [("c3", 2, 8, 32, 128, 128), ("c3", 1, 4, 64, 256, 512), ("c3", 2, 12, 32, 512, 256),
                                              ("c3", 1, 4, 32, 96, 160), ("c3up", 2, 4, 16, 128, 256),
                                              ("c3", 2, 64, 64, 128, 128), ("c3", 5, 128, 128, 128, 128), ("c3", 3, 36, 96, 256, 128)])
def test_bf16_conv_fwd_dgrad(cuda, bf16_mode, packed_weights, packed, kind, B, H, W, Ci, Co):
    from vaehip import ops
    gen = torch.Generator().manual_seed(99 + Ci + Co)
    x = torch.randn(B, Ci, H, W, generator=gen)
    w = torch.randn(Co, Ci, 3, 3, generator=gen) / math.sqrt(Ci * 9)
    b = torch.randn(Co, generator=gen)
    _dev = _to_dev_ohwi(w)
    if packed:
        packed_weights(_dev)
    xr = _r16(x).requires_grad_(True)
    y_ref = _ref_conv(xr, _r16(w), b, kind)
    dy = torch.randn(y_ref.shape, generator=gen)
    y = ops.conv_fwd(_nhwc(x), _dev, b.cuda(), kind)
    assert _rel(_nchw(y), y_ref.detach()) < 2e-5
    # dgrad: dY and W rounded to bf16
    (gx,) = torch.autograd.grad(_ref_conv(xr, _r16(w), None, kind), xr, _r16(dy))
    dx = ops.conv_dgrad(_nhwc(dy), _dev, kind, (H, W))
    assert _rel(_nchw(dx), gx) < 2e-5
    # and the bf16 result is close to the true fp32 conv at bf16 accuracy
    assert _rel(_nchw(y), _ref_conv(x, w, b, kind)) < 2e-2
    # wgrad: dY and X rounded to bf16; bias gradient is summed in fp32 from the unrounded dY
    if Ci % 64 == 0:
        wr = w.clone().requires_grad_(True)
        _ref_conv(_r16(x), wr, None, kind).backward(_r16(dy))
        gw = torch.empty_like(_to_dev_ohwi(w).permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
        gb = torch.empty(Co, device="cuda")
        ops.conv_wgrad(_nhwc(dy), _nhwc(x), kind, gw, gb)
        assert _rel(gw.cpu(), wr.grad) < 3e-5
        assert _rel(gb.cpu(), dy.sum(dim=(0, 2, 3))) < 3e-5


# (6,128,128,128,128): 768 tiles -- a persistent workgroup's second tile belongs to another image (other GroupNorm rows)
@pytest.mark.parametrize("B,C,H,W,Co,packed", [(2, 128, 8, 32, 256, False), (2, 128, 8, 32, 256, True), (6, 128, 128, 128, 128, True)])
def test_bf16_conv_fused_gn_silu(cuda, bf16_mode, packed_weights, B, C, H, W, Co, packed):
    from vaehip import ops
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=gen) * 1.3 + 0.2
    gamma, beta = 1 + 0.3 * torch.randn(C, generator=gen), 0.2 * torch.randn(C, generator=gen)
    w = torch.randn(Co, C, 3, 3, generator=gen) / math.sqrt(9 * C)
    res = torch.randn(B, Co, H, W, generator=gen)
    act = F.silu(F.group_norm(x, 32, gamma, beta, 1e-6))
    y_ref = F.conv2d(_r16(act), _r16(w), None, 1, 1) + res
    xd = _nhwc(x)
    st = ops.gn_stats(xd, gamma.cuda(), beta.cuda())
    wd = _to_dev_ohwi(w)
    if packed:
        packed_weights(wd)
    y = ops.conv_fwd(xd, wd, None, "c3", xf=ops.XF_AFFINE_SILU, stats=st, res=_nhwc(res))
    # rounding of the fp32 activation to bf16 can differ by one bf16 ulp where the GPU/CPU fp32 values differ in the
    # last bit: allow a few 1e-4 of the output scale
    assert _rel(_nchw(y), y_ref) < 5e-4


# shapes the bf16 FLAT kernels serve (igemm_bf16.hip): 1x1, stride 2 (gather dgrad and parity-class dgrad), ragged and
# tiny 3x3, skinny N / skinny M, K tails that are not a multiple of the 64-channel step
BF16_FLAT_CASES = [
    ("c1", 2, 8, 8, 128, 256), ("c1", 2, 4, 4, 8, 8), ("c1", 1, 6, 10, 320, 96),
    ("c3s2", 2, 16, 16, 128, 128), ("c3s2", 1, 10, 14, 128, 128), ("c3s2", 2, 32, 32, 128, 128), ("c3s2", 1, 16, 64, 256, 256),
    ("c3", 2, 5, 7, 256, 128), ("c3", 2, 8, 8, 128, 128), ("c3", 2, 4, 4, 512, 8), ("c3", 2, 4, 4, 4, 512),
    ("c3", 2, 16, 16, 128, 3), ("c3up", 1, 5, 6, 256, 256), ("c3", 1, 9, 5, 100, 36),
]


@pytest.mark.parametrize("kind,B,H,W,Ci,Co", BF16_FLAT_CASES)
def test_bf16_flat_conv_fwd_dgrad_wgrad(cuda, bf16_mode, kind, B, H, W, Ci, Co):
    from vaehip import ops
    gen = torch.Generator().manual_seed(7 + Ci + Co + H)
    k = 1 if kind == "c1" else 3
    x = torch.randn(B, Ci, H, W, generator=gen)
    w = torch.randn(Co, Ci, k, k, generator=gen) / math.sqrt(Ci * k * k)
    b = torch.randn(Co, generator=gen)
    wd = _to_dev_ohwi(w)
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        y = ops.conv_fwd(_nhwc(x), wd, b.cuda(), kind)
        dy = torch.randn(y.permute(0, 3, 1, 2).shape, generator=gen)
        dx = ops.conv_dgrad(_nhwc(dy), wd, kind, (H, W))
        gw = torch.full_like(wd.permute(0, 2, 3, 1).contiguous(), float("nan")).permute(0, 3, 1, 2)
        gb = torch.full((Co,), float("nan"), device="cuda")
        ops.conv_wgrad(_nhwc(dy), _nhwc(x), kind, gw, gb)
    finally:
        ops.PROFILER = None
    names = [r[0] for r in prof.records]
    assert len(names) == 3
    # channel counts that are not a multiple of 4 have no vectorised form and stay on the exact fp32 kernels; a <= 4-channel
    # side puts the layer on the (exact fp32) VALU kernels of skinny.hip
    expect16 = [Ci % 4 == 0 and Ci > 4, Co % 4 == 0 and Ci % 4 == 0 and Co > 4, Co % 4 == 0 and Ci % 4 == 0 and min(Ci, Co) > 4]
    assert ("smallk" in names[0]) == (Ci <= 4) and ("smallk" in names[1]) == (Co <= 4) and ("smallk" in names[2]) == (min(Ci, Co) <= 4), names
    assert [("bf16" in n) for n in names] == expect16, names
    rf, rd, rw = [(_r16 if e else (lambda t: t)) for e in expect16]
    assert _rel(_nchw(y), _ref_conv(rf(x), rf(w), b, kind)) < 2e-5
    xr = x.clone().requires_grad_(True)
    (gx,) = torch.autograd.grad(_ref_conv(xr, rd(w), None, kind), xr, rd(dy))
    assert _rel(_nchw(dx), gx) < 2e-5
    wr = w.clone().requires_grad_(True)
    _ref_conv(rw(x), wr, None, kind).backward(rw(dy))
    assert _rel(gw.cpu(), wr.grad) < 3e-5
    assert _rel(gb.cpu(), dy.sum(dim=(0, 2, 3))) < 3e-5  # bias gradient: fp32 sums of the unrounded dY on either path


@pytest.mark.parametrize("C,H,W,silu", [(128, 16, 16, True), (512, 4, 4, True), (512, 6, 10, False)])
def test_bf16_flat_fused_gn(cuda, bf16_mode, C, H, W, silu):
    """GroupNorm(+SiLU) fused into the bf16 flat rows / wgrad kernels (1x1 and small 3x3)."""
    from vaehip import ops
    gen = torch.Generator().manual_seed(11 + C)
    B, Co = 2, 128
    x = torch.randn(B, C, H, W, generator=gen) * 1.3 + 0.2
    gamma, beta = 1 + 0.3 * torch.randn(C, generator=gen), 0.2 * torch.randn(C, generator=gen)
    w = torch.randn(Co, C, 3, 3, generator=gen) / math.sqrt(9 * C)
    act = F.group_norm(x, 32, gamma, beta, 1e-6)
    act = F.silu(act) if silu else act
    xf = ops.XF_AFFINE_SILU if silu else ops.XF_AFFINE
    xd = _nhwc(x)
    st = ops.gn_stats(xd, gamma.cuda(), beta.cuda())
    y = ops.conv_fwd(xd, _to_dev_ohwi(w), None, "c3", xf=xf, stats=st)
    assert _rel(_nchw(y), F.conv2d(_r16(act), _r16(w), None, 1, 1)) < 5e-4
    dy = torch.randn(B, Co, H, W, generator=gen)
    wr = w.clone().requires_grad_(True)
    F.conv2d(_r16(act), wr, None, 1, 1).backward(_r16(dy))
    gw = torch.empty_like(_to_dev_ohwi(w).permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
    ops.conv_wgrad(_nhwc(dy), xd, "c3", gw, None, xf=xf, stats=st)
    assert _rel(gw.cpu(), wr.grad) < 5e-4


@pytest.mark.parametrize("z,M,N,K", [(2, 64, 64, 512), (1, 256, 512, 256), (3, 100, 36, 40)])
def test_bf16_batched_gemms(cuda, bf16_mode, z, M, N, K):
    """attention GEMM forms in bf16 mode: operands rounded to bf16, exact products, fp32 accumulation"""
    from vaehip import ops
    gen = torch.Generator().manual_seed(z * 1000 + M)
    A = torch.randn(z, M, K, generator=gen)
    Bt = torch.randn(z, N, K, generator=gen)
    Bn = torch.randn(z, K, N, generator=gen)
    At = torch.randn(z, K, M, generator=gen)
    assert _rel(ops.gemm_nt(A.cuda(), Bt.cuda(), 0.5), 0.5 * _r16(A) @ _r16(Bt).transpose(1, 2)) < 2e-5
    assert _rel(ops.gemm_nn(A.cuda(), Bn.cuda()), _r16(A) @ _r16(Bn)) < 2e-5
    assert _rel(ops.gemm_tn(At.cuda(), Bn.cuda(), 2.0), 2.0 * _r16(At).transpose(1, 2) @ _r16(Bn)) < 2e-5


@pytest.mark.parametrize("B,C,H,W,Co", [(2, 128, 8, 32, 256), (6, 128, 128, 128, 128), (1, 256, 4, 64, 128)])
def test_bf16_activation_image(cuda, bf16_mode, packed_weights, B, C, H, W, Co):
    """bf16 mode, large 3x3 layers: silu(gn(x)) is rounded to bf16 ONCE (vae_gn_apply_bf16) and the forward and the
    weight gradient read that image (A16 / X16) instead of transforming the fp32 input while staging it."""
    from vaehip import ops
    gen = torch.Generator().manual_seed(17 + C + H)
    x = torch.randn(B, C, H, W, generator=gen) * 1.3 + 0.2
    gamma, beta = 1 + 0.3 * torch.randn(C, generator=gen), 0.2 * torch.randn(C, generator=gen)
    w = torch.randn(Co, C, 3, 3, generator=gen) / math.sqrt(9 * C)
    act = F.silu(F.group_norm(x, 32, gamma, beta, 1e-6))
    xd, wd = _nhwc(x), packed_weights(_to_dev_ohwi(w))
    assert ops.act_image_ok("c3", xd.shape, Co, C)
    assert not ops.act_image_ok("c3", (B, 5, 7, C), Co, C) and not ops.act_image_ok("c1", xd.shape, Co, C)
    st = ops.gn_stats(xd, gamma.cuda(), beta.cuda())
    a16 = ops.gn_apply_bf16(xd, st, ops.XF_AFFINE_SILU)
    # the image is the bf16 rounding of the fp32 transform (one bf16 ulp where GPU / CPU fp32 differ in the last bit)
    assert _rel(a16.float().permute(0, 3, 1, 2), _r16(act)) < 8e-3
    img = a16.float().permute(0, 3, 1, 2).cpu()           # what the kernels must contract, exactly
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        y = ops.conv_fwd(xd, wd, None, "c3", xf=ops.XF_AFFINE_SILU, stats=st, a16=a16)
        dy = torch.randn(B, Co, H, W, generator=gen)
        gw = torch.empty_like(wd.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
        gb = torch.empty(Co, device="cuda")
        ops.conv_wgrad(_nhwc(dy), xd, "c3", gw, gb, xf=ops.XF_AFFINE_SILU, stats=st, x16=a16)
    finally:
        ops.PROFILER = None
    names = [r[0] for r in prof.records]
    assert names[0] == "conv3_wide_bf16_kernel<false,3>" or (names[0].startswith("conv3_tile_bf16_kernel") and names[0].endswith(",true>")), names
    assert names[1].startswith("wgrad3_tile_bf16_kernel") and ",true," in names[1], names  # <UP, XF, X16 = true, Y16>
    assert _rel(_nchw(y), F.conv2d(img, _r16(w), None, 1, 1)) < 2e-5
    wr = w.clone().requires_grad_(True)
    F.conv2d(img, wr, None, 1, 1).backward(_r16(dy))
    assert _rel(gw.cpu(), wr.grad) < 3e-5
    assert _rel(gb.cpu(), dy.sum(dim=(0, 2, 3))) < 3e-5


@pytest.mark.parametrize("mode", ["no", "bf16"])
@pytest.mark.parametrize("kind,B,H,W,Ci,Co", [("c3", 2, 8, 32, 128, 128), ("c3", 1, 4, 64, 256, 512), ("c3up", 2, 4, 16, 128, 256)])
def test_groupnorm_statistics_from_conv_epilogue(cuda, packed_weights, mode, kind, B, H, W, Ci, Co):
    """the halo-tile conv kernels leave the GroupNorm partial sums of their OUTPUT (bias and residual included); gn_stats
    on that tensor then only runs the final pass.  Same mean / rstd / scale / shift as the separate pass over the tensor."""
    from vaehip import ops
    gen = torch.Generator().manual_seed(23 + Co + H)
    x = torch.randn(B, Ci, H, W, generator=gen)
    w = torch.randn(Co, Ci, 3, 3, generator=gen) / math.sqrt(9 * Ci)
    b = torch.randn(Co, generator=gen)
    gamma, beta = (1 + 0.3 * torch.randn(Co, generator=gen)).cuda(), (0.2 * torch.randn(Co, generator=gen)).cuda()
    ho, wo = (2 * H, 2 * W) if kind == "c3up" else (H, W)
    res = torch.randn(B, Co, ho, wo, generator=gen)
    wd = _to_dev_ohwi(w)
    ops.PRECISION = ops.PREC_BF16 if mode == "bf16" else ops.PREC_F32
    try:
        if mode == "bf16":
            packed_weights(wd)
        y = ops.conv_fwd(_nhwc(x), wd, b.cuda(), kind, res=_nhwc(res), gstat_groups=32)
        assert getattr(y, "_gstat", None) is not None and y._gstat[1] == 32
        st_fused = ops.gn_stats(y, gamma, beta)
        plain = y.clone()                      # a tensor without the attached partial sums: the ordinary two-pass path
        st_ref = ops.gn_stats(plain, gamma, beta)
        for a_, b_ in zip(st_fused, st_ref):
            assert _rel(a_, b_) < 2e-6
        yr = F.group_norm(_nchw(y).float(), 32, gamma.cpu(), beta.cpu(), 1e-6)  # (bf16 mode stores y as bf16: its values as stored)
        got = ops.gn_apply(y, st_fused, ops.XF_AFFINE)
        assert _rel(_nchw(got), yr) < 2e-5
        # no epilogue for shapes the tile kernels do not serve: gn_stats falls back silently
        y2 = ops.conv_fwd(_nhwc(x)[:, :5, :7].contiguous(), wd, b.cuda(), "c3", gstat_groups=32)
        assert getattr(y2, "_gstat", None) is None
    finally:
        ops.PRECISION = ops.PREC_F32


def test_upconv_phase_decomposition(cuda):
    """conv3x3(nearest_upsample_2x(x)) == four phase convolutions on x with the effective kernels of
    vae_upconv_phase_weights: checks the effective kernels themselves and that the phase path is the one that runs."""
    from vaehip import ops
    gen = torch.Generator().manual_seed(77)
    B, H, W, Ci, Co = 2, 4, 32, 128, 128
    x = torch.randn(B, Ci, H, W, generator=gen)
    w = torch.randn(Co, Ci, 3, 3, generator=gen) / math.sqrt(9 * Ci)
    we = ops.upconv_phase_weights(_to_dev_ohwi(w).permute(0, 2, 3, 1)).cpu()      # [4, Co, 3, 3, Ci]
    y_ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, 1, 1)
    for pa in (0, 1):
        for pb in (0, 1):
            wp = we[pa * 2 + pb].permute(0, 3, 1, 2)                                 # OIHW
            mask = torch.zeros(3, 3)
            for kh in ops._PHASE_MASK[pa]:
                for kw in ops._PHASE_MASK[pb]:
                    mask[kh, kw] = 1
            assert float((wp * (1 - mask)).abs().max()) == 0.0                        # zero outside the 2x2 support
            assert _rel(F.conv2d(x, wp, None, 1, 1), y_ref[:, :, pa::2, pb::2]) < 2e-6
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        with ops.option("no_wino"):  # (round 3: the default fp32 path of the upsampler is the 9-position scheme, tested below)
            y = ops.conv_fwd(_nhwc(x), _to_dev_ohwi(w), None, "c3up")
            dy = torch.randn(B, Co, 2 * H, 2 * W, generator=gen)
            dx = ops.conv_dgrad(_nhwc(dy), _to_dev_ohwi(w), "c3up", (H, W))
            gw = torch.full_like(_to_dev_ohwi(w).permute(0, 2, 3, 1).contiguous(), float("nan")).permute(0, 3, 1, 2)
            gb = torch.full((Co,), float("nan"), device="cuda")
            ops.conv_wgrad(_nhwc(dy), _nhwc(x), "c3up", gw, gb)
    finally:
        ops.PROFILER = None
    assert len(prof.records) == 12 and all(r[0].startswith("conv3_tile_kernel") for r in prof.records[:8])
    assert all(r[0].startswith("wgrad3_tile_kernel") for r in prof.records[8:])
    assert _rel(_nchw(y), y_ref) < 2e-5
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(F.interpolate(xr, scale_factor=2.0, mode="nearest"), wr, None, 1, 1).backward(dy)
    assert _rel(_nchw(dx), xr.grad) < 2e-5
    assert _rel(gw.cpu(), wr.grad) < 3e-5
    assert _rel(gb.cpu(), dy.sum(dim=(0, 2, 3))) < 3e-5


# (2,4,32,...): the 128-pixel halo-tile kernel with a tap mask; (7,32,64,128,256) and (3,64,64,256,256): >= 192 tiles of 8x32 pixels
# on the low-resolution grid -- the wide-tile kernel's 2x2 tap blocks on a bf16 image of x / of dy (strided views)
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 4, 32, 128, 128), (7, 32, 64, 128, 256), (3, 64, 64, 256, 256)])
def test_bf16_upconv_phase_decomposition(cuda, bf16_mode, packed_weights, B, H, W, Ci, Co):
    """bf16 mode: the upsampler's forward and dgrad as four phase convolutions; the effective kernels (sums of fp32 taps) are
    rounded to bf16 once, so the reference rounds the SUMS, not the taps."""
    from vaehip import ops
    gen = torch.Generator().manual_seed(78)
    x = torch.randn(B, Ci, H, W, generator=gen)
    w = torch.randn(Co, Ci, 3, 3, generator=gen) / math.sqrt(9 * Ci)
    wd = packed_weights(_to_dev_ohwi(w))
    we = ops.upconv_phase_weights(wd.permute(0, 2, 3, 1)).cpu()
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        y = ops.conv_fwd(_nhwc(x), wd, None, "c3up")
        dy = torch.randn(B, Co, 2 * H, 2 * W, generator=gen)
        dx = ops.conv_dgrad(_nhwc(dy), wd, "c3up", (H, W))
    finally:
        ops.PROFILER = None
    names = [r[0] for r in prof.records]
    wide_f = B * (H // 8) * (W // 32) * ((Co + 127) // 128) >= 192 and H % 8 == 0
    wide_d = B * (H // 8) * (W // 32) * ((Ci + 127) // 128) >= 192 and H % 8 == 0
    assert len(names) == 8, names
    assert all(n == "conv3_wide_bf16_kernel<false,2>" if wide_f else n.startswith("conv3_tile_bf16_kernel") for n in names[:4]), names
    assert all(n == "conv3_wide_bf16_kernel<true,2>" if wide_d else n.startswith("conv3_tile_bf16_kernel") for n in names[4:]), names
    y_ref = torch.zeros(B, Co, 2 * H, 2 * W)
    xr = _r16(x).requires_grad_(True)
    for pa in (0, 1):
        for pb in (0, 1):
            y_ref[:, :, pa::2, pb::2] = F.conv2d(xr, _r16(we[pa * 2 + pb].permute(0, 3, 1, 2)), None, 1, 1)
    assert _rel(_nchw(y), y_ref.detach()) < 2e-5
    (gx,) = torch.autograd.grad(sum(F.conv2d(xr, _r16(we[pa * 2 + pb].permute(0, 3, 1, 2)), None, 1, 1).mul(_r16(dy)[:, :, pa::2, pb::2]).sum()
                                    for pa in (0, 1) for pb in (0, 1)), xr)
    assert _rel(_nchw(dx), gx) < 2e-5
    # and close to the fp32 upsample + conv at bf16 accuracy
    assert _rel(_nchw(y), F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, 1, 1)) < 2e-2


# ---------------------------------------------------------------------------------------------------------
# bf16 gradient images (bf16 mode): dgrad outputs stored as bf16 (out_bf16), GroupNorm backward reading a bf16 g and
# writing fp32 and / or bf16, dgrad / wgrad reading the output gradient as a bf16 image (A16 / dY16)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 8, 32, 128, 128), (1, 4, 64, 256, 512), (5, 128, 128, 128, 128), (2, 12, 32, 512, 256)])
def test_bf16_gradient_images_conv(cuda, bf16_mode, packed_weights, B, H, W, Ci, Co):
    from vaehip import ops
    gen = torch.Generator().manual_seed(11 + Ci + Co)
    x = torch.randn(B, Ci, H, W, generator=gen)
    w = torch.randn(Co, Ci, 3, 3, generator=gen) / math.sqrt(Ci * 9)
    dy = torch.randn(B, Co, H, W, generator=gen)
    wd = packed_weights(_to_dev_ohwi(w))
    assert ops.grad_image_ok("c3", (B, H, W, Ci), Co, Ci)
    dyd = _nhwc(dy)
    dy16 = dyd.bfloat16()                       # what gn_bwd(want16) hands over: round to nearest even
    xr = _r16(x).requires_grad_(True)
    (gx,) = torch.autograd.grad(F.conv2d(xr, _r16(w), None, 1, 1), xr, _r16(dy))
    # dgrad: fp32 dy vs bf16 image -> the same numbers (the kernel rounds the fp32 dy exactly like .bfloat16())
    d32 = ops.conv_dgrad(dyd, wd, "c3", (H, W))
    d16 = ops.conv_dgrad(dy16, wd, "c3", (H, W))
    assert d32.dtype == torch.float32 and d16.dtype == torch.float32
    wide = B * (H // 8) * (W // 32) * ((Ci + 127) // 128) >= 192 and H % 8 == 0  # the image goes to the wide-tile kernel there:
    assert torch.equal(d32, d16) if not wide else _rel(d16, d32) < 1e-5          # same sums in another order
    assert _rel(_nchw(d32), gx) < 2e-5 and _rel(_nchw(d16), gx) < 2e-5
    # bf16 output: the fp32 result rounded once
    o16 = ops.conv_dgrad(dy16, wd, "c3", (H, W), out_bf16=True)
    assert o16.dtype == torch.bfloat16 and torch.equal(o16, d16.bfloat16())
    # an fp32 tensor carrying the image uses it too
    dyd2 = dyd.clone()
    dyd2._b16 = dy16
    assert torch.equal(ops.conv_dgrad(dyd2, wd, "c3", (H, W)), d16)
    # wgrad from the image: the weight gradient is bitwise the fp32-dy result; the bias gradient sums the ROUNDED values
    wr = w.clone().requires_grad_(True)
    F.conv2d(_r16(x), wr, None, 1, 1).backward(_r16(dy))
    gw = [torch.empty_like(wd.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2) for _ in range(2)]
    gb = [torch.empty(Co, device="cuda") for _ in range(2)]
    ops.conv_wgrad(dyd, _nhwc(x), "c3", gw[0], gb[0])
    ops.conv_wgrad(dy16, _nhwc(x), "c3", gw[1], gb[1])
    assert torch.equal(gw[0], gw[1]) and _rel(gw[1].cpu(), wr.grad) < 3e-5
    assert _rel(gb[1].cpu(), _r16(dy).sum(dim=(0, 2, 3))) < 3e-5 and _rel(gb[0].cpu(), dy.sum(dim=(0, 2, 3))) < 3e-5
    # a bf16-only gradient on a shape the halo-tile kernels do not serve: the flat kernel reads it as it is (round 3)
    w5 = wd if Co == 128 else _to_dev_ohwi(torch.randn(128, Ci, 3, 3, generator=gen) / math.sqrt(9 * Ci))
    d5 = torch.randn(1, 5, 7, 128, generator=gen).cuda().bfloat16()
    assert torch.equal(ops.conv_dgrad(d5, w5, "c3", (5, 7)), ops.conv_dgrad(d5.float(), w5, "c3", (5, 7)))


@pytest.mark.parametrize("C,H,W,silu,B", [(128, 16, 16, True, 2), (256, 8, 32, True, 3), (512, 6, 10, False, 1)])
def test_bf16_gradient_images_groupnorm_backward(cuda, C, H, W, silu, B):
    from vaehip import ops
    gen = torch.Generator().manual_seed(13 + C)
    x = torch.randn(B, C, H, W, generator=gen) * 1.3 + 0.2
    g = torch.randn(B, C, H, W, generator=gen)
    add = torch.randn(B, C, H, W, generator=gen)
    gamma, beta = 1 + 0.3 * torch.randn(C, generator=gen), 0.2 * torch.randn(C, generator=gen)
    xd, gd, ad = _nhwc(x), _nhwc(g), _nhwc(add)
    st = ops.gn_stats(xd, gamma.cuda(), beta.cuda())
    dg, db = [torch.empty(C, device="cuda") for _ in range(3)], [torch.empty(C, device="cuda") for _ in range(3)]
    g16 = gd.bfloat16()
    ref = ops.gn_bwd(xd, g16.float(), st, gamma.cuda(), beta.cuda(), silu, ad, dg[0], db[0])     # fp32 path on the rounded g
    both = ops.gn_bwd(xd, g16, st, gamma.cuda(), beta.cuda(), silu, ad, dg[1], db[1], want32=True, want16=True)
    only = ops.gn_bwd(xd, g16, st, gamma.cuda(), beta.cuda(), silu, ad, dg[2], db[2], want32=False, want16=True)
    assert both.dtype == torch.float32 and torch.equal(both, ref) and torch.equal(dg[1], dg[0]) and torch.equal(db[1], db[0])
    assert both._b16.dtype == torch.bfloat16 and torch.equal(both._b16, ref.bfloat16())
    assert only.dtype == torch.bfloat16 and torch.equal(only, ref.bfloat16()) and torch.equal(dg[2], dg[0])
    # and against autograd on the rounded g
    xr = x.clone().requires_grad_(True)
    gm, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.group_norm(xr, 32, gm, bt, 1e-6)
    (F.silu(y) if silu else y).backward(_r16(g))
    assert _rel(_nchw(ref), xr.grad + add) < 3e-5 and _rel(dg[0].cpu(), gm.grad) < 3e-5 and _rel(db[0].cpu(), bt.grad) < 3e-5


# ---------------------------------------------------------------------------------------------------------
# wide-tile bf16 kernel (csrc/conv3_wide_bf16.hip): forward from an activation image, dgrad from a gradient image; shapes
# with >= 192 tiles of 8 x 32 pixels x 128 channels.  (5,128,128,...): 320 tiles on 256 persistent workgroups (the cross-tile
# pipeline runs), (3,64,64,256,512): 4 channel tiles per pixel tile, (7,40,96,...): rows not a multiple of 16, N tail-free
# ---------------------------------------------------------------------------------------------------------
WIDE_CASES = [(5, 128, 128, 128, 128), (3, 64, 64, 256, 512), (7, 40, 96, 128, 256), (2, 256, 256, 128, 128), (13, 32, 32, 512, 512)]


@pytest.mark.parametrize("B,H,W,Ci,Co", WIDE_CASES)
def test_bf16_wide_tile_kernel(cuda, bf16_mode, packed_weights, B, H, W, Ci, Co):
    from vaehip import ops
    gen = torch.Generator().manual_seed(17 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=gen) * 1.3 + 0.2
    gamma, beta = 1 + 0.3 * torch.randn(Ci, generator=gen), 0.2 * torch.randn(Ci, generator=gen)
    w = torch.randn(Co, Ci, 3, 3, generator=gen) / math.sqrt(9 * Ci)
    bias = torch.randn(Co, generator=gen)
    res = torch.randn(B, Co, H, W, generator=gen)
    xd = _nhwc(x)
    st = ops.gn_stats(xd, gamma.cuda(), beta.cuda())
    wd = packed_weights(_to_dev_ohwi(w))
    a16 = ops.gn_apply_bf16(xd, st, ops.XF_AFFINE_SILU)
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        y = ops.conv_fwd(xd, wd, bias.cuda(), "c3", xf=ops.XF_AFFINE_SILU, stats=st, res=_nhwc(res), a16=a16, gstat_groups=32)
        dy = torch.randn(B, Co, H, W, generator=gen)
        dy16 = _nhwc(dy).bfloat16()
        dx = ops.conv_dgrad(dy16, wd, "c3", (H, W))
        dx16 = ops.conv_dgrad(dy16, wd, "c3", (H, W), out_bf16=True)
    finally:
        ops.PROFILER = None
    names = [r[0] for r in prof.records]
    # the dgrad's channel tiles are over Ci: it needs its own >= 192 tiles to run on the wide kernel
    dg = "conv3_wide_bf16_kernel<true,3>" if B * (H // 8) * (W // 32) * ((Ci + 127) // 128) >= 192 else "conv3_tile_bf16_kernel<true,false,0,true>"
    # a residual input on a 128-channel contraction stays on the 128-pixel kernel (its second workgroup per CU hides the epilogue)
    fw = "conv3_wide_bf16_kernel<false,3>" if Ci > 128 else "conv3_tile_bf16_kernel<false,false,0,true>"
    assert names == [fw, dg, dg], names
    act = a16.float().permute(0, 3, 1, 2).cpu()            # the image the kernel read: exact bf16 values
    y_ref = F.conv2d(act, _r16(w), bias, 1, 1) + res
    assert _rel(_nchw(y), y_ref) < 2e-5
    # GroupNorm statistics of the output from the epilogue == statistics of the tensor it wrote
    g2, b2 = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
    # chunks: the wide-tile kernel one per 4-row band of a tile, the 128-pixel kernel one per 2-row band
    assert hasattr(y, "_gstat") and y._gstat[2] == ((H // 4) * (W // 32) if Ci > 128 else (H // 2) * (W // 32))
    st_f = ops.gn_stats(y, g2, b2)
    st_p = ops.gn_stats(y.clone(), g2, b2)
    assert _rel(st_f.mean, st_p.mean) < 1e-5 and _rel(st_f.rstd, st_p.rstd) < 1e-5
    xr = torch.zeros(B, Ci, H, W).requires_grad_(True)
    (gx,) = torch.autograd.grad(F.conv2d(xr, _r16(w), None, 1, 1), xr, dy16.float().permute(0, 3, 1, 2).cpu())
    assert dx.dtype == torch.float32 and _rel(_nchw(dx), gx) < 2e-5
    assert dx16.dtype == torch.bfloat16 and torch.equal(dx16, dx.bfloat16())
    # without a residual input every channel count runs on the wide kernel (conv1 of a resnet)
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        y0 = ops.conv_fwd(xd, wd, bias.cuda(), "c3", xf=ops.XF_AFFINE_SILU, stats=st, a16=a16, gstat_groups=32)
    finally:
        ops.PROFILER = None
    assert [r[0] for r in prof.records] == ["conv3_wide_bf16_kernel<false,3>"] and y0._gstat[2] == (H // 4) * (W // 32)
    assert _rel(_nchw(y0), y_ref - res) < 2e-5
    st_f, st_p = ops.gn_stats(y0, g2, b2), ops.gn_stats(y0.clone(), g2, b2)
    assert _rel(st_f.mean, st_p.mean) < 1e-5 and _rel(st_f.rstd, st_p.rstd) < 1e-5
    # the 128-pixel kernel computes the same sums in another order
    with ops.option("no_wide"):
        y2 = ops.conv_fwd(xd, wd, bias.cuda(), "c3", xf=ops.XF_AFFINE_SILU, stats=st, res=_nhwc(res), a16=a16)
        dx2 = ops.conv_dgrad(dy16, wd, "c3", (H, W))
    assert _rel(y2, y) < 1e-5 and _rel(dx2, dx) < 1e-5
    # library option "wide_reserved_cus" (CUs left to RCCL in data-parallel runs): a smaller persistent grid deals the same
    # tiles to fewer workgroups -- bitwise the same tensor and statistics
    with ops.option("wide_reserved_cus", 24):
        assert ops.get_option("wide_reserved_cus") == 24
        y0r = ops.conv_fwd(xd, wd, bias.cuda(), "c3", xf=ops.XF_AFFINE_SILU, stats=st, a16=a16, gstat_groups=32)
    assert ops.get_option("wide_reserved_cus") == 0
    assert torch.equal(y0r, y0) and torch.equal(y0r._gstat[0], y0._gstat[0])


@pytest.mark.parametrize("ratio", [30.0, 1000.0])
def test_groupnorm_statistics_with_large_mean(cuda, ratio):
    """|mean| / std of 30 (plausible for conv outputs of a trained VAE) and 1000: E[x^2] - mean^2 in fp32 would lose
    ratio^2 * 1e-7 of the variance (10 % at 1000); the centred-moment statistics (streaming pass AND conv epilogue) stay at
    fp32 accuracy against a float64 computation."""
    from vaehip import ops
    gen = torch.Generator().manual_seed(31)
    B, C, H, W = 2, 128, 32, 32
    x = torch.randn(B, C, H, W, generator=gen) + ratio * (1 + 0.1 * torch.randn(1, C, 1, 1, generator=gen))
    gamma, beta = torch.ones(C), torch.zeros(C)
    xd = _nhwc(x)
    st = ops.gn_stats(xd, gamma.cuda(), beta.cuda())
    xg = x.double().view(B, 32, -1)
    mean64, var64 = xg.mean(-1), xg.var(-1, unbiased=False)
    rstd64 = 1.0 / torch.sqrt(var64 + 1e-6)
    assert _rel(st.mean, mean64) < 1e-6
    assert float(((st.rstd.cpu().double() - rstd64) / rstd64).abs().max()) < 2e-5
    # the normalised tensor (what the next conv sees) against float64
    y = ops.gn_apply(xd, st, ops.XF_AFFINE)
    ref = ((xg - mean64[..., None]) * rstd64[..., None]).view(B, C, H, W)
    assert float((_nchw(y).double() - ref).abs().max()) < 3e-4 * max(1.0, ratio / 30)  # fp32 x itself carries ratio * 6e-8
    # the conv epilogue: a 1-tap-dominated conv whose output keeps the large mean (bias)
    w = torch.randn(C, C, 3, 3, generator=gen) / math.sqrt(9 * C)
    bias = torch.full((C,), ratio) * (1 + 0.1 * torch.randn(C, generator=gen))
    z = ops.conv_fwd(_nhwc(torch.randn(B, C, H, W, generator=gen)), _to_dev_ohwi(w), bias.cuda(), "c3", gstat_groups=32)
    assert hasattr(z, "_gstat")
    st_f = ops.gn_stats(z, gamma.cuda(), beta.cuda())
    zg = _nchw(z).double().reshape(B, 32, -1)
    r64 = 1.0 / torch.sqrt(zg.var(-1, unbiased=False) + 1e-6)
    assert float(((st_f.rstd.cpu().double() - r64) / r64).abs().max()) < 2e-5
    assert _rel(st_f.mean, zg.mean(-1)) < 1e-6


@pytest.mark.parametrize("B,C,H,W", [(2, 512, 60, 60), (1, 128, 148, 148), (2, 256, 72, 72), (1, 512, 120, 120)])
def test_groupnorm_statistics_ragged_chunk_plans(cuda, B, C, H, W):
    """feature maps whose pixel count the chunk plan does not divide (resolutions 480 / 576 / 960: trailing chunks start
    beyond H*W and are empty): statistics, the normalised tensor and the backward against torch.group_norm"""
    from vaehip import ops
    gen = torch.Generator().manual_seed(5 + H)
    x = (torch.randn(B, C, H, W, generator=gen) * 0.5 + 1.0).requires_grad_(True)
    gamma = (1.0 + 0.2 * torch.randn(C, generator=gen)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=gen)).requires_grad_(True)
    nch = ops._gn_nchunk(B, H * W, C)
    per = -(-H * W // nch)
    xd = _nhwc(x.detach())
    st = ops.gn_stats(xd, gamma.detach().cuda(), beta.detach().cuda())
    xg = x.detach().double().view(B, 32, -1)
    assert _rel(st.mean, xg.mean(-1)) < 1e-6, (nch, per)
    r64 = 1.0 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-6)
    assert float(((st.rstd.cpu().double() - r64) / r64).abs().max()) < 2e-5, (nch, per, (nch - 1) * per >= H * W)
    ref = F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-6))
    assert _rel(_nchw(ops.gn_apply(xd, st, ops.XF_AFFINE_SILU)), ref.detach()) < 1e-5
    g = torch.randn(B, C, H, W, generator=gen)
    ref.backward(g)
    dga, dbe = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dx = ops.gn_bwd(xd, _nhwc(g), st, gamma.detach().cuda(), beta.detach().cuda(), True, None, dga, dbe)
    assert _rel(_nchw(dx), x.grad) < 2e-5 and _rel(dga, gamma.grad) < 2e-5 and _rel(dbe, beta.grad) < 2e-5


# ---------------------------------------------------------------------------------------------------------
# Winograd F(2x2,3x3) (csrc/conv3_wino.hip): fp32 forward (plain, with GroupNorm+SiLU fused on the patch loads, bias, residual)
# and dgrad against torch's CPU convolution AND against the direct halo-tile kernels (library option "no_wino").  The transforms only
# add / subtract / halve: the result must stay at fp32 summation accuracy.  (6,16,48,...): ragged tile counts, N not a multiple
# of 128; (2,8,16,128,32): one tile, the smallest N served.
# ---------------------------------------------------------------------------------------------------------
WINO_CASES = [(2, 32, 32, 128, 128), (1, 64, 64, 256, 128), (2, 16, 32, 512, 512), (6, 16, 48, 128, 160), (2, 8, 16, 128, 32),
              (3, 32, 64, 128, 192), (1, 48, 32, 256, 64)]


def _wino_kernel(xf, H, W, N, f4):
    """name of the Winograd instantiation that serves a 3x3 stride-1 launch with N output channels (csrc/igemm.hip: F(4x4,3x3)
    on whole 16 x 32 tiles with whole 64-channel blocks unless the library option "no_wino4" is set, else F(2x2,3x3))"""
    if f4 and H % 16 == 0 and W % 32 == 0 and N % 64 == 0:
        return f"conv3_wino4_kernel<{xf}>", (H // 16) * (W // 32)
    return f"conv3_wino_kernel<{xf},2>", (H // 8) * (W // 16)


# F(4x4,3x3) (csrc/conv3_wino4.hip, round 4) serves the cases whose maps are whole 16 x 32 tiles; its transforms multiply by up to
# 8 and its weights carry 1/6, 1/24.  Measured on MI355X against a float64 convolution (tools/wino4_accuracy.py, forward and
# dgrad, 128 / 256 / 512 channels): rms 2.3e-6 / 2.9e-6 / 4.5e-6, worst element 1.1e-5 / 1.3e-5 / 2.1e-5 of the tensor's max
# (F(2x2): rms 3.5-6.9e-7, worst 0.4-1e-6; direct: rms 0.6-1.2e-6; the CPU emulation tools/wino_f4_error_study.py predicts the
# same).  _rel is the worst element, so the F(4x4) bars are 2x the measured worst: 4e-5.  "f2" runs every case under the
# library option "no_wino4": the F(2x2) kernel keeps its coverage and its bars.
@pytest.mark.parametrize("algo", ["f4", "f2"])
@pytest.mark.parametrize("B,H,W,Ci,Co", WINO_CASES)
def test_winograd_forward_and_dgrad(cuda, B, H, W, Ci, Co, algo):
    from vaehip import ops
    f4 = algo == "f4"
    if f4 and not (H % 16 == 0 and W % 32 == 0 and (Ci % 64 == 0 or Co % 64 == 0)):
        pytest.skip("no F(4x4) launch in this case (covered by algo = f2)")
    gen = torch.Generator().manual_seed(41 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=gen) * 1.3 + 0.2
    w = torch.randn(Co, Ci, 3, 3, generator=gen) / math.sqrt(9 * Ci)
    bias = torch.randn(Co, generator=gen)
    res = torch.randn(B, Co, H, W, generator=gen)
    gamma, beta = 1 + 0.3 * torch.randn(Ci, generator=gen), 0.2 * torch.randn(Ci, generator=gen)
    xd, wd = _nhwc(x), _to_dev_ohwi(w)
    st = ops.gn_stats(xd, gamma.cuda(), beta.cuda())
    dy = torch.randn(B, Co, H, W, generator=gen)
    with ops.option("no_wino4", 0 if f4 else 1):
        prof = ops.PROFILER = ops.LaunchProfiler()
        try:
            y0 = ops.conv_fwd(xd, wd, bias.cuda(), "c3")
            y1 = ops.conv_fwd(xd, wd, bias.cuda(), "c3", xf=ops.XF_AFFINE_SILU, stats=st, res=_nhwc(res), gstat_groups=32)
            dx = ops.conv_dgrad(_nhwc(dy), wd, "c3", (H, W))
        finally:
            ops.PROFILER = None
        k0, _ = _wino_kernel(0, H, W, Co, f4)
        k1, chunks = _wino_kernel(2, H, W, Co, f4)
        dgk = _wino_kernel(0, H, W, Ci, f4)[0] if Co >= 64 else "igemm_rows_kernel<128,128,4,2,true,true,0>"  # the dgrad contracts over Co
        assert [r[0] for r in prof.records] == [k0, k1, dgk], [r[0] for r in prof.records]
        if f4:
            assert "wino4" in k0 or "wino4" in dgk
        if Co % 128 == 0:  # GroupNorm moments of the output from the epilogue == those of the tensor it wrote
            assert hasattr(y1, "_gstat") and y1._gstat[2] == chunks
            g2, b2 = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
            st_f, st_p = ops.gn_stats(y1, g2, b2), ops.gn_stats(y1.clone(), g2, b2)
            assert _rel(st_f.mean, st_p.mean) < 1e-5 and _rel(st_f.rstd, st_p.rstd) < 1e-5
        # deterministic
        assert torch.equal(ops.conv_fwd(xd, wd, bias.cuda(), "c3"), y0)
    xr = x.clone().requires_grad_(True)
    ref0 = F.conv2d(xr, w, bias, 1, 1)
    (gx,) = torch.autograd.grad(ref0, xr, dy)
    ref1 = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, 1e-6)), w, bias, 1, 1) + res
    tol_t = 4e-5 if f4 else 1e-5
    assert _rel(_nchw(y0), ref0.detach()) < tol_t and _rel(_nchw(y1), ref1) < 2 * tol_t and _rel(_nchw(dx), gx) < tol_t
    with ops.option("no_wino"):
        z0 = ops.conv_fwd(xd, wd, bias.cuda(), "c3")
        z1 = ops.conv_fwd(xd, wd, bias.cuda(), "c3", xf=ops.XF_AFFINE_SILU, stats=st, res=_nhwc(res))
        dz = ops.conv_dgrad(_nhwc(dy), wd, "c3", (H, W))
    tol = 4e-5 if f4 else 5e-6
    assert _rel(y0, z0) < tol and _rel(y1, z1) < tol and _rel(dx, dz) < tol and not torch.equal(y0, z0)
    print(f"{algo} B={B} {H}x{W} {Ci}->{Co}: vs torch {_rel(_nchw(y0), ref0.detach()):.2e} / {_rel(_nchw(dx), gx):.2e}, vs direct {_rel(y0, z0):.2e} / {_rel(dx, dz):.2e}")


@pytest.mark.parametrize("algo", ["f4", "f2"])
@pytest.mark.parametrize("B,H,W,Ci,Co,silu", [(2, 16, 32, 128, 64, True), (1, 24, 16, 256, 128, True), (2, 8, 16, 128, 192, False),
                                              (2, 32, 64, 256, 128, True), (1, 16, 32, 512, 512, False)])
def test_winograd_dgrad_leaves_groupnorm_backward_sums(cuda, B, H, W, Ci, Co, silu, algo):
    """dgrad epilogue with gnb_*: the per-tile sums it leaves make gn_bwd (without its first pass) return what the three-pass
    form returns from the same tensors, and what autograd returns for silu(gn(x)) -> conv; both Winograd kernels"""
    from vaehip import ops
    f4 = algo == "f4"
    kname, chunks = _wino_kernel(0, H, W, Ci, f4)
    if f4 and "wino4" not in kname:
        pytest.skip("no F(4x4) launch in this case (covered by algo = f2)")
    gen = torch.Generator().manual_seed(47 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=gen) * 1.3 + 0.2
    w = torch.randn(Co, Ci, 3, 3, generator=gen) / math.sqrt(9 * Ci)
    dy = torch.randn(B, Co, H, W, generator=gen)
    gamma, beta = 1 + 0.3 * torch.randn(Ci, generator=gen), 0.2 * torch.randn(Ci, generator=gen)
    xd, wd, dyd = _nhwc(x), _to_dev_ohwi(w), _nhwc(dy)
    gd, bd = gamma.cuda(), beta.cuda()
    st = ops.gn_stats(xd, gd, bd)
    ctx = ops.GnCtx(xd, st, gd, bd, silu, 32)

    def bwd(dA):
        dg, db = torch.full((Ci,), float("nan"), device="cuda"), torch.full((Ci,), float("nan"), device="cuda")
        return ops.gn_bwd(xd, dA, st, gd, bd, silu, None, dg, db), dg, db

    with ops.option("no_wino4", 0 if f4 else 1):
        prof = ops.PROFILER = ops.LaunchProfiler()
        try:
            dA_f = ops.conv_dgrad(dyd, wd, "c3", (H, W), gnb=ctx)
        finally:
            ops.PROFILER = None
        assert [r[0] for r in prof.records] == [kname]
        assert hasattr(dA_f, "_gnb") and dA_f._gnb[1] == chunks
        dA_p = ops.conv_dgrad(dyd, wd, "c3", (H, W))
        assert torch.equal(dA_f, dA_p) and not hasattr(dA_p, "_gnb")  # the epilogue does not touch the gradient itself
        (dx_f, dg_f, db_f), (dx_p, dg_p, db_p) = bwd(dA_f), bwd(dA_p)
        assert _rel(dx_f, dx_p) < 2e-6 and _rel(dg_f, dg_p) < 2e-6 and _rel(db_f, db_p) < 2e-6
        assert torch.equal(bwd(ops.conv_dgrad(dyd, wd, "c3", (H, W), gnb=ctx))[0], dx_f)  # deterministic
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    h = F.group_norm(xr, 32, gr, br, 1e-6)
    F.conv2d(F.silu(h) if silu else h, w, None, 1, 1).backward(dy)
    assert _rel(_nchw(dx_f), xr.grad) < 2e-5 and _rel(dg_f.cpu(), gr.grad) < 2e-5 and _rel(db_f.cpu(), br.grad) < 2e-5
    # a gradient tensor that is not this GroupNorm's (another x) falls back to the first pass
    other = xd.clone()
    dg, db = torch.empty(Ci, device="cuda"), torch.empty(Ci, device="cuda")
    assert torch.equal(ops.gn_bwd(other, dA_f, st, gd, bd, silu, None, dg, db), dx_p)


# (B,H,W,Ci,Co): several strips per row and rows per image, several images per split, Ci / Co blocks, a single unit row
WINO_WGRAD_CASES = [(2, 32, 32, 128, 128), (1, 64, 64, 256, 128), (3, 16, 32, 512, 256), (5, 6, 48, 128, 128), (1, 2, 16, 128, 128)]


@pytest.mark.parametrize("B,H,W,Ci,Co", WINO_WGRAD_CASES)
def test_winograd_wgrad(cuda, B, H, W, Ci, Co):
    """F(3x3,2x2) weight gradient (wgrad3_wino.hip) against autograd and against the direct halo-tile kernel, plain and behind
    GroupNorm+SiLU, with the bias gradient riding along"""
    from vaehip import ops
    gen = torch.Generator().manual_seed(43 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=gen) * 1.3 + 0.2
    dy = torch.randn(B, Co, H, W, generator=gen)
    gamma, beta = 1 + 0.3 * torch.randn(Ci, generator=gen), 0.2 * torch.randn(Ci, generator=gen)
    xd, dyd = _nhwc(x), _nhwc(dy)
    st = ops.gn_stats(xd, gamma.cuda(), beta.cuda())

    def run(xf):
        gw = torch.full((Co, 3, 3, Ci), float("nan"), device="cuda")
        gb = torch.full((Co,), float("nan"), device="cuda")
        ops.conv_wgrad(dyd, xd, "c3", gw.permute(0, 3, 1, 2), gb, xf=xf, stats=st if xf else None)
        return gw, gb

    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        gw0, gb0 = run(ops.XF_NONE)
        gw1, gb1 = run(ops.XF_AFFINE_SILU)
    finally:
        ops.PROFILER = None
    assert [r[0] for r in prof.records if r[1] > 0] == ["wgrad3_wino_kernel<0>", "wgrad3_wino_kernel<2>"], [r[0] for r in prof.records]
    assert sum(r[0].startswith("wgrad_wino_reduce") for r in prof.records) == 2  # the split sum + output transform is timed too
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(x, w, torch.zeros(Co), 1, 1).backward(dy)
    ref0 = w.grad.clone()
    w.grad = None
    F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, 1e-6)), w, None, 1, 1).backward(dy)
    ref1 = w.grad.clone()
    refb = dy.sum(dim=(0, 2, 3))
    assert _rel(gw0.permute(0, 3, 1, 2).cpu(), ref0) < 2e-5 and _rel(gw1.permute(0, 3, 1, 2).cpu(), ref1) < 2e-5
    assert _rel(gb0.cpu(), refb) < 1e-5 and torch.equal(gb0, gb1)
    with ops.option("no_wino"):
        dw0, db0 = run(ops.XF_NONE)
        dw1, _ = run(ops.XF_AFFINE_SILU)
    assert _rel(gw0, dw0) < 2e-5 and _rel(gw1, dw1) < 2e-5 and _rel(gb0, db0) < 1e-5 and not torch.equal(gw0, dw0)
    g2, _ = run(ops.XF_NONE)  # deterministic
    assert torch.equal(g2, gw0)


# ---------------------------------------------------------------------------------------------------------
# Upsampler convolution conv3x3(nearest_upsample_2x(x)) in fp32 with 9 multiplications per low-resolution pixel and channel pair
# (csrc/conv3_upwino.hip, csrc/wgrad3_upwino.hip): forward, dgrad (3x3 dgrad + 2x2 sum-pool in one pass) and wgrad against
# torch's interpolate + conv2d and autograd, and against the phase-convolution path (library option "no_wino").
# (3,4,8,512,64): one tile per image, 64 chunk steps; (2,12,24,128,160): N not a multiple of 64; (1,32,32,128,128): 32 tiles
# ---------------------------------------------------------------------------------------------------------
UPWINO_CASES = [(2, 8, 16, 128, 128), (1, 16, 32, 256, 128), (3, 4, 8, 512, 64), (2, 12, 24, 128, 160), (1, 32, 32, 128, 128)]


@pytest.mark.parametrize("B,H,W,Ci,Co", UPWINO_CASES)
def test_upsampler_winograd_forward_and_dgrad(cuda, B, H, W, Ci, Co):
    from vaehip import ops
    gen = torch.Generator().manual_seed(51 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=gen) * 1.3 + 0.2
    w = torch.randn(Co, Ci, 3, 3, generator=gen) / math.sqrt(9 * Ci)
    bias = torch.randn(Co, generator=gen)
    dy = torch.randn(B, Co, 2 * H, 2 * W, generator=gen)
    xd, wd = _nhwc(x), _to_dev_ohwi(w)
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        y = ops.conv_fwd(xd, wd, bias.cuda(), "c3up")
        dx = ops.conv_dgrad(_nhwc(dy), wd, "c3up", (H, W))
    finally:
        ops.PROFILER = None
    assert [r[0] for r in prof.records] == ["conv3_upwino_kernel<false>", "conv3_upwino_kernel<true>"], [r[0] for r in prof.records]
    assert all(r[2] * 4 == r[1] for r in prof.records)  # 9 of 36 multiplications executed
    xr = x.clone().requires_grad_(True)
    ref = F.conv2d(F.interpolate(xr, scale_factor=2.0, mode="nearest"), w, bias, 1, 1)
    (gx,) = torch.autograd.grad(ref, xr, dy)
    assert y.shape == (B, 2 * H, 2 * W, Co) and _rel(_nchw(y), ref.detach()) < 1e-5 and _rel(_nchw(dx), gx) < 1e-5
    with ops.option("no_wino"):  # the four phase convolutions on the direct kernels
        z = ops.conv_fwd(xd, wd, bias.cuda(), "c3up")
        dz = ops.conv_dgrad(_nhwc(dy), wd, "c3up", (H, W))
    assert _rel(y, z) < 5e-6 and _rel(dx, dz) < 5e-6 and not torch.equal(y, z)
    assert torch.equal(ops.conv_fwd(xd, wd, bias.cuda(), "c3up"), y)  # deterministic


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 8, 16, 128, 128), (1, 16, 32, 256, 128), (3, 4, 8, 512, 256), (5, 6, 24, 128, 256), (1, 1, 8, 128, 128)])
def test_upsampler_winograd_wgrad(cuda, B, H, W, Ci, Co):
    """(5,6,24,...): several strips per row, images per split; (1,1,8,...): a single unit"""
    from vaehip import ops
    gen = torch.Generator().manual_seed(61 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=gen) * 1.3 + 0.2
    w = torch.randn(Co, Ci, 3, 3, generator=gen, requires_grad=True)
    dy = torch.randn(B, Co, 2 * H, 2 * W, generator=gen)
    xd, dyd = _nhwc(x), _nhwc(dy)

    def run():
        gw = torch.full((Co, 3, 3, Ci), float("nan"), device="cuda").permute(0, 3, 1, 2)
        gb = torch.full((Co,), float("nan"), device="cuda")
        ops.conv_wgrad(dyd, xd, "c3up", gw, gb)
        return gw, gb
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        gw, gb = run()
    finally:
        ops.PROFILER = None
    assert [r[0] for r in prof.records if r[1] > 0] == ["wgrad3_upwino_kernel"], [r[0] for r in prof.records]
    F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, 1, 1).backward(dy)
    assert _rel(gw.cpu(), w.grad) < 2e-5 and _rel(gb.cpu(), dy.sum(dim=(0, 2, 3))) < 1e-5
    with ops.option("no_wino"):
        dw, db = run()
    assert _rel(gw, dw) < 2e-5 and _rel(gb, db) < 1e-5 and not torch.equal(gw, dw)
    g2, _ = run()  # deterministic
    assert torch.equal(g2, gw)



def test_flat_kernels_source_pixel_mapping_on_the_device(cuda, tmp_path):
    """common.h SrcMap (one branch-free affine form for the four geometries, what the flat kernels call per operand load) against
    the mode-switch form, ON THE DEVICE: a first version of its builder compiled correctly for the host and wrongly for gfx950
    (hipcc 7.2: the DGRAD arm came out with the UP2X constants), which only a device comparison shows"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "srcmap_check")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "srcmap_check.hip"), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("mode")]
    assert out.returncode == 0 and len(lines) == 11 and all(ln.endswith("bad 0") for ln in lines), out.stdout + out.stderr
