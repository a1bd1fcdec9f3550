"""bf16 STORAGE of activations and gradients in bf16 mode (round 3: what autocast keeps for the reference, src/train.py:147-154):
every kernel that reads or writes an activation tensor, with that tensor stored as bf16, against the SAME kernel on the fp32
copy of the same values.  A bf16 operand is converted to fp32 on load and runs through the same arithmetic in the same order,
and a bf16 result is the fp32 result rounded once -- so the comparisons are bitwise wherever the same kernel serves both calls.
The arithmetic itself is pinned by tests/test_kernels_gpu.py (fp32 storage, against torch's CPU ops)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def _to_dev_ohwi(w):
    return w.permute(0, 2, 3, 1).contiguous().cuda().permute(0, 3, 1, 2)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture
def act16(cuda):
    """bf16 arithmetic + bf16 activation storage + a bf16 image of every weight handed over (what the engine sets up)"""
    from vaehip import ops
    keep = []
    ops.PRECISION = ops.PREC_BF16
    assert ops.ACT_BF16

    def pack(wd):
        buf = wd.permute(0, 2, 3, 1) if wd.ndim == 4 else wd
        assert buf.is_contiguous()
        img = torch.empty(buf.numel(), device="cuda", dtype=torch.bfloat16)
        ops.pack_bf16(buf, img)
        keep.append(img)
        ops.WEIGHTS16 = (buf.data_ptr(), buf.numel() * 4, img.data_ptr())
        return wd
    yield pack
    ops.PRECISION, ops.WEIGHTS16 = ops.PREC_F32, None


def _names(prof):
    return [r[0] for r in prof.records]


# kind, B, H, W, Ci, Co, kernel family expected for the forward
FWD_CASES = [("c3", 2, 8, 32, 128, 128, "conv3_tile_bf16"),      # 128-pixel halo-tile kernel (residual on a 128-channel contraction)
             ("c3", 3, 64, 64, 256, 512, "conv3_wide_bf16"),     # wide-tile kernel, 4 channel tiles
             ("c3", 7, 40, 96, 256, 256, "conv3_wide_bf16"),
             ("c3", 1, 5, 7, 128, 128, "igemm_rows_bf16"),       # ragged: flat kernel
             ("c1", 2, 16, 16, 256, 128, "igemm_rows_bf16"),
             ("c3s2", 2, 32, 32, 128, 128, "igemm_rows_bf16")]


@pytest.mark.parametrize("kind,B,H,W,Ci,Co,family", FWD_CASES)
def test_conv_forward_and_dgrad_with_bf16_storage(act16, kind, B, H, W, Ci, Co, family):
    from vaehip import ops
    gen = torch.Generator().manual_seed(3 + Ci + Co + H)
    k = 1 if kind == "c1" else 3
    x16 = _nhwc(torch.randn(B, Ci, H, W, generator=gen) * 1.2 + 0.1).bfloat16()
    w = torch.randn(Co, Ci, k, k, generator=gen) / math.sqrt(Ci * k * k)
    wd = act16(_to_dev_ohwi(w))
    bias = torch.randn(Co, generator=gen).cuda()
    Ho, Wo = ops.out_hw(kind, H, W)
    res16 = torch.randn(B, Ho, Wo, Co, generator=gen).cuda().bfloat16()
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        y16 = ops.conv_fwd(x16, wd, bias, kind, res=res16, gstat_groups=32)
        # the same kernel on the same operand values with fp32 output / residual
        y32 = ops.conv_fwd(x16.float(), wd, bias, kind, res=res16.float(), a16=x16 if family != "igemm_rows_bf16" else None,
                           gstat_groups=32, out_dtype=torch.float32)
    finally:
        ops.PROFILER = None
    assert y16.dtype == torch.bfloat16 and y32.dtype == torch.float32
    assert all(n.startswith(family) for n in _names(prof)), _names(prof)
    assert torch.equal(y16, y32.bfloat16()), _rel(y16.float(), y32)
    # against torch on the rounded operands (the arithmetic), at the accuracy of one bf16 rounding of the result
    xr = x16.float().cpu().permute(0, 3, 1, 2)
    wr = w.bfloat16().float()
    if kind == "c3s2":
        ref = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wr, bias.cpu(), 2, 0)
    else:
        ref = F.conv2d(xr, wr, bias.cpu(), 1, k // 2)
    ref = ref + res16.float().cpu().permute(0, 3, 1, 2)
    assert _rel(y16.float().permute(0, 3, 1, 2), ref) < 6e-3  # 2^-8 relative per element
    # GroupNorm statistics from the epilogue describe the tensor AS STORED
    if hasattr(y16, "_gstat"):
        g1, b1 = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
        st_f, st_p = ops.gn_stats(y16, g1, b1), ops.gn_stats(y16.clone(), g1, b1)
        assert _rel(st_f.mean, st_p.mean) < 1e-5 and _rel(st_f.rstd, st_p.rstd) < 1e-5
    else:
        assert family == "igemm_rows_bf16"
    # dgrad: a bf16 gradient in, a bf16 input gradient out = the fp32 result rounded once
    dy16 = torch.randn(B, Ho, Wo, Co, generator=gen).cuda().bfloat16()
    d16 = ops.conv_dgrad(dy16, wd, kind, (H, W))
    d32 = ops.conv_dgrad(dy16, wd, kind, (H, W), out_dtype=torch.float32)
    assert d16.dtype == torch.bfloat16 and d32.dtype == torch.float32 and torch.equal(d16, d32.bfloat16())
    xg = xr.clone().requires_grad_(True)
    yy = F.conv2d(F.pad(xg, (0, 1, 0, 1)), wr, None, 2, 0) if kind == "c3s2" else F.conv2d(xg, wr, None, 1, k // 2)
    (gx,) = torch.autograd.grad(yy, xg, dy16.float().cpu().permute(0, 3, 1, 2))
    assert _rel(d32.permute(0, 3, 1, 2), gx) < 3e-5
    # wgrad: both operands bf16 tensors == both operands fp32 copies (rounded identically inside the kernel)
    gw = [torch.empty_like(wd.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2) for _ in range(2)]
    gb = [torch.empty(Co, device="cuda") for _ in range(2)]
    ops.conv_wgrad(dy16, x16, kind, gw[0], gb[0])
    ops.conv_wgrad(dy16.float(), x16.float(), kind, gw[1], gb[1])
    assert _rel(gw[0], gw[1]) < 2e-5 and _rel(gb[0], gb[1]) < 1e-6  # (the halo-tile / flat choice may differ between the two calls)
    wg = wr.clone().requires_grad_(True)
    yy = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wg, None, 2, 0) if kind == "c3s2" else F.conv2d(xr, wg, None, 1, k // 2)
    yy.backward(dy16.float().cpu().permute(0, 3, 1, 2))
    assert _rel(gw[0].cpu(), wg.grad) < 3e-5


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 128, 128, 256, 128), (2, 128, 128, 128, 256), (3, 128, 96, 512, 256), (8, 64, 64, 256, 512)])
def test_streaming_1x1_convolution_with_resident_weights(act16, B, H, W, Ci, Co):
    """conv1_bf16.hip (the shortcut convolutions at full size: weights resident in LDS, A straight into MFMA fragments) against
    the flat kernel on the same bf16 tensors and against torch on the rounded operands, forward and dgrad"""
    from vaehip import ops
    gen = torch.Generator().manual_seed(11 + Ci + Co + H)
    x16 = _nhwc(torch.randn(B, Ci, H, W, generator=gen) * 1.2 + 0.1).bfloat16()
    w = torch.randn(Co, Ci, 1, 1, generator=gen) / math.sqrt(Ci)
    wd = act16(_to_dev_ohwi(w))
    bias = torch.randn(Co, generator=gen).cuda()
    dy16 = torch.randn(B, H, W, Co, generator=gen).cuda().bfloat16()
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        y = ops.conv_fwd(x16, wd, bias, "c1")
        d = ops.conv_dgrad(dy16, wd, "c1", (H, W))
    finally:
        ops.PROFILER = None
    dgk = f"conv1_bf16_kernel<true,{Co // 16}>" if Co <= 256 else "igemm_rows_bf16_kernel<128,128,4,2,true,0>"  # (a 512-row [k][n] slice does not fit LDS)
    assert _names(prof) == [f"conv1_bf16_kernel<false,{Ci // 16}>", dgk], _names(prof)
    assert y.dtype == torch.bfloat16 and d.dtype == torch.bfloat16
    with ops.option("flat_conv"):
        yf = ops.conv_fwd(x16, wd, bias, "c1")
        df = ops.conv_dgrad(dy16, wd, "c1", (H, W))
    assert _rel(y.float(), yf.float()) < 4e-3 and _rel(d.float(), df.float()) < 4e-3  # at most one bf16 ulp apart
    xr, wr = x16.float().cpu().permute(0, 3, 1, 2), w.bfloat16().float()
    assert _rel(y.float().cpu().permute(0, 3, 1, 2), F.conv2d(xr, wr, bias.cpu())) < 6e-3
    xg = xr.clone().requires_grad_(True)
    (gx,) = torch.autograd.grad(F.conv2d(xg, wr), xg, dy16.float().cpu().permute(0, 3, 1, 2))
    assert _rel(d.float().cpu().permute(0, 3, 1, 2), gx) < 6e-3
    assert torch.equal(ops.conv_fwd(x16, wd, bias, "c1"), y)  # deterministic


@pytest.mark.parametrize("C,H,W,silu", [(128, 16, 16, True), (512, 6, 10, False), (256, 8, 32, True)])
def test_groupnorm_kernels_on_bf16_storage(cuda, C, H, W, silu):
    """statistics, apply, image, tracker and backward read a bf16 x exactly as they read its fp32 copy"""
    from vaehip import ops
    gen = torch.Generator().manual_seed(7 + C)
    B = 2
    x16 = _nhwc(torch.randn(B, C, H, W, generator=gen) * 1.3 + 0.4).bfloat16()
    x32 = x16.float()
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=gen)).cuda(), (0.2 * torch.randn(C, generator=gen)).cuda()
    s16, s32 = ops.gn_stats(x16, gamma, beta), ops.gn_stats(x32, gamma, beta)
    assert torch.equal(s16.mean, s32.mean) and torch.equal(s16.rstd, s32.rstd) and torch.equal(s16.scale, s32.scale)
    xf = ops.XF_AFFINE_SILU if silu else ops.XF_AFFINE
    assert torch.equal(ops.gn_apply(x16, s16, xf), ops.gn_apply(x32, s32, xf))
    assert torch.equal(ops.gn_apply_bf16(x16, s16, xf), ops.gn_apply_bf16(x32, s32, xf))
    assert torch.equal(ops.gn_track(x16, s16), ops.gn_track(x32, s32))
    g16 = torch.randn(B, H, W, C, generator=gen).cuda().bfloat16()
    add16 = torch.randn(B, H, W, C, generator=gen).cuda().bfloat16()
    dg, db = [torch.empty(C, device="cuda") for _ in range(2)], [torch.empty(C, device="cuda") for _ in range(2)]
    d16 = ops.gn_bwd(x16, g16, s16, gamma, beta, silu, add16, dg[0], db[0], want32=False, want16=True)
    d32 = ops.gn_bwd(x32, g16, s32, gamma, beta, silu, add16.float(), dg[1], db[1], want32=True, want16=False)
    assert d16.dtype == torch.bfloat16 and torch.equal(d16, d32.bfloat16()) and torch.equal(dg[0], dg[1]) and torch.equal(db[0], db[1])
    # an fp32 gradient with bf16 x / add (the attention block's GroupNorm backward)
    d16b = ops.gn_bwd(x16, g16.float(), s16, gamma, beta, silu, add16, dg[0], db[0], want32=False, want16=True)
    assert torch.equal(d16b, d16)
    # add / unpack helpers
    assert torch.equal(ops.add(x16, add16), (x32 + add16.float()).bfloat16()) and torch.equal(ops.to_f32(x16), x32)


def test_fused_transform_on_bf16_storage(act16):
    """flat kernels: GroupNorm(+SiLU) applied on load to an operand stored as bf16 (a_bf16 / x_bf16), 1x1 and stride-2 layers"""
    from vaehip import ops
    gen = torch.Generator().manual_seed(17)
    B, C, H, W, Co = 2, 256, 16, 16, 128
    x16 = _nhwc(torch.randn(B, C, H, W, generator=gen) + 0.3).bfloat16()
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=gen)).cuda(), (0.2 * torch.randn(C, generator=gen)).cuda()
    st = ops.gn_stats(x16, gamma, beta)
    for kind, k in (("c1", 1), ("c3s2", 3)):
        wd = act16(_to_dev_ohwi(torch.randn(Co, C, k, k, generator=gen) / math.sqrt(C * k * k)))
        y16 = ops.conv_fwd(x16, wd, None, kind, xf=ops.XF_AFFINE_SILU, stats=st)
        y32 = ops.conv_fwd(x16.float(), wd, None, kind, xf=ops.XF_AFFINE_SILU, stats=st, out_dtype=torch.float32)
        assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.bfloat16())
        dy16 = torch.randn(y16.shape, generator=gen).cuda().bfloat16()
        gw = [torch.empty_like(wd.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2) for _ in range(2)]
        ops.conv_wgrad(dy16, x16, kind, gw[0], None, xf=ops.XF_AFFINE_SILU, stats=st)
        ops.conv_wgrad(dy16.float(), x16.float(), kind, gw[1], None, xf=ops.XF_AFFINE_SILU, stats=st)
        assert torch.equal(gw[0], gw[1])


def test_narrow_side_kernels_with_bf16_wide_side(act16):
    """conv_in (3 -> 128, tracked), conv_out (128 -> 3 behind GroupNorm+SiLU), the latent convs: the wide side bf16, the narrow
    side, the weights and the arithmetic fp32 (csrc/skinny.hip)"""
    from vaehip import ops
    gen = torch.Generator().manual_seed(23)
    B, H, W = 2, 16, 32
    # conv_in: fp32 image (padded to 4 channels) -> 128 channels stored as bf16, tracker partials from the stored values
    x4 = _nhwc(torch.cat([torch.randn(B, 3, H, W, generator=gen), torch.zeros(B, 1, H, W)], 1))
    w_in = act16(_to_dev_ohwi(torch.randn(128, 3, 3, 3, generator=gen) / 5))
    b_in = torch.randn(128, generator=gen).cuda()
    tb16, tb32 = ops.conv_track_buffer(B * H * W, 128, "cuda"), ops.conv_track_buffer(B * H * W, 128, "cuda")
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        with ops.option("no_thin_mfma"):  # the VALU kernel on both storages of the output: bitwise
            y16 = ops.conv_fwd(x4, w_in, b_in, "c3", track=tb16)
            y32 = ops.conv_fwd(x4, w_in, b_in, "c3", track=tb32, out_dtype=torch.float32)
    finally:
        ops.PROFILER = None
    assert _names(prof) == ["conv_smallk_kernel"] * 2 and y16.dtype == torch.bfloat16 and torch.equal(y16, y32.bfloat16())
    t16 = ops.track_final(tb16, B * H * W)
    assert _rel(t16, y16.float().abs().mean(dim=(0, 1, 2))) < 1e-5 and _rel(t16, ops.track_final(tb32, B * H * W)) < 5e-3
    # round 4: a bf16 output runs on the matrix pipe (csrc/conv_thin_bf16.hip): image and weights rounded to bf16, fp32
    # accumulation, bias in fp32, one rounding of the result -- against that arithmetic in float64; tracker from the stored values
    tbm = ops.conv_track_buffer(B * H * W, 128, "cuda")
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        ym = ops.conv_fwd(x4, w_in, b_in, "c3", track=tbm)
    finally:
        ops.PROFILER = None
    assert _names(prof) == ["conv_thin_bf16_kernel"] and ym.dtype == torch.bfloat16
    w_r = w_in.bfloat16().double().cpu()
    ref = F.conv2d(x4[..., :3].bfloat16().double().cpu().permute(0, 3, 1, 2), w_r, b_in.double().cpu(), 1, 1).permute(0, 2, 3, 1)
    assert _rel(ym, ref) < 6e-3 and float(((ym.double().cpu() - ref).abs() > 2.0 ** -7 * ref.abs() + 1e-6).float().mean()) == 0.0  # <= 1 bf16 ulp
    assert _rel(ops.track_final(tbm, B * H * W), ym.float().abs().mean(dim=(0, 1, 2))) < 1e-5
    assert torch.equal(ops.conv_fwd(x4, w_in, b_in, "c3"), ym)  # deterministic, with and without the tracker
    # its weight gradient: dY wide (bf16), X narrow (fp32)
    dy16 = torch.randn(B, H, W, 128, generator=gen).cuda().bfloat16()
    gw = [torch.empty(128, 3, 3, 3, device="cuda").permute(0, 3, 1, 2) for _ in range(2)]
    gb = [torch.empty(128, device="cuda") for _ in range(2)]
    with ops.option("no_thin_mfma"):  # the VALU kernel (skinny.hip) on both storages: bitwise
        ops.conv_wgrad(dy16, x4, "c3", gw[0], gb[0])
        ops.conv_wgrad(dy16.float(), x4, "c3", gw[1], gb[1])
    assert torch.equal(gw[0], gw[1]) and torch.equal(gb[0], gb[1])
    # round 4: with the wide side stored as bf16 the launch runs on the matrix pipe (csrc/wgrad_thin_bf16.hip): bf16 products of
    # the bf16 gradient with the image ROUNDED to bf16, fp32 accumulation -- against that arithmetic in float64
    gwm, gbm = torch.empty(128, 3, 3, 3, device="cuda").permute(0, 3, 1, 2), torch.empty(128, device="cuda")
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        ops.conv_wgrad(dy16, x4, "c3", gwm, gbm)
    finally:
        ops.PROFILER = None
    assert _names(prof) == ["wgrad_thin_bf16_kernel<true,0>"], _names(prof)
    xr = x4[..., :3].bfloat16().double().cpu().permute(0, 3, 1, 2)
    wz = torch.zeros(128, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wz, None, 1, 1).backward(dy16.double().cpu().permute(0, 3, 1, 2))
    assert _rel(gwm.cpu(), wz.grad) < 2e-6 and _rel(gbm, dy16.double().sum(dim=(0, 1, 2))) < 2e-6
    assert _rel(gwm, gw[1]) < 1e-2  # (the fp32 kernel on the unrounded image: bf16 rounding of the narrow side only)
    gwm2, gbm2 = torch.empty_like(gwm), torch.empty_like(gbm)
    ops.conv_wgrad(dy16, x4, "c3", gwm2, gbm2)
    assert torch.equal(gwm2, gwm) and torch.equal(gbm2, gbm)  # deterministic
    # conv_out: GroupNorm+SiLU fused on a bf16 input, 3 fp32 outputs; dgrad back to a bf16 128-channel gradient; wgrad
    h16 = _nhwc(torch.randn(B, 128, H, W, generator=gen) + 0.2).bfloat16()
    gamma, beta = (1 + 0.3 * torch.randn(128, generator=gen)).cuda(), (0.2 * torch.randn(128, generator=gen)).cuda()
    st = ops.gn_stats(h16, gamma, beta)
    w_out = act16(_to_dev_ohwi(torch.randn(3, 128, 3, 3, generator=gen) / 30))
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        with ops.option("no_thin_mfma"):
            r16 = ops.conv_fwd(h16, w_out, None, "c3", xf=ops.XF_AFFINE_SILU, stats=st)
            r32 = ops.conv_fwd(h16.float(), w_out, None, "c3", xf=ops.XF_AFFINE_SILU, stats=st)
            dr = torch.randn(B, H, W, 3, generator=gen).cuda()
            g16 = ops.conv_dgrad(dr, w_out, "c3", (H, W))
            g32 = ops.conv_dgrad(dr, w_out, "c3", (H, W), out_dtype=torch.float32)
    finally:
        ops.PROFILER = None
    assert _names(prof) == ["conv_smalln_kernel<2>"] * 2 + ["conv_smallk_kernel"] * 2, _names(prof)
    assert r16.dtype == torch.float32 and torch.equal(r16, r32)
    assert g16.dtype == torch.bfloat16 and torch.equal(g16, g32.bfloat16())
    # conv_out's forward on a bf16 input: the matrix-pipe kernel (fp32 GroupNorm+SiLU per halo element, rounded once; bf16 weights)
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        rm = ops.conv_fwd(h16, w_out, None, "c3", xf=ops.XF_AFFINE_SILU, stats=st)
        rmb = ops.conv_fwd(h16, w_out, torch.tensor([0.5, -1.0, 2.0], device="cuda"), "c3", xf=ops.XF_AFFINE_SILU, stats=st)
    finally:
        ops.PROFILER = None
    assert _names(prof) == ["conv_thinn_bf16_kernel<2>"] * 2 and rm.dtype == torch.float32
    actr = ops.gn_apply(h16, st, ops.XF_AFFINE_SILU).bfloat16().double().cpu().permute(0, 3, 1, 2)
    refr = F.conv2d(actr, w_out.bfloat16().double().cpu(), None, 1, 1).permute(0, 2, 3, 1)
    assert _rel(rm, refr) < 2e-5 and _rel(rm, r32) < 2e-2
    assert _rel(rmb - rm, torch.tensor([0.5, -1.0, 2.0]).expand_as(refr)) < 1e-5
    # conv_out's dgrad with a bf16 result: the matrix-pipe kernel (3 -> 128 channels is a <= 4-channel contraction)
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        gm = ops.conv_dgrad(dr, w_out, "c3", (H, W))
    finally:
        ops.PROFILER = None
    assert _names(prof) == ["conv_thin_bf16_kernel"] and gm.dtype == torch.bfloat16
    drr = dr.bfloat16().double().cpu().permute(0, 3, 1, 2).requires_grad_(False)
    hz = torch.zeros(B, 128, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(hz, w_out.bfloat16().double().cpu(), None, 1, 1).backward(drr)
    refg = hz.grad.permute(0, 2, 3, 1)
    assert _rel(gm, refg) < 6e-3 and float(((gm.double().cpu() - refg).abs() > 2.0 ** -7 * refg.abs() + 1e-6).float().mean()) == 0.0
    gw = [torch.empty(3, 3, 3, 128, device="cuda").permute(0, 3, 1, 2) for _ in range(2)]
    with ops.option("no_thin_mfma"):
        ops.conv_wgrad(dr, h16, "c3", gw[0], None, xf=ops.XF_AFFINE_SILU, stats=st)
        ops.conv_wgrad(dr, h16.float(), "c3", gw[1], None, xf=ops.XF_AFFINE_SILU, stats=st)
    assert torch.equal(gw[0], gw[1])
    # the matrix-pipe kernel: silu(gn(x)) in fp32 from the bf16 tensor, rounded once to bf16; dY rounded to bf16; fp32 accumulation
    gwm, gbm = torch.empty(3, 3, 3, 128, device="cuda").permute(0, 3, 1, 2), torch.empty(3, device="cuda")
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        ops.conv_wgrad(dr, h16, "c3", gwm, gbm, xf=ops.XF_AFFINE_SILU, stats=st)
    finally:
        ops.PROFILER = None
    assert _names(prof) == ["wgrad_thin_bf16_kernel<false,2>"], _names(prof)
    act = ops.gn_apply(h16, st, ops.XF_AFFINE_SILU).bfloat16().double().cpu().permute(0, 3, 1, 2)  # (the same fp32 transform, rounded)
    wz = torch.zeros(3, 128, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(act, wz, None, 1, 1).backward(dr.bfloat16().double().cpu().permute(0, 3, 1, 2))
    assert _rel(gwm.cpu(), wz.grad) < 2e-5 and _rel(gbm, dr.double().sum(dim=(0, 1, 2))) < 2e-6
    assert _rel(gwm, gw[1]) < 1e-2
    # decoder.conv_in (4 -> 512): fp32 latents in, bf16 out; its dgrad: a bf16 512-channel gradient -> 4 fp32 channels
    z = _nhwc(torch.randn(B, 4, 8, 8, generator=gen))
    w_z = act16(_to_dev_ohwi(torch.randn(512, 4, 3, 3, generator=gen) / 6))
    with ops.option("no_thin_mfma"):
        h = ops.conv_fwd(z, w_z, None, "c3")
        assert h.dtype == torch.bfloat16 and torch.equal(h, ops.conv_fwd(z, w_z, None, "c3", out_dtype=torch.float32).bfloat16())
    hm = ops.conv_fwd(z, w_z, None, "c3")  # (4 -> 512 at 8 x 8: one 128-pixel tile, four channel blocks on the matrix-pipe kernel)
    refz = F.conv2d(z.bfloat16().double().cpu().permute(0, 3, 1, 2), w_z.bfloat16().double().cpu(), None, 1, 1).permute(0, 2, 3, 1)
    assert hm.dtype == torch.bfloat16 and _rel(hm, refz) < 6e-3
    # its weight gradient on a 8 x 16 map (128 pixels per image: whole tiles; 512 x 4 channels: four 128-channel blocks of the
    # matrix-pipe kernel); an 8 x 8 map (64 pixels) stays on the VALU kernel
    z2 = _nhwc(torch.randn(3, 4, 8, 16, generator=gen))
    dh2 = torch.randn(3, 8, 16, 512, generator=gen).cuda().bfloat16()
    gwz, gbz = torch.empty(512, 3, 3, 4, device="cuda").permute(0, 3, 1, 2), torch.empty(512, device="cuda")
    prof = ops.PROFILER = ops.LaunchProfiler()
    try:
        ops.conv_wgrad(dh2, z2, "c3", gwz, gbz)
    finally:
        ops.PROFILER = None
    assert _names(prof) == ["wgrad_thin_bf16_kernel<true,0>"], _names(prof)
    wz = torch.zeros(512, 4, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(z2.bfloat16().double().cpu().permute(0, 3, 1, 2), wz, None, 1, 1).backward(dh2.double().cpu().permute(0, 3, 1, 2))
    assert _rel(gwz.cpu(), wz.grad) < 2e-6 and _rel(gbz, dh2.double().sum(dim=(0, 1, 2))) < 2e-6
    dh16 = torch.randn(B, 8, 8, 512, generator=gen).cuda().bfloat16()
    dz = ops.conv_dgrad(dh16, w_z, "c3", (8, 8))
    assert dz.dtype == torch.float32 and torch.equal(dz, ops.conv_dgrad(dh16.float(), w_z, "c3", (8, 8)))


@pytest.mark.parametrize("B,H,W,Ci,Co", [(13, 32, 64, 128, 256), (2, 4, 32, 128, 128)])
def test_upsampler_phase_convolutions_with_bf16_storage(act16, B, H, W, Ci, Co):
    """conv3x3(nearest_upsample_2x(x)) on a bf16 x: bf16 output through the strided view, bf16 gradients back"""
    from vaehip import ops
    gen = torch.Generator().manual_seed(29 + B)
    x16 = _nhwc(torch.randn(B, Ci, H, W, generator=gen)).bfloat16()
    w = torch.randn(Co, Ci, 3, 3, generator=gen) / math.sqrt(9 * Ci)
    wd = act16(_to_dev_ohwi(w))
    bias = torch.randn(Co, generator=gen).cuda()
    y16 = ops.conv_fwd(x16, wd, bias, "c3up")
    y32 = ops.conv_fwd(x16, wd, bias, "c3up", out_dtype=torch.float32)
    assert y16.dtype == torch.bfloat16 and y16.shape == (B, 2 * H, 2 * W, Co) and torch.equal(y16, y32.bfloat16())
    ref = F.conv2d(F.interpolate(x16.float().cpu().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest"), w.bfloat16().float(), bias.cpu(), 1, 1)
    assert _rel(y32.permute(0, 3, 1, 2), ref) < 2e-2  # the phase kernels round SUMS of taps to bf16 (tests/test_kernels_gpu.py holds them to 2e-5 against that)
    dy16 = torch.randn(y16.shape, generator=gen).cuda().bfloat16()
    dx16 = ops.conv_dgrad(dy16, wd, "c3up", (H, W))
    dx32 = ops.conv_dgrad(dy16.float(), wd, "c3up", (H, W), out_dtype=torch.float32)
    assert dx16.dtype == torch.bfloat16 and _rel(dx16.float(), dx32) < 6e-3
    gw = [torch.empty_like(wd.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2) for _ in range(2)]
    gb = [torch.empty(Co, device="cuda") for _ in range(2)]
    ops.conv_wgrad(dy16, x16, "c3up", gw[0], gb[0])
    ops.conv_wgrad(dy16.float(), x16.float(), "c3up", gw[1], gb[1])
    assert _rel(gw[0], gw[1]) < 2e-5 and _rel(gb[0], gb[1]) < 1e-6


def test_storage_flags_are_refused_where_no_kernel_honours_them(cuda):
    """fp32 arithmetic has fp32 storage: the library says so (vae_conv_io16_ok) and vae_igemm_rows refuses the launch"""
    import ctypes as C
    from vaehip import ops
    from vaehip.lib import VaeHipError, lib
    x = torch.randn(1, 8, 32, 128, device="cuda")
    wd = _to_dev_ohwi(torch.randn(128, 128, 3, 3) / 30)
    y = ops.conv_fwd(x, wd, None, "c3")
    assert y.dtype == torch.float32 and not ops.act16()
    a = ops.IgemmArgs()
    a.A, a.W, a.C = ops._p(x), ops._p(ops.ohwi(wd)), ops._p(y)
    a.g = ops._fwd_geom("c3", 1, 8, 32, 128)
    a.M, a.N, a.K, a.ldc, a.sn, a.sk, a.st, a.batch, a.alpha = 256, 128, 128, 128, 9 * 128, 1, 128, 1, 1.0
    assert lib.query("vae_conv_io16_ok", C.byref(a)) == 1
    a.out_bf16 = 1
    assert lib.query("vae_conv_io16_ok", C.byref(a)) == 0
    with pytest.raises(VaeHipError):
        lib.call("vae_igemm_rows", C.byref(a), ops._stream())


@pytest.mark.parametrize("kind,B,H,W,Ci,Co", [("c3", 2, 16, 32, 64, 128),      # one ci block, one co block
                                               ("c3", 3, 8, 64, 128, 256),      # 2 x 2 blocks, units beyond the last one requested
                                               ("c3", 1, 64, 64, 256, 136),     # a co tail: channels beyond M are never fetched
                                               ("c3", 5, 6, 96, 192, 128),      # three ci blocks, odd unit counts per split
                                               ("c3up", 2, 8, 32, 128, 128),    # upsampler: the four phase launches (sub-sampled dY view, tap masks)
                                               ("c3up_virtual", 2, 8, 32, 128, 128),  # upsampler as ONE launch over the virtual nearest-2x upsample
                                               ("c3s2", 3, 16, 64, 128, 128),   # stride-2 downsampler: 1 x 32-pixel units, 3 x 65 halo
                                               ("c3s2", 2, 64, 128, 256, 264),  # several strips per row, a co tail
                                               ("c3s2", 5, 6, 64, 64, 128)])    # odd row counts per split
def test_weight_gradient_staged_by_lds_dma_equals_register_staging(act16, kind, B, H, W, Ci, Co, monkeypatch):
    """wgrad3_dma_bf16_kernel (both operands as bf16 images, staged by LDS-DMA into swizzled images) against
    wgrad3_tile_bf16_kernel (library option no_wgrad_dma: the same images through registers into padded images): same MFMA order,
    same bias-sum order -> bitwise the same dW and db; and against torch on the rounded operands."""
    from vaehip import ops
    if kind == "c3up_virtual":
        monkeypatch.setattr(ops, "PHASE_UPCONV", False)
        kind = "c3up"
    gen = torch.Generator().manual_seed(101 + B + Ci + Co)
    x16 = _nhwc(torch.randn(B, Ci, H, W, generator=gen) * 1.1 - 0.2).bfloat16()
    Ho, Wo = ops.out_hw(kind, H, W)
    dy16 = torch.randn(B, Ho, Wo, Co, generator=gen).cuda().bfloat16()
    wshape = torch.empty(Co, 3, 3, Ci, device="cuda")
    out = {}
    for dma in (True, False):
        gw = torch.full_like(wshape, float("nan")).permute(0, 3, 1, 2)
        gb = torch.full((Co,), float("nan"), device="cuda")
        prof = ops.PROFILER = ops.LaunchProfiler()
        try:
            with ops.option("no_wgrad_dma", 0 if dma else 1):
                ops.conv_wgrad(dy16, x16, kind, gw, gb)
        finally:
            ops.PROFILER = None
        other = "wgrad_bf16_kernel" if kind == "c3s2" else "wgrad3_tile_bf16_kernel"  # (stride 2 without the DMA kernel: the flat kernel)
        names = [n for n in _names(prof) if n.startswith("wgrad")]
        assert names and all(n.startswith("wgrad3_dma_bf16_kernel" if dma else other) for n in names), _names(prof)
        out[dma] = (gw.clone(), gb.clone())
    if kind == "c3s2":  # another kernel, another order of the fp32 sums
        assert _rel(out[True][0], out[False][0]) < 2e-5 and _rel(out[True][1], out[False][1]) < 2e-6
    else:
        assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
    xr = x16.float().cpu().permute(0, 3, 1, 2)
    if kind == "c3up":
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    wg = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    (F.conv2d(F.pad(xr, (0, 1, 0, 1)), wg, None, 2, 0) if kind == "c3s2" else F.conv2d(xr, wg, None, 1, 1)).backward(dy16.float().cpu().permute(0, 3, 1, 2))
    assert _rel(out[True][0].cpu(), wg.grad) < 3e-5
    assert _rel(out[True][1].cpu(), dy16.float().sum((0, 1, 2)).cpu()) < 1e-5
