#!/usr/bin/env python
"""CPU study (no GPU): what would Winograd F(4x4,3x3) cost in accuracy against today's F(2x2,3x3)?

VERDICT r03, next-round item 2(a): emulate F(4x4,3x3) in fp32 on the CPU, operands from the oracle's R = 64 step, against an
fp64 direct convolution, next to F(2x2,3x3).  Two levels:

  layer level   the 8 plain 3x3 stride-1 shapes of SURVEY 8a: forward, dgrad and wgrad of ONE layer on the operands the
                oracle's R = 64 step really feeds it (silu(gn(x)) and dL/dy), error vs the fp64 direct result;
  step level    the oracle's whole train step with EVERY layer the Winograd kernels serve today (H % 8 == 0, W % 16 == 0,
                Cin, Cout >= 64) replaced by the emulation (forward, dgrad, wgrad), vs the fp64 oracle: losses, the tracked
                per-channel statistics and every gradient tensor relative to its own max (the bars of tests/test_engine_gpu.py).

Every transform, product and accumulation of the emulation runs in fp32 (torch CPU; the products are accumulated over input
channels by torch's fp32 matmul, i.e. in blocked order rather than the MFMA's sequential chain: same error growth law).
Variants: "f2" = F(2x2,3x3) (points 0, +-1, inf), "f4" = F(4x4,3x3) with Lavin's points (0, +-1, +-2, inf), "f4h" =
F(4x4,3x3) with the points (0, +-1, +-1/2, inf), which keep the transform entries closer to 1.

Writes profiles/r04_wino_f4_error_study.json.
"""
import json
import os
import sys
from fractions import Fraction

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vae_oracle as vo  # noqa: E402


def cook_toom(m: int, r: int, pts):
    """matrices of F(m, r) from interpolation points (finite ones; infinity is appended): AT [m, a], G [a, r], BT [a, a],
    a = m + r - 1, exact rationals -> float64 arrays.  y = AT [(G g) * (BT d)]."""
    a = m + r - 1
    assert len(pts) == a - 1
    pts = [Fraction(p) for p in pts]
    # AT[i][k] = p_k^i, last column = [0..0,1]
    AT = [[(p ** i) for p in pts] + [Fraction(1 if i == m - 1 else 0)] for i in range(m)]
    # G[k][j] = p_k^j / prod_{l != k}(p_k - p_l); last row = [0..0,1]
    G = []
    for k, p in enumerate(pts):
        den = Fraction(1)
        for l, q in enumerate(pts):
            if l != k:
                den *= (p - q)
        G.append([(p ** j) / den for j in range(r)])
    G.append([Fraction(1 if j == r - 1 else 0) for j in range(r)])
    # BT rows: coefficients of prod_{l != k}(x - p_l) for finite k; last row: prod_l (x - p_l)
    def polymul(a_, b_):
        out = [Fraction(0)] * (len(a_) + len(b_) - 1)
        for i, x in enumerate(a_):
            for j, y in enumerate(b_):
                out[i + j] += x * y
        return out
    BT = []
    for k in range(len(pts)):
        poly = [Fraction(1)]
        for l, q in enumerate(pts):
            if l != k:
                poly = polymul(poly, [-q, Fraction(1)])
        BT.append(poly + [Fraction(0)] * (a - len(poly)))
    poly = [Fraction(1)]
    for q in pts:
        poly = polymul(poly, [-q, Fraction(1)])
    BT.append(poly)
    f = lambda M: np.array([[float(x) for x in row] for row in M], dtype=np.float64)
    return f(AT), f(G), f(BT)


def _check(m, r, pts):
    AT, G, BT = cook_toom(m, r, pts)
    rng = np.random.default_rng(0)
    g, d = rng.standard_normal(r), rng.standard_normal(m + r - 1)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(g[j] * d[i + j] for j in range(r)) for i in range(m)])
    assert np.allclose(y, ref, atol=1e-9), (y, ref)
    return AT, G, BT


VARIANTS = {
    "f2": (2, _check(2, 3, [0, 1, -1])),
    "f4": (4, _check(4, 3, [0, 1, -1, 2, -2])),
    "f4h": (4, _check(4, 3, [0, 1, -1, Fraction(1, 2), Fraction(-1, 2)])),
}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.float32)


class WinoConv(torch.autograd.Function):
    """3x3 stride-1 pad-1 convolution, forward / dgrad / wgrad all as fp32 Winograd of the given variant"""

    @staticmethod
    def _fwd(x, w, m, mats):
        AT, G, BT = (_t(a) for a in mats)
        Bn, Cc, H, W = x.shape
        a = m + 2
        xp = F.pad(x, (1, 1, 1, 1))
        d = xp.unfold(2, a, m).unfold(3, a, m)                      # [B,C,nh,nw,a,a]
        V = torch.einsum("ij,bcnmjk,lk->ilbnmc", BT, d, BT).contiguous()   # [a,a,B,nh,nw,C]
        U = torch.einsum("ij,ocjk,lk->ilco", G, w, G).contiguous()         # [a,a,C,O]
        nh, nw = V.shape[3], V.shape[4]
        M = torch.matmul(V.reshape(a, a, Bn * nh * nw, Cc), U)             # [a,a,P,O] fp32 accumulation over C
        M = M.reshape(a, a, Bn, nh, nw, -1)
        Y = torch.einsum("pi,ilbnmo,ql->bonpmq", AT, M, AT)                # [B,O,nh,m,nw,m]
        return Y.reshape(Bn, -1, nh * m, nw * m)

    @staticmethod
    def forward(ctx, x, w, b, m, mats):
        ctx.save_for_backward(x, w)
        ctx.m, ctx.mats = m, mats
        y = WinoConv._fwd(x, w, m, mats)
        return y + b.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        m, mats = ctx.m, ctx.mats
        AT, G, BT = (_t(a) for a in mats)
        # dgrad: correlation of dy with the flipped, transposed kernel
        wt = w.flip(2, 3).transpose(0, 1).contiguous()
        dx = WinoConv._fwd(dy.contiguous(), wt, m, mats)
        # wgrad: dW = GT [ (A dY AT) (.) (BT X B) ] G, summed over tiles in the transform domain
        Bn, Cc, H, W = x.shape
        a = m + 2
        xp = F.pad(x, (1, 1, 1, 1))
        d = xp.unfold(2, a, m).unfold(3, a, m)
        V = torch.einsum("ij,bcnmjk,lk->ilbnmc", BT, d, BT)               # [a,a,B,nh,nw,C]
        nh, nw = V.shape[3], V.shape[4]
        dyt = dy.reshape(Bn, -1, nh, m, nw, m)
        Yt = torch.einsum("ip,bonpmq,lq->ilbnmo", AT.t().contiguous(), dyt, AT.t().contiguous())  # [a,a,B,nh,nw,O]
        P = Bn * nh * nw
        S = torch.matmul(Yt.reshape(a, a, P, -1).transpose(2, 3), V.reshape(a, a, P, Cc))  # [a,a,O,C] fp32 sum over pixels
        dw = torch.einsum("ij,iloc,lk->ocjk", G, S, G)
        db = dy.sum(dim=(0, 2, 3))
        return dx, dw, db, None, None


class WinoConv2d(nn.Conv2d):
    variant = "f2"

    def forward(self, x):
        m, mats = VARIANTS[self.variant]
        return WinoConv.apply(x, self.weight, self.bias, m, mats)


def eligible(mod: nn.Conv2d, H: int, W: int) -> bool:
    """what csrc/conv3_wino.hip serves today"""
    return (mod.kernel_size == (3, 3) and mod.stride == (1, 1) and mod.padding == (1, 1) and mod.in_channels >= 64
            and mod.out_channels >= 64 and H % 8 == 0 and W % 16 == 0)


def patch(wrapper, variant, shapes):
    """swap eligible convs for the emulation; `shapes`: module name -> (H, W) of its input at this resolution"""
    n = 0
    for name, mod in list(wrapper.named_modules()):
        if isinstance(mod, nn.Conv2d) and name in shapes and "upsamplers" not in name and eligible(mod, *shapes[name]):
            mod.__class__ = WinoConv2d
            mod.variant = variant
            n += 1
    return n


def conv_input_shapes(wrapper, x, eps):
    shapes = {}
    hs = []
    for name, mod in wrapper.named_modules():
        if isinstance(mod, nn.Conv2d):
            hs.append(mod.register_forward_hook(lambda m, i, o, name=name: shapes.__setitem__(name, tuple(i[0].shape[2:]))))
    with torch.no_grad():
        wrapper(x, sample_posterior=True, eps=eps)
    for h in hs:
        h.remove()
    return shapes


TRACKED = ["vae.encoder.conv_in", "vae.encoder.down_blocks.0.resnets.0.norm1", "vae.decoder.up_blocks.1.resnets.0.norm1"]


def run_step(variant, R, B, dtype):
    torch.manual_seed(0)
    w = vo.OracleWrapper(seed=42)
    x, eps = vo.synthetic_pixels(B, R, 42), vo.synthetic_eps(B, R, 42)
    shapes = conv_input_shapes(w, x, eps)
    nsw = 0
    if variant != "direct":
        nsw = patch(w, variant, shapes)
    if dtype == torch.float64:
        w = w.double()
        x, eps = x.double(), eps.double()
    stats = {}
    hs = [w.get_submodule(n).register_forward_hook(lambda m, i, o, n=n: stats.__setitem__(n, o.detach().abs().mean(dim=(0, 2, 3)).double()))
          for n in TRACKED]
    out = w(x, sample_posterior=True, eps=eps)
    rec, kl, total = vo.losses(out, x, 1e-6) if dtype == torch.float32 else (
        F.mse_loss(out["reconstruction"], x), out["latent_dist"].kl().mean(), None)
    if total is None:
        total = rec + 1e-6 * kl
    total.backward()
    for h in hs:
        h.remove()
    grads = {n: p.grad.detach().double() for n, p in w.named_parameters()}
    return dict(rec=float(rec), kl=float(kl), total=float(total), grads=grads, stats=stats, swapped=nsw,
                recon=out["reconstruction"].detach().double())


def compare(res, ref):
    out = {"swapped_layers": res["swapped"]}
    for k in ("rec", "kl", "total"):
        out[k + "_rel"] = abs(res[k] - ref[k]) / abs(ref[k])
    out["recon_rel_to_max"] = float((res["recon"] - ref["recon"]).abs().max() / ref["recon"].abs().max())
    gmax = max(float(g.abs().max()) for g in ref["grads"].values())
    worst, wname = 0.0, None
    for n, g in ref["grads"].items():
        own = float(g.abs().max())
        if own < 1e-6 * gmax:   # the mathematically-zero tensor (attention to_k.bias): judged against the global scale
            continue
        e = float((res["grads"][n] - g).abs().max()) / own
        if e > worst:
            worst, wname = e, n
    out["grad_worst_rel_to_own_max"], out["grad_worst_tensor"] = worst, wname
    gn = lambda d: float(torch.sqrt(sum((g ** 2).sum() for g in d["grads"].values())))
    out["grad_norm_rel"] = abs(gn(res) - gn(ref)) / gn(ref)
    out["tracker_worst_rel"] = max(float(((res["stats"][n] - ref["stats"][n]).abs() / ref["stats"][n].abs().clamp_min(1e-30)).max()) for n in TRACKED)
    return out


def layer_level(R=64, B=2, spatial=32):
    """single-layer errors on real operands: input = silu(gn(x)) as the oracle's step feeds the layer, dy = the real dL/dy"""
    w = vo.OracleWrapper(seed=42)
    x, eps = vo.synthetic_pixels(B, R, 42), vo.synthetic_eps(B, R, 42)
    want = {  # one example of each of the 8 plain 3x3 stride-1 shapes
        "128->128": "vae.encoder.down_blocks.0.resnets.0.conv1", "512->512": "vae.decoder.up_blocks.0.resnets.0.conv1",
        "256->256": "vae.encoder.down_blocks.1.resnets.0.conv2", "512->256": "vae.decoder.up_blocks.2.resnets.0.conv1",
        "256->128": "vae.decoder.up_blocks.3.resnets.0.conv1", "128->256": "vae.encoder.down_blocks.1.resnets.0.conv1",
        "256->512": "vae.encoder.down_blocks.2.resnets.0.conv1", "512->512 (mid)": "vae.encoder.mid_block.resnets.0.conv1",
    }
    cap = {}
    hs = []
    for tag, name in want.items():
        mod = w.get_submodule(name)
        hs.append(mod.register_forward_hook(lambda m, i, o, tag=tag: cap.__setitem__(tag, [i[0].detach(), None])))
        hs.append(mod.register_full_backward_hook(lambda m, gi, go, tag=tag: cap[tag].__setitem__(1, go[0].detach())))
    out = w(x, sample_posterior=True, eps=eps)
    _, _, total = vo.losses(out, x, 1e-6)
    total.backward()
    for h in hs:
        h.remove()
    rows = {}
    for tag, name in want.items():
        mod = w.get_submodule(name)
        xin, dy = cap[tag]
        # tile the real operands up to `spatial` (small maps at R = 64) so every variant has whole tiles
        rep = max(1, spatial // xin.shape[2])
        xin, dy = xin.repeat(1, 1, rep, rep), dy.repeat(1, 1, rep, rep)
        wt, b = mod.weight.detach(), mod.bias.detach()
        x64 = xin.double().requires_grad_(True)
        w64 = wt.double().requires_grad_(True)
        y64 = F.conv2d(x64, w64, b.double(), padding=1)
        y64.backward(dy.double())
        ref = dict(y=y64.detach(), dx=x64.grad, dw=w64.grad)
        row = {"input_hw": list(xin.shape[2:])}
        for var in ("direct", "f2", "f4", "f4h"):
            x32 = xin.clone().requires_grad_(True)
            w32 = wt.clone().requires_grad_(True)
            if var == "direct":
                y = F.conv2d(x32, w32, b, padding=1)
            else:
                m, mats = VARIANTS[var]
                y = WinoConv.apply(x32, w32, b, m, mats)
            y.backward(dy)
            got = dict(y=y.detach().double(), dx=x32.grad.double(), dw=w32.grad.double())
            row[var] = {k: {"max_rel_to_max": float((got[k] - ref[k]).abs().max() / ref[k].abs().max()),
                            "rms_rel": float((got[k] - ref[k]).pow(2).mean().sqrt() / ref[k].pow(2).mean().sqrt())} for k in ("y", "dx", "dw")}
        rows[tag] = row
        print(tag, json.dumps({v: {k: f"{row[v][k]['max_rel_to_max']:.2e}" for k in ("y", "dx", "dw")} for v in ("direct", "f2", "f4", "f4h")}), flush=True)
    return rows


def main():
    torch.set_num_threads(8)
    res = {"what": __doc__.split("\n\n")[0], "layer_level": layer_level()}
    R, B = 64, 2
    ref = run_step("direct", R, B, torch.float64)
    step = {}
    for var in ("direct", "f2", "f4", "f4h"):
        r = run_step(var, R, B, torch.float32)
        step[var] = compare(r, ref)
        print(var, json.dumps(step[var]), flush=True)
    res["step_level"] = {"R": R, "B": B, "reference": "the same oracle step in float64 (direct convolutions)", "variants": step}
    with open(os.path.join(ROOT, "profiles", "r04_wino_f4_error_study.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
