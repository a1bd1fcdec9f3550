"""Execution engine of the SDXL-VAE train step on the HIP kernels.

The engine walks the module tree itself (encode -> sample -> decode, the composition of
reference src/models/sdxl_vae_wrapper.py:42-77) and records a tape of backward closures, so
GroupNorm+SiLU can be fused into the consuming convolution and the normalised activations
never touch HBM.  The torch.nn hook protocol the reference's plugins rely on
(monitor.py:126-133, sdxl_vae_wrapper.py:104-107) is honoured: a module that carries foreign
forward hooks gets its input/output materialised and the hooks are called synchronously;
the shipped metric `mean_abs_activation_per_channel` (monitor.py:66) is served by fused
device-side reductions instead (add_tracker), without any tensor or host sync.
"""
from __future__ import annotations

import contextlib
import weakref
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .ops import XF_AFFINE, XF_AFFINE_SILU, XF_NONE, Stats

_TRACK_C = {4, 8, 16, 32, 64, 128, 256, 512, 1024}


class _TrackHandle:
    def __init__(self, store: dict, key, point: str, sink):
        self._s, self._k, self._p, self._f = store, key, point, sink

    def remove(self):
        lst = self._s.get(self._k, {}).get(self._p, [])
        if self._f in lst:
            lst.remove(self._f)


class Engine:
    def __init__(self, vae: nn.Module):
        self._vae = weakref.ref(vae)
        self._trackers: Dict[int, Dict[str, List[Callable]]] = {}
        self._gtarget: Optional[torch.Tensor] = None
        self.reducer = None  # optional DP bucket reducer with .ready(low_offset)
        self._low: Dict[int, int] = {}
        self._ident: Dict[tuple, Stats] = {}
        self.precision = ops.PREC_F32  # arithmetic of the conv contractions (set_precision)
        self._flat16: Optional[torch.Tensor] = None  # bf16 image of the parameter arena (bf16 mode)
        # bf16 mode: the bf16 image of the GroupNorm+SiLU'd input the LAST _conv call made (None if it made none); a
        # recording caller takes it right away (_take16) into its backward closure, so the image lives and dies with
        # the tape that recorded it (two recorded forwards before one backward keep two images; an abandoned tape frees its own)
        self._last16: Optional[torch.Tensor] = None
        # activation checkpointing of the decoder (BASELINE config 5): every resnet / attention / sampler of the decoder
        # keeps only its input; its forward is run again (recording) right before its own backward
        self.checkpoint_decoder = False
        self._ckpt = False    # inside a region whose segments are checkpointed
        self._replay = False  # True while a checkpointed segment is re-run: trackers / hooks already fired

    # ------------------------------------------------------------------ plumbing
    @property
    def vae(self):
        return self._vae()

    @property
    def arena(self):
        return self.vae.arena

    def set_precision(self, mode) -> None:
        """'no'/'fp32' (exact fp32 MFMA) or 'bf16' (operands rounded to bf16 in LDS, fp32 accumulate; tensors,
        master weights and statistics stay fp32).  Mirrors training.mixed_precision of the reference's YAML."""
        if mode in (None, "no", "fp32", torch.float32, ops.PREC_F32):
            self.precision = ops.PREC_F32
        elif mode in ("bf16", torch.bfloat16) or (mode == ops.PREC_BF16 and not isinstance(mode, bool)):
            self.precision = ops.PREC_BF16
        else:
            raise NotImplementedError(f"mixed_precision={mode!r}: only 'no' and 'bf16' exist on this path "
                                      "(fp16 needs loss scaling, which the reference does not configure either)")

    @contextlib.contextmanager
    def _mode(self, repack: bool = True):
        """arithmetic mode of the contraction kernels for one pass.  In bf16 mode the kernels read a bf16 image of the
        parameter arena; a forward pass refreshes it first (parameters may have been stepped, nudged or loaded since),
        a backward pass reuses the image its forward made."""
        prev16 = ops.WEIGHTS16
        with ops.precision(self.precision):
            if self.precision == ops.PREC_BF16:
                flat = self.arena.flat
                img = self._flat16
                if img is None or img.numel() != flat.numel() or img.device != flat.device:
                    img = self._flat16 = torch.empty(flat.numel(), device=flat.device, dtype=torch.bfloat16)
                    repack = True
                if repack:
                    ops.pack_bf16(flat, img)
                ops.WEIGHTS16 = (flat.data_ptr(), flat.numel() * 4, img.data_ptr())
            try:
                yield
            finally:
                ops.WEIGHTS16 = prev16

    def _require_gpu(self):
        if self.arena.flat.device.type != "cuda":
            raise RuntimeError("the VAE parameters are on %s; the HIP engine needs a GPU (there is no CPU fallback); "
                               "call .to('cuda') first" % self.arena.flat.device)

    def _g(self, p: Optional[nn.Parameter]):
        return None if p is None else self.arena.grad_view(p, self._gtarget)

    def _done(self, module: nn.Module):
        if self.reducer is not None:
            k = id(module)
            if k not in self._low:
                self._low[k] = self.arena.block_low_offset(module)
            self.reducer.ready(self._low[k])

    # ------------------------------------------------------------------ trackers and hooks
    def add_tracker(self, module: nn.Module, point: str, sink: Callable[[torch.Tensor], None]):
        """fused `mean_abs_activation_per_channel` for `module`'s input or output: sink(tensor[C]) per forward."""
        assert point in ("input", "output")
        self._trackers.setdefault(id(module), {}).setdefault(point, []).append(sink)
        return _TrackHandle(self._trackers, id(module), point, sink)

    def _tracked(self, m, point):
        if self._replay:
            return None
        d = self._trackers.get(id(m))
        return d.get(point) if d else None

    def _mean_abs(self, t: torch.Tensor) -> torch.Tensor:
        B, Cc = t.shape[0], t.shape[-1]
        if Cc in _TRACK_C:
            key = (B, Cc, t.device)
            st = self._ident.get(key)
            if st is None:
                one = torch.ones((B, Cc), device=t.device)
                st = Stats(None, None, one, torch.zeros_like(one))
                self._ident[key] = st
            return ops.gn_track(t.reshape(B, -1, 1, Cc), st)
        return t.float().abs().mean(dim=tuple(range(t.ndim - 1)))

    @staticmethod
    def _present(m, t: torch.Tensor) -> torch.Tensor:
        """NHWC buffer -> the tensor a torch hook on `m` expects (NCHW-logical view; [B,T,C] for Linear).  A bf16-stored
        activation is handed over as an fp32 copy: the reference's hooks call .numpy() on it (monitor.py:67), which bf16
        tensors do not support."""
        t = ops.to_f32(t)
        if isinstance(m, nn.Linear):
            return t.reshape(t.shape[0], -1, t.shape[-1])
        return t.permute(0, 3, 1, 2)

    def _pre(self, m, make_in):
        sinks = self._tracked(m, "input")
        if (m._forward_pre_hooks and not self._replay) or sinks:
            t = make_in()
            if sinks:
                v = self._mean_abs(t)
                for s in sinks:
                    s(v)
            for h in list(m._forward_pre_hooks.values()):
                h(m, (self._present(m, t),))

    def _post(self, m, make_in, out: torch.Tensor, tracked_already=False):
        sinks = None if tracked_already else self._tracked(m, "output")
        if sinks:
            v = self._mean_abs(out)
            for s in sinks:
                s(v)
        if m._forward_hooks and not self._replay:
            tin = self._present(m, make_in())
            for h in list(m._forward_hooks.values()):
                h(m, (tin,), self._present(m, out))

    @staticmethod
    def _no_hooks(m, what):
        if m is not None and (m._forward_hooks or m._forward_pre_hooks):
            raise NotImplementedError(f"forward hooks on {what} are not supported: it is fused into the adjacent "
                                      f"convolution. Hook the GroupNorm before it or the convolution after it.")

    # ------------------------------------------------------------------ leaves (forward)
    def _gn(self, norm, x: torch.Tensor) -> Stats:
        self._pre(norm, lambda: x)
        st = ops.gn_stats(x, norm.weight, norm.bias, norm.num_groups, norm.eps)
        sinks = self._tracked(norm, "output")
        if sinks:
            v = ops.gn_track(x, st)
            for s in sinks:
                s(v)
        if norm._forward_hooks and not self._replay:
            self._post(norm, lambda: x, ops.gn_apply(x, st, XF_AFFINE), tracked_already=True)
        return st

    def _conv(self, m, x: torch.Tensor, xf: int, st: Optional[Stats], res: Optional[torch.Tensor] = None,
              out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """out_dtype: storage of the output when the default (ops.conv_fwd: bf16 for wide layers in bf16 mode) is not wanted"""
        kind = getattr(m, "kind", "c1")
        cache = {}

        def make_in():
            if "t" not in cache:
                t = x if xf == XF_NONE else ops.gn_apply(x, st, xf)
                ci = m.weight.shape[1]
                cache["t"] = t if t.shape[-1] == ci else t[..., :ci]
            return cache["t"]

        self._pre(m, make_in)
        sinks = self._tracked(m, "output")
        fuse_res = res is not None and not m._forward_hooks and not sinks
        tb = None
        if sinks:
            ho, wo = ops.out_hw(kind, x.shape[1], x.shape[2])
            tb = ops.conv_track_buffer(x.shape[0] * ho * wo, m.weight.shape[0], x.device)
        a16 = None
        if xf != XF_NONE and ops.act_image_ok(kind, x.shape, m.weight.shape[0], m.weight.shape[1]):
            # transform and round once (2 B/element written); forward and wgrad then read the image and transform nothing
            a16 = ops.gn_apply_bf16(x, st, xf)
        xin, xfc, stc = x, xf, st
        if a16 is None and xf != XF_NONE and tb is None and ops.act_image32_ok(kind, x.shape, m.weight.shape[0], m.weight.shape[1]):
            # fp32 mode, wide layers: the transformed tensor once, forward and wgrad read it as is (ops.ACT_IMAGE32_MIN_CIN)
            a16 = xin = make_in() if "t" in cache else ops.gn_apply(x, st, xf)
            xfc, stc = XF_NONE, None
        self._last16 = a16
        # every 3x3 output of this model feeds a GroupNorm(32) next (or is summed first: then the statistics are dropped)
        want_stats = ops.GN_GROUPS if (kind in ("c3", "c3up") and (res is None or fuse_res)) else None
        y = ops.conv_fwd(xin, m.weight, m.bias, kind, xf=xfc, stats=stc, res=res if fuse_res else None, track=tb,
                         a16=a16 if a16 is not xin else None, gstat_groups=want_stats, out_dtype=out_dtype)
        if sinks:
            v = ops.track_final(tb, y.shape[0] * y.shape[1] * y.shape[2])
            for s in sinks:
                s(v)
        if m._forward_hooks:
            self._post(m, make_in, y, tracked_already=True)
        if res is not None and not fuse_res:
            y = ops.add(y, res)
        return y

    def _take16(self, recording: bool) -> Optional[torch.Tensor]:
        a16, self._last16 = self._last16, None
        return a16 if recording else None

    def _conv_bwd(self, m, x, dy, xf, st, need_dx=True, x16=None, dx_to_gn=False, dx_dtype=None, norm=None):
        """dy: fp32 (possibly carrying a bf16 image) or bf16; dx_to_gn: the input gradient goes to a GroupNorm backward and
        nowhere else, so bf16 mode may store it as bf16 even with fp32 activation storage; dx_dtype: storage of the input
        gradient when the default (ops.conv_dgrad) is not wanted"""
        kind = getattr(m, "kind", "c1")
        # norm: the GroupNorm(+SiLU) between x and this convolution, whose backward consumes the input gradient next: the dgrad
        # epilogue can leave that backward's first pass (ops.conv_dgrad gnb)
        gnb = None
        if norm is not None and need_dx and st is not None and xf in (XF_AFFINE, XF_AFFINE_SILU):
            gnb = ops.GnCtx(x, st, norm.weight, norm.bias, xf == XF_AFFINE_SILU, norm.num_groups)
        if (need_dx and dy.dtype == torch.float32 and getattr(dy, "_b16", None) is None
                and ops.grad_image_ok(kind, x.shape[:3] + (m.weight.shape[1],), m.weight.shape[0], m.weight.shape[1])):
            # a gradient that arrives without its bf16 image (from an upsampler, an attention block, the loss): rounding it
            # once here costs what the two consumers save in reads, and puts both on the kernels that take an image
            dy._b16 = ops.pack_bf16(dy, torch.empty(dy.shape, device=dy.device, dtype=torch.bfloat16))
        if x16 is not None and x16.dtype == torch.float32:  # the fp32 activation image of _conv: already transformed
            x, xf, st, x16 = x16, XF_NONE, None, None
        ops.conv_wgrad(dy, x, kind, self._g(m.weight), self._g(m.bias), xf=xf, stats=st, x16=x16)
        if need_dx:
            return ops.conv_dgrad(dy, m.weight, kind, (x.shape[1], x.shape[2]), out_bf16=dx_to_gn, out_dtype=dx_dtype, gnb=gnb)
        return None

    def _gn_bwd(self, norm, x, g, st, silu, add, conv_only=None, feeds_conv3=False):
        """conv_only: the convolution that is the ONLY consumer of the result (a resnet's dL/dh -> conv1): stored as bf16
        alone when its kernels take a bf16 gradient image.  feeds_conv3: the result continues the residual stream (fp32) and
        is also the output gradient of a 3x3 convolution upstream: a bf16 image is attached for that consumer."""
        want32, want16 = True, False
        if ops.act16():  # bf16 storage: every activation gradient is a bf16 tensor
            want32, want16 = False, True
        elif conv_only is not None and ops.grad_image_ok(getattr(conv_only, "kind", "c1"), x.shape[:3] + (conv_only.weight.shape[1],),
                                                      conv_only.weight.shape[0], conv_only.weight.shape[1]):
            want32, want16 = False, True
        elif feeds_conv3 and ops.grad_image_ok("c3", x.shape, x.shape[3], x.shape[3]):
            want16 = True
        return ops.gn_bwd(x, g, st, norm.weight, norm.bias, silu, add, self._g(norm.weight), self._g(norm.bias),
                          norm.num_groups, want32=want32, want16=want16)

    # ------------------------------------------------------------------ blocks
    def _plain_conv(self, m, x, tape, need_dx=True, owner=None):
        y = self._conv(m, x, XF_NONE, None)
        if tape is not None:
            def bwd(d):
                dx = self._conv_bwd(m, x, d, XF_NONE, None, need_dx)
                self._done(owner or m)
                return dx
            tape.append(bwd)
        return y

    def _seg(self, run, x, tape):
        """run(x, tape) -> y, checkpointed when the enclosing region asks for it: the forward records nothing and the
        backward closure re-runs `run` on the kept input with a private tape (same kernels, bitwise the same values)."""
        if tape is None or not self._ckpt:
            return run(x, tape)
        y = run(x, None)

        def bwd(d):
            local: list = []
            self._replay = True
            try:
                run(x, local)
            finally:
                self._replay = False
            while local:
                d = local.pop()(d)
            return d
        tape.append(bwd)
        return y

    def _resnet(self, r, x, tape, notify=True, after_conv3=False):
        """after_conv3: what produced x is the output of a 3x3 stride-1 convolution of the same width (a resnet's conv2)"""
        self._no_hooks(r.nonlinearity, "ResnetBlock2D.nonlinearity")
        self._no_hooks(r.dropout, "ResnetBlock2D.dropout")
        self._pre(r, lambda: x)
        rec = tape is not None
        st1 = self._gn(r.norm1, x)
        h = self._conv(r.conv1, x, XF_AFFINE_SILU, st1)
        x16 = self._take16(rec)
        st2 = self._gn(r.norm2, h)
        sc = self._conv(r.conv_shortcut, x, XF_NONE, None) if r.conv_shortcut is not None else x
        out = self._conv(r.conv2, h, XF_AFFINE_SILU, st2, res=sc)
        h16 = self._take16(rec)
        self._post(r, lambda: x, out)
        if tape is not None:
            def bwd(dout):
                g2 = self._conv_bwd(r.conv2, h, dout, XF_AFFINE_SILU, st2, x16=h16, dx_to_gn=True, norm=r.norm2)
                dh = self._gn_bwd(r.norm2, h, g2, st2, True, None, conv_only=r.conv1)  # dL/dh only feeds conv1's wgrad + dgrad
                g1 = self._conv_bwd(r.conv1, x, dh, XF_AFFINE_SILU, st1, x16=x16, dx_to_gn=True, norm=r.norm1)
                dsc = self._conv_bwd(r.conv_shortcut, x, dout, XF_NONE, None) if r.conv_shortcut is not None else dout
                dx = self._gn_bwd(r.norm1, x, g1, st1, True, dsc, feeds_conv3=after_conv3)
                if notify is True:
                    self._done(r)
                elif notify is not False:
                    self._done(notify)  # an enclosing module whose parameters are all final now
                return dx
            tape.append(bwd)
        return out

    def _attention(self, a, x, tape, notify=True, after_conv3=False):
        self._pre(a, lambda: x)
        B, H, W, Cc = x.shape
        T = H * W
        scale = float(Cc) ** -0.5
        st = self._gn(a.group_norm, x)
        # the attention block's internals (q, k, v, scores, context and their gradients) stay fp32 whatever the storage of the
        # residual stream: they are small (512 channels at 1/8 resolution) and the softmax statistics want them
        f32 = torch.float32
        q = self._conv(a.to_q, x, XF_AFFINE, st, out_dtype=f32)
        k = self._conv(a.to_k, x, XF_AFFINE, st, out_dtype=f32)
        v = self._conv(a.to_v, x, XF_AFFINE, st, out_dtype=f32)
        qf, kf, vf = q.view(B, T, Cc), k.view(B, T, Cc), v.view(B, T, Cc)
        self._no_hooks(a.to_out[1], "Attention.to_out.1 (dropout)")
        # long sequences (R >= 512): blockwise kernels, online softmax, no T x T tensor; P is recomputed in the backward
        # from the saved row log-sum-exp.  Short ones keep the materialised scores (4 MB per image at T = 1024).
        blockwise = ops.attn_blockwise_ok(T, Cc)
        if blockwise:
            of, saved = ops.attn_fwd(qf, kf, vf, scale)
            P = qf = kf = vf = None  # the saved operands (fp32 tensors or their bf16 images) are what the backward reads
        else:
            P = ops.softmax_rows_(ops.gemm_nt(qf, kf, scale))
            of = ops.gemm_nn(P, vf)
            ops.ATTN_CALLS["materialised_fwd"] += 1
            saved = None
        if tape is None:
            P = saved = None
        o = of.view(B, H, W, Cc)
        out = self._conv(a.to_out[0], o, XF_NONE, None, res=x)
        self._post(a, lambda: x, out)
        if tape is not None:
            def bwd(dout):
                do = self._conv_bwd(a.to_out[0], o, dout, XF_NONE, None, dx_dtype=f32).view(B, T, Cc)
                if blockwise:
                    dq, dk, dv = ops.attn_bwd(saved, of, do, scale)
                else:
                    dP = ops.gemm_nt(do, vf)
                    dv = ops.gemm_tn(P, do)
                    dS = ops.softmax_bwd_rows_(P, dP)
                    dq = ops.gemm_nn(dS, kf, scale)
                    dk = ops.gemm_tn(dS, qf, scale)
                g = self._conv_bwd(a.to_q, x, dq.view(B, H, W, Cc), XF_AFFINE, st, dx_dtype=f32)
                g = ops.add(g, self._conv_bwd(a.to_k, x, dk.view(B, H, W, Cc), XF_AFFINE, st, dx_dtype=f32))
                g = ops.add(g, self._conv_bwd(a.to_v, x, dv.view(B, H, W, Cc), XF_AFFINE, st, dx_dtype=f32))
                dx = self._gn_bwd(a.group_norm, x, g, st, False, dout, feeds_conv3=after_conv3)
                if notify:
                    self._done(a)
                return dx
            tape.append(bwd)
        return out

    def _mid(self, mb, x, tape, after_conv3=False):
        self._pre(mb, lambda: x)
        # registration order (attentions, resnets) differs from execution order, so the DP watermark
        # only moves once the whole mid block is final: after resnets[0]'s backward (last on the tape)
        h = self._seg(lambda t, tp: self._resnet(mb.resnets[0], t, tp, notify=mb, after_conv3=after_conv3), x, tape)
        h = self._seg(lambda t, tp: self._attention(mb.attentions[0], t, tp, notify=False, after_conv3=True), h, tape)
        h = self._seg(lambda t, tp: self._resnet(mb.resnets[1], t, tp, notify=False), h, tape)
        self._post(mb, lambda: x, h)
        return h

    def _sampler(self, s, x, tape):
        self._pre(s, lambda: x)
        y = self._plain_conv(s.conv, x, tape, owner=s)
        self._post(s, lambda: x, y)
        return y

    def _updown_block(self, blk, x, tape, first_after_conv3=False):
        """first_after_conv3: the block's input is the output of a 3x3 stride-1 convolution (a resnet's conv2 or an upsampler's
        convolution), so the gradient that leaves the block is worth a bf16 image too"""
        self._pre(blk, lambda: x)
        h = x
        for i, r in enumerate(blk.resnets):  # a resnet after the first one continues the output of the previous one's conv2
            h = self._seg(lambda t, tp, r=r, i=i: self._resnet(r, t, tp, after_conv3=(i > 0 or first_after_conv3)), h, tape)
        extra = getattr(blk, "downsamplers", None) or getattr(blk, "upsamplers", None)
        if extra is not None:
            h = self._seg(lambda t, tp: self._sampler(extra[0], t, tp), h, tape)
        self._post(blk, lambda: x, h)
        return h

    def _norm_act_conv(self, owner, x, tape):
        norm, act, conv = owner.conv_norm_out, owner.conv_act, owner.conv_out
        self._no_hooks(act, "conv_act")
        st = self._gn(norm, x)
        y = self._conv(conv, x, XF_AFFINE_SILU, st)
        x16 = self._take16(tape is not None)
        if tape is not None:
            def bwd(d):
                g = self._conv_bwd(conv, x, d, XF_AFFINE_SILU, st, x16=x16, dx_to_gn=True, norm=norm)
                dx = self._gn_bwd(norm, x, g, st, True, None, feeds_conv3=True)
                self._done(conv)
                self._done(norm)
                return dx
            tape.append(bwd)
        return y

    # ------------------------------------------------------------------ encoder / decoder on NHWC
    def encoder_nhwc(self, x4: torch.Tensor, tape) -> torch.Tensor:
        enc = self.vae.encoder
        self._pre(enc, lambda: x4[..., :3])
        h = self._plain_conv(enc.conv_in, x4, tape, need_dx=False)
        for blk in enc.down_blocks:
            h = self._updown_block(blk, h, tape)
        h = self._mid(enc.mid_block, h, tape, after_conv3=True)  # behind down_blocks[-1]'s last resnet
        h = self._norm_act_conv(enc, h, tape)
        self._post(enc, lambda: x4[..., :3], h)
        return h

    def encode_nhwc(self, x4, tape):
        return self._plain_conv(self.vae.quant_conv, self.encoder_nhwc(x4, tape), tape)

    def decoder_nhwc(self, z: torch.Tensor, tape) -> torch.Tensor:
        dec = self.vae.decoder
        self._pre(dec, lambda: z)
        self._ckpt = bool(self.checkpoint_decoder) and tape is not None
        try:
            h = self._plain_conv(dec.conv_in, z, tape)
            h = self._mid(dec.mid_block, h, tape)
            for blk in dec.up_blocks:  # behind the mid block's last resnet / the previous block's upsampler convolution
                h = self._updown_block(blk, h, tape, first_after_conv3=True)
            h = self._norm_act_conv(dec, h, tape)
        finally:
            self._ckpt = False
        self._post(dec, lambda: z, h)
        return h

    def decode_nhwc(self, z, tape):
        return self.decoder_nhwc(self._plain_conv(self.vae.post_quant_conv, z, tape), tape)

    def run_tape(self, tape: list, grad: torch.Tensor, gtarget: torch.Tensor):
        self._gtarget = gtarget
        try:
            while tape:
                grad = tape.pop()(grad)
        finally:
            self._gtarget = None
        return grad

    # ------------------------------------------------------------------ fused train / eval step (fast path)
    def forward_backward(self, pixel_values: torch.Tensor, eps: Optional[torch.Tensor], kl_weight: float,
                         sample_posterior: bool = True, generator: Optional[torch.Generator] = None,
                         grad_scale: float = 1.0):
        """fwd + loss (train.py:289-291) + bwd; gradients are WRITTEN into arena.grad (no accumulation), scaled by
        `grad_scale` (= 1/gradient_accumulation_steps: accelerate divides the loss before backward, train.py:286,299).
        Returns dict(scalars[3]=mse,kl,total on device, reconstruction, moments, latents) as NHWC buffers."""
        self._require_gpu()
        pv = pixel_values.contiguous()
        with self._mode():
            x4 = ops.nchw_to_nhwc(pv, 4)
            tgt = ops.nchw_to_nhwc(pv, 3)
            te: list = []
            td: list = []
            mom = self.encode_nhwc(x4, te)
            e = self._eps_nhwc(eps, mom, sample_posterior, generator)
            z, klp = ops.sample_kl(mom, e)
            recon = self.decode_nhwc(z, td)
            scalars = ops.mse_kl_loss(recon, tgt, klp, kl_weight)
            drec = ops.mse_bwd(recon, tgt, grad_scale)
            self.arena.attach_grads()
            dz = self.run_tape(td, drec, self.arena.grad)
            dmom = ops.sample_kl_bwd(mom, e, dz, kl_weight * grad_scale)
            self.run_tape(te, dmom, self.arena.grad)
        if self.reducer is not None:
            self.reducer.ready(0)
        return {"scalars": scalars, "reconstruction": recon, "moments": mom, "latents": z, "kl_partial": klp}

    @torch.no_grad()
    def forward_eval(self, pixel_values: torch.Tensor, eps: Optional[torch.Tensor] = None, sample_posterior: bool = False,
                     kl_weight: float = 0.0, generator: Optional[torch.Generator] = None):
        self._require_gpu()
        pv = pixel_values.contiguous()
        x4 = ops.nchw_to_nhwc(pv, 4)
        tgt = ops.nchw_to_nhwc(pv, 3)
        with self._mode():
            mom = self.encode_nhwc(x4, None)
            e = self._eps_nhwc(eps, mom, sample_posterior, generator)
            z, klp = ops.sample_kl(mom, e)
            recon = self.decode_nhwc(z, None)
        scalars = ops.mse_kl_loss(recon, tgt, klp, kl_weight)
        return {"scalars": scalars, "reconstruction": recon, "moments": mom, "latents": z, "kl_partial": klp}

    @staticmethod
    def _eps_nhwc(eps, mom, sample_posterior, generator):
        if not sample_posterior:
            return None
        B, h, w, L2 = mom.shape
        if eps is None:
            return torch.randn((B, h, w, L2 // 2), device=mom.device, dtype=torch.float32, generator=generator)
        assert eps.shape == (B, L2 // 2, h, w), eps.shape
        return ops.nchw_to_nhwc(eps.to(mom.device).contiguous())

    # ------------------------------------------------------------------ autograd-compatible path
    def encode_autograd(self, x: torch.Tensor) -> torch.Tensor:
        self._require_gpu()
        return _EncodeFn.apply(self, torch.is_grad_enabled(), x, *self._params_of(self.vae.encoder, self.vae.quant_conv))

    def decode_autograd(self, z: torch.Tensor) -> torch.Tensor:
        self._require_gpu()
        return _DecodeFn.apply(self, torch.is_grad_enabled(), z, *self._params_of(self.vae.post_quant_conv, self.vae.decoder))

    @staticmethod
    def _params_of(*mods):
        out = []
        for m in mods:
            out.extend(m.parameters())
        return out

    # ------------------------------------------------------------------ stand-alone module calls (inference only)
    def _standalone(self, x: torch.Tensor):
        self._require_gpu()
        if torch.is_grad_enabled() and x.requires_grad:
            raise RuntimeError("sub-module calls are inference-only; differentiate through SDXLVAEWrapper.forward / "
                               "vae.encode / vae.decode")
        if x.ndim == 3:  # [B,T,C] linear input
            return x.contiguous().unsqueeze(1)
        return x.permute(0, 2, 3, 1).contiguous()

    def leaf_conv(self, m, x):
        t = self._standalone(x)
        if t.shape[-1] == 3:
            t = torch.nn.functional.pad(t, (0, 1))
        with torch.no_grad():
            return self._conv(m, t, XF_NONE, None, out_dtype=torch.float32).permute(0, 3, 1, 2)

    def leaf_linear(self, m, x):
        t = self._standalone(x)
        with torch.no_grad():
            y = self._conv(m, t, XF_NONE, None, out_dtype=torch.float32)
        return y.squeeze(1) if x.ndim == 3 else y.permute(0, 3, 1, 2)

    def leaf_groupnorm(self, m, x):
        t = self._standalone(x)
        with torch.no_grad():
            st = ops.gn_stats(t, m.weight, m.bias, m.num_groups, m.eps)
            return ops.gn_apply(t, st, XF_AFFINE).permute(0, 3, 1, 2)

    def block_call(self, m, x):
        from . import autoencoder as A
        t = self._standalone(x)
        with torch.no_grad():
            if isinstance(m, A.ResnetBlock2D):
                y = self._resnet(m, t, None)
            elif isinstance(m, A.Attention):
                y = self._attention(m, t, None)
            elif isinstance(m, A.UNetMidBlock2D):
                y = self._mid(m, t, None)
            elif isinstance(m, (A.DownEncoderBlock2D, A.UpDecoderBlock2D)):
                y = self._updown_block(m, t, None)
            elif isinstance(m, (A.Downsample2D, A.Upsample2D)):
                y = self._sampler(m, t, None)
            elif isinstance(m, A.Encoder):
                y = self.encoder_nhwc(torch.nn.functional.pad(t, (0, 1)) if t.shape[-1] == 3 else t, None)
            elif isinstance(m, A.Decoder):
                y = self.decoder_nhwc(t, None)
            else:
                raise NotImplementedError(type(m).__name__)
        return ops.to_f32(y).permute(0, 3, 1, 2)


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng: Engine, record: bool, x: torch.Tensor, *params):
        need = record and any(p.requires_grad for p in params)
        tape = [] if need else None
        x4 = ops.nchw_to_nhwc(x.detach().to(dtype=torch.float32).contiguous(), 4)
        with eng._mode():
            mom = eng.encode_nhwc(x4, tape)
        ctx.eng, ctx.tape, ctx.params = eng, tape, params
        return mom.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dmom):
        eng = ctx.eng
        if ctx.tape is None:
            raise RuntimeError("backward through a forward that recorded no tape")
        gbuf = torch.zeros_like(eng.arena.grad)
        with eng._mode(repack=False):
            eng.run_tape(ctx.tape, dmom.permute(0, 2, 3, 1).contiguous(), gbuf)
        grads = [eng.arena.grad_view(p, gbuf) if p.requires_grad else None for p in ctx.params]
        return (None, None, None, *grads)


class _DecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng: Engine, record: bool, z: torch.Tensor, *params):
        need = record and (z.requires_grad or any(p.requires_grad for p in params))
        tape = [] if need else None
        zz = z.detach().to(dtype=torch.float32).permute(0, 2, 3, 1).contiguous()
        with eng._mode():
            rec = eng.decode_nhwc(zz, tape)
        ctx.eng, ctx.tape, ctx.params = eng, tape, params
        return rec.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, drec):
        eng = ctx.eng
        if ctx.tape is None:
            raise RuntimeError("backward through a forward that recorded no tape")
        gbuf = torch.zeros_like(eng.arena.grad)
        with eng._mode(repack=False):
            dz = eng.run_tape(ctx.tape, drec.permute(0, 2, 3, 1).contiguous(), gbuf)
        grads = [eng.arena.grad_view(p, gbuf) if p.requires_grad else None for p in ctx.params]
        return (None, None, dz.permute(0, 3, 1, 2), *grads)
