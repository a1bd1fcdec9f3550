"""ctypes loader for libvaehip.so (C ABI declared in include/vaehip.h)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VAEHIP_LIB: another build of the library (development: instrumented builds such as csrc/libvaehip_timing.so)
LIB_PATH = os.environ.get("VAEHIP_LIB") or os.path.normpath(os.path.join(_HERE, "..", "..", "csrc", "libvaehip.so"))


class VaeHipError(RuntimeError):
    pass


class ConvGeom(C.Structure):
    _fields_ = [("B", C.c_int32), ("Hs", C.c_int32), ("Ws", C.c_int32), ("Cs", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("taps", C.c_int32), ("stride", C.c_int32),
                ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("mode", C.c_int32)]


_fp = C.c_void_p


class IgemmArgs(C.Structure):
    _fields_ = [("A", _fp), ("W", _fp), ("C", _fp), ("bias", _fp), ("res", _fp), ("scale", _fp), ("shift", _fp),
                ("track", _fp), ("g", ConvGeom), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("ldc", C.c_int32), ("sn", C.c_int64), ("sk", C.c_int64), ("st", C.c_int64),
                ("batch", C.c_int32), ("sAb", C.c_int64), ("sWb", C.c_int64), ("sCb", C.c_int64),
                ("xf", C.c_int32), ("alpha", C.c_float), ("prec", C.c_int32), ("Wh", _fp), ("A16", _fp),
                ("tapmask", C.c_int32), ("a_step", C.c_int32), ("a_oy", C.c_int32), ("a_ox", C.c_int32),
                ("c_step", C.c_int32), ("c_oy", C.c_int32), ("c_ox", C.c_int32),
                ("gstat", _fp), ("gstat_groups", C.c_int32), ("Wu", _fp), ("out_bf16", C.c_int32),
                ("a_bf16", C.c_int32), ("res_bf16", C.c_int32),
                ("gnb_x", _fp), ("gnb_mean", _fp), ("gnb_rstd", _fp), ("gnb_gamma", _fp), ("gnb_beta", _fp), ("gnb_ws", _fp),
                ("gnb_groups", C.c_int32), ("gnb_silu", C.c_int32), ("gnb_x_bf16", C.c_int32)]


class WgradArgs(C.Structure):
    _fields_ = [("dY", _fp), ("X", _fp), ("out", _fp), ("partial", _fp), ("bias_partial", _fp), ("scale", _fp), ("shift", _fp),
                ("g", ConvGeom), ("M", C.c_int32), ("N", C.c_int32), ("ldy", C.c_int32), ("npix", C.c_int32),
                ("nsplit", C.c_int32), ("batch", C.c_int32), ("sYb", C.c_int64), ("sXb", C.c_int64),
                ("sOb", C.c_int64), ("xf", C.c_int32), ("alpha", C.c_float), ("prec", C.c_int32), ("X16", _fp), ("dY16", _fp),
                ("tapmask", C.c_int32), ("y_step", C.c_int32), ("y_oy", C.c_int32), ("y_ox", C.c_int32),
                ("x_bf16", C.c_int32), ("y_bf16", C.c_int32)]


i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
EXPECTED_ABI = 12  # vae_abi_version() of the library these structures and signatures describe

# name -> argtypes (every function returns int); must list EVERY symbol of include/vaehip.h
SIGNATURES = {
    "vae_set_option": [C.c_char_p, i32],
    "vae_get_option": [C.c_char_p],
    "vae_igemm_rows": [C.POINTER(IgemmArgs), vp],
    "vae_conv_gstat_chunks": [C.POINTER(IgemmArgs)],
    "vae_conv_gnb_chunks": [C.POINTER(IgemmArgs)],
    "vae_conv_phase_ok": [C.POINTER(IgemmArgs)],
    "vae_conv_io16_ok": [C.POINTER(IgemmArgs)],
    "vae_wgrad_io16_ok": [C.POINTER(WgradArgs)],
    "vae_add_bf16": [vp, vp, i64, vp, vp],
    "vae_unpack_bf16": [vp, i64, vp, vp],
    "vae_wino_ok": [C.POINTER(IgemmArgs)],
    "vae_wino_weight_floats": [C.POINTER(IgemmArgs)],
    "vae_wino_weights": [C.POINTER(IgemmArgs), vp, vp],
    "vae_bf16_grad_image_ok": [C.POINTER(ConvGeom), i32, i32],
    "vae_upconv_phase_weights": [vp, i32, i32, vp, vp],
    "vae_wgrad_phase_ok": [C.POINTER(WgradArgs)],
    "vae_wgrad_wino_plan": [C.POINTER(WgradArgs), C.POINTER(i32)],
    "vae_wgrad_wino": [C.POINTER(WgradArgs), vp],
    "vae_wgrad_wino_reduce": [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "vae_wgrad_wino_positions": [C.POINTER(WgradArgs)],
    "vae_upconv_fold_wgrad": [vp, vp, i32, i32, vp, vp, vp],
    "vae_wgrad": [C.POINTER(WgradArgs), vp],
    "vae_xf_fusable_rows": [C.POINTER(ConvGeom), i32, i32],
    "vae_wgrad_plan": [C.POINTER(WgradArgs), C.POINTER(i32), C.POINTER(i32)],
    "vae_igemm_kernel_name": [C.POINTER(IgemmArgs), C.c_char_p, i32],
    "vae_wgrad_kernel_name": [C.POINTER(WgradArgs), C.c_char_p, i32],
    "vae_reduce_splits": [vp, i32, i64, vp, vp],
    "vae_reduce_splits2": [vp, i32, i64, vp, vp, i32, vp, vp],
    "vae_gn_stats_partial": [vp, i32, i32, i32, i32, i32, i32, vp, vp],
    "vae_gn_stats_final": [vp, i32, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp],
    "vae_gn_apply": [vp, i32, vp, vp, i32, i32, i32, i32, vp, vp],
    "vae_gn_apply_bf16": [vp, i32, vp, vp, i32, i32, i32, i32, vp, vp],
    "vae_bf16_act_image_ok": [C.POINTER(ConvGeom), i32, i32],
    "vae_gn_track_partial": [vp, i32, vp, vp, i32, i32, i32, i32, vp, vp],
    "vae_track_final": [vp, i32, i32, f32, vp, vp],
    "vae_gn_bwd_partial": [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "vae_gn_bwd_final": [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp],
    "vae_gn_bwd_apply": [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp],
    "vae_attn_supported": [i32, i32],
    "vae_attn_fwd": [vp, vp, vp, i32, i32, i32, f32, i32, vp, vp, vp],
    "vae_attn_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp],
    "vae_softmax_rows": [vp, i64, i32, vp],
    "vae_softmax_bwd_rows": [vp, vp, i64, i32, vp],
    "vae_sample_kl": [vp, vp, i32, i32, i32, vp, vp, vp],
    "vae_mse_partial": [vp, vp, i64, vp, i32, vp],
    "vae_loss_final": [vp, i32, i64, vp, i32, i32, f32, vp, vp],
    "vae_mse_bwd": [vp, vp, i64, f32, vp, vp],
    "vae_sample_kl_bwd": [vp, vp, vp, i32, i32, i32, f32, vp, vp],
    "vae_nchw_to_nhwc": [vp, i32, i32, i32, i32, vp, vp],
    "vae_nhwc_to_nchw": [vp, i32, i32, i32, vp, vp],
    "vae_sumpool2x2": [vp, i32, i32, i32, i32, vp, vp],
    "vae_add": [vp, vp, i64, vp, vp],
    "vae_pack_bf16": [vp, i64, vp, vp],
    "vae_preprocess_u8": [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp],
    "vae_sqnorm": [vp, i64, vp, i32, vp, vp],
    "vae_adamw": [vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, f32, f32, i32, vp],
    "vae_dead_scan_chunk": [],
    "vae_dead_scan": [vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp],
    "vae_dead_scan_adaptive": [vp, vp, vp, i32, i32, f32, i32, vp, vp, vp, vp],
}


class _Lib:
    """lazy loader; attribute access returns a checked wrapper around the C symbol."""

    def __init__(self):
        self._dll = None

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise VaeHipError(
                    f"libvaehip.so not found at {LIB_PATH}; build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
            # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 but loads it lazily.  If this
            # library were opened first, its libamdhip64 dependency would resolve to the system copy and the process
            # would end up with two runtimes (kernels launched through the second find "no ROCm-capable device").
            # Opening torch's copy first makes the dependency resolve to it.
            import torch
            bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
            if os.path.exists(bundled):
                C.CDLL(bundled, mode=C.RTLD_GLOBAL)
            dll = C.CDLL(LIB_PATH)
            dll.vae_last_error.restype = C.c_char_p
            dll.vae_last_error.argtypes = []
            dll.vae_abi_version.restype = C.c_int
            if dll.vae_abi_version() != EXPECTED_ABI:
                raise VaeHipError(f"{LIB_PATH} has ABI version {dll.vae_abi_version()}, this binding expects {EXPECTED_ABI}: "
                                  f"rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
            dll.vae_sizeof_args.restype = C.c_int
            dll.vae_sizeof_args.argtypes = [C.c_int32]
            for which, mirror in enumerate((ConvGeom, IgemmArgs, WgradArgs)):
                if dll.vae_sizeof_args(which) != C.sizeof(mirror):
                    raise VaeHipError(f"{mirror.__name__}: ctypes mirror is {C.sizeof(mirror)} bytes, the library's struct "
                                      f"{dll.vae_sizeof_args(which)} -- lib.py and include/vaehip.h are out of sync")
            for name, argt in SIGNATURES.items():
                fn = getattr(dll, name)
                fn.restype = C.c_int64 if name == "vae_wino_weight_floats" else C.c_int
                fn.argtypes = argt
            self._dll = dll
        return self._dll

    def call(self, name, *args):
        dll = self.load()
        rc = getattr(dll, name)(*args)
        if rc != 0:
            raise VaeHipError(f"{name} failed ({rc}): {dll.vae_last_error().decode()}")

    def query(self, name, *args) -> int:
        """functions that return a value instead of a status"""
        return getattr(self.load(), name)(*args)

    def abi_version(self) -> int:
        return self.load().vae_abi_version()


lib = _Lib()
