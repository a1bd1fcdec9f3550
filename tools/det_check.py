"""cross-process repeatability of the engine's gradients: prints one hash per forward_backward; run it several times (and
several copies at once) and compare the lines"""
import hashlib, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (os.path.join(ROOT, "vae-channel-dynamics_amd", "src"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import vae_oracle as vo
from models.sdxl_vae_wrapper import SDXLVAEWrapper
dev = torch.device("cuda", 0)
if os.environ.get("LDS_POISON"):
    # every library call is preceded by a kernel that fills all LDS with NaN patterns: consuming LDS that was never written
    # then shows as NaN ("!" behind the hash) instead of as a value that depends on what ran on the CU before
    import ctypes
    from vaehip.lib import lib as _lib
    _pz = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "liblds_poison.so"))
    _orig = _lib.call
    def _call(name, *a):
        rc = _pz.lds_poison(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_uint(0x7FC07FC0))
        assert rc == 0, rc
        return _orig(name, *a)
    _lib.call = _call
w = SDXLVAEWrapper("synthetic:7", device=dev)
hs = []
models = [w]
if os.environ.get("TWO_MODELS"):
    # a second replica allocated after the first has run (as tests/test_dp_gpu.py does): other allocator state
    x = vo.synthetic_pixels(2, 32, 42, 3).to(dev); e = vo.synthetic_eps(2, 32, 42, 3).to(dev)
    w.vae.engine.forward_backward(x, e, 1e-4)
    junk = [torch.full((n,), float("nan"), device=dev) for n in (1 << 20, 1 << 22, 1 << 24, 3 << 24)]
    del junk  # NaN-filled blocks back in the caching allocator: an uninitialised read shows up as NaN
    w2 = SDXLVAEWrapper("synthetic:7", device=dev)
    with torch.no_grad():
        w2.vae.arena.flat.copy_(w.vae.arena.flat)
    models = [w2]
for m in models:
    if os.environ.get("BF16"):
        m.vae.engine.set_precision("bf16")
    for r in range(4):
        x = vo.synthetic_pixels(2, 32, 42, 10 + r).to(dev); e = vo.synthetic_eps(2, 32, 42, 10 + r).to(dev)
        m.vae.engine.forward_backward(x, e, 1e-4)
        torch.cuda.synchronize()
        g = m.vae.arena.grad
        hs.append(hashlib.sha1(g.cpu().numpy().tobytes()).hexdigest()[:10] + ("!" if not bool(torch.isfinite(g).all()) else ""))
print(" ".join(hs))
