"""C-ABI: the library loads without a GPU and exports every symbol include/vaehip.h declares.
Arena / module-tree invariants that the kernels rely on (no compute calls here)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vaehip.lib import lib, SIGNATURES, LIB_PATH, EXPECTED_ABI
    hdr = open(os.path.join(ROOT, "include", "vaehip.h")).read()
    declared = set(re.findall(r"\b(vae_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vae_conv_geom", "vae_igemm_args", "vae_wgrad_args"}
    assert os.path.exists(LIB_PATH), "run __graft_entry__.build() first"
    dll = lib.load()
    for name in sorted(declared):
        assert hasattr(dll, name), f"{name} declared in vaehip.h but not exported"
    assert declared - {"vae_last_error", "vae_abi_version", "vae_sizeof_args"} == set(SIGNATURES), "python binding table out of sync with the header"
    from vaehip.lib import ConvGeom, IgemmArgs, WgradArgs
    import ctypes as C
    assert [dll.vae_sizeof_args(i) for i in range(3)] == [C.sizeof(ConvGeom), C.sizeof(IgemmArgs), C.sizeof(WgradArgs)]
    assert lib.abi_version() == EXPECTED_ABI
    assert isinstance(dll.vae_last_error(), bytes)


def test_argument_validation_needs_no_gpu():
    """shape/pointer checks run on the host before any launch: errors are loud and carry a message."""
    from vaehip.lib import lib, VaeHipError
    with pytest.raises(VaeHipError, match="add: bad args"):
        lib.call("vae_add", None, None, 0, None, None)
    with pytest.raises(VaeHipError, match="null args"):
        lib.call("vae_igemm_rows", None, None)


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from models.sdxl_vae_wrapper import SDXLVAEWrapper
    w = SDXLVAEWrapper("synthetic:3")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        w(torch.zeros(1, 3, 32, 32))
    with pytest.raises(FileNotFoundError):
        SDXLVAEWrapper("stabilityai/sdxl-vae")  # hub names need the network: loud, like sdxl_vae_wrapper.py:38-40


def test_module_tree_and_arena(tmp_path):
    import vae_oracle as vo
    from models.sdxl_vae_wrapper import SDXLVAEWrapper
    w = SDXLVAEWrapper("synthetic:3")
    vae = w.vae
    o = vo.OracleAutoencoderKL()
    assert {k: tuple(v.shape) for k, v in vae.named_parameters()} == {k: tuple(v.shape) for k, v in o.named_parameters()}
    assert sum(p.numel() for p in vae.parameters()) == 83_653_863
    assert abs(w.scaling_factor - 0.13025) < 1e-9 and vae.config.scaling_factor == w.scaling_factor
    # type identity the reference's plugins test with isinstance (classifier.py:56, deadneuron.py:62)
    gn = vae.get_submodule("encoder.down_blocks.0.resnets.0.norm1")
    assert isinstance(gn, torch.nn.GroupNorm) and gn.num_channels == 128 and isinstance(gn.weight, torch.nn.Parameter)
    assert isinstance(vae.get_submodule("decoder.up_blocks.2.resnets.0.conv_shortcut"), torch.nn.Conv2d)
    assert isinstance(vae.get_submodule("encoder.mid_block.attentions.0.to_q"), torch.nn.Linear)
    assert sum(isinstance(m, torch.nn.GroupNorm) for m in vae.modules()) == 52
    assert sum(isinstance(m, torch.nn.Conv2d) for m in vae.modules()) == 64
    # arena: every parameter is a leaf view of ONE buffer, conv weights OHWI in memory, 16-B aligned segments
    a = vae.arena
    assert a.owns(vae)
    cw = vae.encoder.down_blocks[1].resnets[0].conv1.weight
    assert cw.is_leaf and tuple(cw.shape) == (256, 128, 3, 3) and cw.permute(0, 2, 3, 1).is_contiguous()
    assert all(off % 4 == 0 for _, _, off, _ in a.entries)
    # in-place mutation by third parties (nudger.py:140) lands in the arena the kernels read
    with torch.no_grad():
        gn.weight.data[5] = 1.25
    assert float(a.flat[a.offset_of[id(gn.weight)] + 5]) == 1.25
    # registration order == execution order: decoder parameters sit above the encoder's (backward-monotone buckets)
    off = {n: o_ for n, _, o_, _ in a.entries}
    assert off["encoder.conv_in.weight"] == 0
    assert off["decoder.conv_in.weight"] < off["decoder.mid_block.resnets.1.conv2.weight"] < off["decoder.up_blocks.0.resnets.0.norm1.weight"] \
        < off["decoder.conv_out.weight"]
    assert off["quant_conv.weight"] < off["post_quant_conv.weight"] < off["decoder.conv_in.weight"]
    # save_pretrained / from_pretrained round trip with diffusers key names + legacy attention keys
    d = str(tmp_path / "vae")
    vae.save_pretrained(d)
    assert sorted(os.listdir(d)) == ["config.json", "diffusion_pytorch_model.safetensors"]
    w2 = SDXLVAEWrapper(d)
    assert all(torch.equal(x, y) for x, y in zip(vae.state_dict().values(), w2.vae.state_dict().values()))
    sd = {k.replace("to_q", "query").replace("to_k", "key").replace("to_v", "value").replace("to_out.0", "proj_attn"): v
          for k, v in vae.state_dict().items()}
    w2.vae.init_synthetic(9)
    w2.vae.load_state_dict(sd)
    assert torch.equal(w2.vae.encoder.mid_block.attentions[0].to_out[0].weight, vae.encoder.mid_block.attentions[0].to_out[0].weight)
    with pytest.raises(RuntimeError):
        w2.vae.load_state_dict({"bogus": torch.zeros(1)})


def test_bucket_planner():
    from vaehip.dp import plan_buckets
    b = plan_buckets(1003, 256)
    assert b[0][1] == 1003 and b[-1][0] == 0
    assert all(lo % 4 == 0 for lo, _ in b)
    assert all(b[i][0] == b[i + 1][1] for i in range(len(b) - 1))  # contiguous, from the end of the arena
    assert sum(hi - lo for lo, hi in b) == 1003
