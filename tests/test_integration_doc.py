"""INTEGRATION.md's C-ABI example is executable documentation: a stale struct mirror there makes the library read past
the caller's buffer, so the snippet is extracted from the markdown and checked -- on CPU against the library's own
struct sizes and ABI version, on the GPU by running it against F.conv2d(F.silu(F.group_norm(x)))."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _snippet() -> str:
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    hits = [b for b in blocks if "vae_igemm_rows" in b]
    assert len(hits) == 1
    return hits[0]


def test_documented_struct_mirrors_match_the_library():
    import importlib
    L = importlib.import_module("vaehip.lib")
    src = _snippet()
    # the two class definitions of the snippet, executed on their own (no GPU call)
    start = src.index("class ConvGeom")
    end = src.index("# a stale mirror")
    ns = {"C": C}
    exec(src[start:end], ns)
    dll = L.lib.load()
    assert C.sizeof(ns["ConvGeom"]) == dll.vae_sizeof_args(0) == C.sizeof(L.ConvGeom)
    assert C.sizeof(ns["IgemmArgs"]) == dll.vae_sizeof_args(1) == C.sizeof(L.IgemmArgs)
    # same field names, order and C types as the binding the product uses
    assert [(n, t) for n, t in ns["IgemmArgs"]._fields_ if n != "g"] == [(n, t) for n, t in L.IgemmArgs._fields_ if n != "g"]
    assert [n for n, _ in ns["ConvGeom"]._fields_] == [n for n, _ in L.ConvGeom._fields_]
    m = re.search(r"vae_abi_version\(\) == (\d+)", src)
    assert m and int(m.group(1)) == dll.vae_abi_version() == L.EXPECTED_ABI
    assert "vae_sizeof_args(1) == C.sizeof(IgemmArgs)" in src
    # every header field is in the mirror
    hdr = open(os.path.join(ROOT, "include", "vaehip.h")).read()
    body = hdr[hdr.index("typedef struct vae_igemm_args {"):hdr.index("} vae_igemm_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.sub(r"[\*\s]", " ", part).split()[-1])
    assert names == [n for n, _ in ns["IgemmArgs"]._fields_]
    # files the document points at exist
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for path in re.findall(r"--config_path (\S+\.yaml)", md):
        assert os.path.exists(os.path.join(ROOT, path)), path
    n_exported = len(re.findall(r"^(?:int|int64_t|const char\*) (vae_\w+)\(", hdr, flags=re.M))
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"(\d+) entry points, ABI version (\d+)", design)
    assert m and int(m.group(1)) == n_exported and int(m.group(2)) == L.EXPECTED_ABI


@pytest.mark.gpu
def test_documented_example_runs_and_is_right(cuda):
    import torch.nn.functional as F
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        ns = {}
        torch.manual_seed(0)
        exec(_snippet(), ns)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    x, w, y = ns["x"], ns["w"], ns["y"]
    xn = x.permute(0, 3, 1, 2).cpu().double()
    ref = F.conv2d(F.silu(F.group_norm(xn, ns["G"], ns["gamma"].cpu().double(), ns["beta"].cpu().double(), 1e-6)),
                   w.permute(0, 3, 1, 2).cpu().double(), ns["bias"].cpu().double(), padding=1)
    got = y.permute(0, 3, 1, 2).cpu().double()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-5
