"""CPU checks of the C-ABI boundary: the library loads without a GPU and exports every symbol
include/amber_mpn.h declares; descriptor validation and size queries work on the host."""
import ctypes as C
import os
import re

import pytest

from amber_amd import _lib, engine

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "amber_mpn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(amber_mpn_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 14
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/amber_mpn.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "amber_amd/_lib.py must bind exactly the declared entry points"


def test_abi_version_and_struct_size():
    lib = _lib.load()
    assert lib.amber_mpn_abi_version() == 2
    assert C.sizeof(_lib.AmberMpnDesc) == 17 * 4


def _desc(**kw):
    base = dict(latent=64, num_steps=2, num_step_repeats=1, mlp_layers=2, activation="leakyrelu", residual="inner",
                layer_norm="inner", in_node_features=5, in_edge_features=2)
    base.update(kw)
    return engine.make_desc(**base)


def test_param_count_matches_survey_formula():
    """7L^2 + 8L per block + L(F_n+1) + L(F_e+1) [+ 2(L^2+L) + L + 1 with the head];
    L=64, 2 steps, F_n=5, F_e=2 -> 58,944 MPN parameters (verified against the instantiated reference)."""
    lib = _lib.load()
    d = _desc()
    assert lib.amber_mpn_param_floats(C.byref(d)) == 58944
    d = _desc(has_head=True, decoder_layers=2)
    assert lib.amber_mpn_param_floats(C.byref(d)) == 58944 + 2 * (64 * 64 + 64) + 64 + 1
    # packed: per step 3 + 1 edge matrices, 2 + 1 node matrices; 2 decoder matrices in the exact format (stand-alone head,
    # training) and, in the split-precision modes, 2 more in the format of the node launch the head rides on; each L*L body
    # + 8-float header
    # ... and the folded tables of the node embedding + first projections, [3][F_n + 1][L] floats rounded up to 64
    heads = 4 if engine.default_precision != "fp32" else 2
    fold = (3 * (5 + 1) * 64 + 63) // 64 * 64
    assert lib.amber_mpn_packed_floats(C.byref(d)) == (2 * 7 + heads) * (64 * 64 + 8) + fold
    d32 = _desc(has_head=True, decoder_layers=2, precision="fp32")
    assert lib.amber_mpn_packed_floats(C.byref(d32)) == (2 * 7 + 2) * (64 * 64 + 8) + fold
    dmlp = _desc(has_head=True, decoder_layers=2, precision="fp32", embed_layers=2)  # MLP embedding: no fold
    assert lib.amber_mpn_packed_floats(C.byref(dmlp)) == (2 * 7 + 2 + 2) * (64 * 64 + 8)


def test_descriptor_validation_reports_errors():
    lib = _lib.load()
    bad = _desc(latent=48)
    assert lib.amber_mpn_param_floats(C.byref(bad)) == -1
    assert b"latent_dimension 48" in lib.amber_mpn_last_error()
    assert lib.amber_mpn_workspace_bytes(C.byref(bad), 10, 10) == 0
    with pytest.raises(ValueError):
        _desc(activation="swiglu")


def test_workspace_size_formula():
    lib = _lib.load()
    d = _desc(latent=128)
    n, e = 1000, 7000
    tiles = (e + 15) // 16  # partial-sum slots: one per 16-edge reduction tile
    a256 = lambda b: (b + 255) // 256 * 256  # noqa: E731
    expect = 4 * a256(n * 128 * 4) + 2 * a256(tiles * 128 * 4) + a256(e * 128 * 4) + 256
    assert lib.amber_mpn_workspace_bytes(C.byref(d), n, e) == expect


def test_null_pointer_arguments_are_rejected_without_touching_the_gpu():
    lib = _lib.load()
    d = _desc()
    rc = lib.amber_mpn_forward(C.byref(d), 0, 0, 10, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    assert rc == -1 and b"NULL" in lib.amber_mpn_last_error()
    rc = lib.amber_mpn_csr_build(0, -1, 5, 0, 0, 0, 0, 0, 0, 0, 0)
    assert rc == -1
