"""Precision modes and kernel variants of the split-precision path (through the C ABI).  Needs an MI355X: `-m gpu`.

* "f16" (AMBER_PREC_F16): plain fp16 operands, one MFMA per product - the reduced-precision option BASELINE's north
  star allows for bf16-class MLPs, tolerance 1e-2 (measured as max|a - b| / max|b|).
* The first edge step with the linear edge embedding fused in must agree with the two-kernel form (embedding kernel,
  then the ordinary edge step) to the fp32 bar.
"""
import os

import pytest
import torch

from amber_amd import engine, synthetic as syn
from conftest import close, golden_cases, load_golden, rel_err
from oracle import mpn_oracle as orc
from test_gpu_parity import _batch, _model_from_golden, _supervised

pytestmark = pytest.mark.gpu

F16_TOL = 1e-2
FP32_TOL = 1e-5


@pytest.fixture
def f16_mode():
    old = engine.default_precision
    engine.default_precision = "f16"
    yield
    engine.default_precision = old


@pytest.mark.parametrize("name", [n for n in golden_cases() if "tanh" not in n])
def test_f16_mode_within_1e_2_of_reference_golden(gpu_device, f16_mode, name):
    cfg, params, d = load_golden(name)
    model = _model_from_golden(cfg, params, d, gpu_device)
    with torch.no_grad():
        out = model(_batch(d, gpu_device))
    assert out.shape == d["out_sizing"].shape
    assert rel_err(out.cpu(), d["out_sizing"]) <= F16_TOL


@pytest.mark.parametrize("latent,steps,graphs,nx,ny", [(64, 20, 3, 60, 50), (128, 10, 2, 120, 120)])
def test_f16_mode_paper_and_synthetic_shapes(gpu_device, f16_mode, latent, steps, graphs, nx, ny):
    model, sd, cfg = _supervised(latent, steps, 4, 2, gpu_device)
    g = syn.grid_batch(num_graphs=graphs, nx=nx, ny=ny)
    want = orc.supervised_mpn_forward(sd, cfg, g.x, g.edge_attr, g.edge_index, g.mask_output)
    with torch.no_grad():
        got = model(g.to(gpu_device))
    err = rel_err(got.cpu(), want)
    assert err <= F16_TOL
    assert err > 1e-7  # the reduced-precision kernels really ran (the exact paths land near 1e-7)


@pytest.mark.parametrize("fe", [0, 1, 2, 4, 6])
def test_fused_edge_embedding_matches_separate_embedding(gpu_device, fe):
    """F_e <= 4 takes the fused first step, F_e = 6 the embedding kernel; with the switch set every width does."""
    old = engine.default_precision
    engine.default_precision = "f16x2"
    try:
        model, sd, cfg = _supervised(64, 3, 5, fe, gpu_device, seed=11)
        g = syn.grid_batch(num_graphs=2, nx=23, ny=17, in_node_features=5, in_edge_features=fe)
        want = orc.supervised_mpn_forward(sd, cfg, g.x, g.edge_attr, g.edge_index, g.mask_output)
        gd = g.to(gpu_device)
        with torch.no_grad():
            fused = model(gd).cpu()
            os.environ["AMBER_MPN_NO_FUSED_EMBED"] = "1"
            try:
                separate = model(gd).cpu()
            finally:
                del os.environ["AMBER_MPN_NO_FUSED_EMBED"]
        assert close(fused, want)
        assert close(separate, want)
        assert close(fused, separate)
    finally:
        engine.default_precision = old
