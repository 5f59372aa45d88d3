"""Precision modes and kernel variants of the split-precision path (through the C ABI).  Needs an MI355X: `-m gpu`.

* "f16" (AMBER_PREC_F16): plain fp16 operands, one MFMA per product - the reduced-precision option BASELINE's north
  star allows for bf16-class MLPs, tolerance 1e-2 (measured as max|a - b| / max|b|).
* The first edge step with the linear edge embedding fused in must agree with the two-kernel form (embedding kernel,
  then the ordinary edge step) to the fp32 bar.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from amber_amd import engine, synthetic as syn
from conftest import close, golden_cases, load_golden, rel_err
from oracle import mpn_oracle as orc
from test_gpu_parity import _batch, _model_from_golden, _supervised

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
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


def _forward_in_child(tmp_path, tag, fe, **env):
    """The library reads its environment switches ONCE per process (api.hip: EnvSwitches), so a switch set in this process
    after the first library call changes nothing: every variant runs in a process of its own."""
    out = str(tmp_path / f"{tag}.npz")
    clean = {k: v for k, v in os.environ.items() if k not in ("AMBER_MPN_NO_FUSED_EMBED", "AMBER_MPN_NO_FUSED_HEAD")}
    r = subprocess.run([sys.executable, os.path.join(HERE, "_fwd_path_worker.py"), out, str(fe)], capture_output=True, text=True,
                       timeout=240, env=dict(clean, **env))
    assert r.returncode == 0 and "FWD_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.from_numpy(np.load(out)["pred"])


@pytest.mark.parametrize("fe", [0, 1, 2, 4, 6])
def test_fused_edge_embedding_matches_separate_embedding(gpu_device, tmp_path, fe):
    """F_e <= 4 takes the fused first step (and the node embedding folded into the first projections), F_e = 6 the embedding
    kernel for the edges; with AMBER_MPN_NO_FUSED_EMBED every width takes the separate kernels.  Both against the oracle,
    and - the proof that the switch took effect - the two results differ in at least one bit."""
    _, sd, cfg = _supervised(64, 3, 5, fe, gpu_device, seed=11)
    g = syn.grid_batch(num_graphs=2, nx=23, ny=17, in_node_features=5, in_edge_features=fe)
    want = orc.supervised_mpn_forward(sd, cfg, g.x, g.edge_attr, g.edge_index, g.mask_output)
    fused = _forward_in_child(tmp_path, "fused", fe)
    separate = _forward_in_child(tmp_path, "separate", fe, AMBER_MPN_NO_FUSED_EMBED="1")
    assert close(fused, want)
    assert close(separate, want)
    assert close(fused, separate)
    assert not torch.equal(fused, separate)  # other kernels, another summation order: the switch selected them


def test_unfused_head_matches_fused_head(gpu_device, tmp_path):
    """AMBER_MPN_NO_FUSED_HEAD: decoder + readout as a launch of their own instead of riding on the last node launch."""
    _, sd, cfg = _supervised(64, 3, 5, 2, gpu_device, seed=11)
    g = syn.grid_batch(num_graphs=2, nx=23, ny=17, in_node_features=5, in_edge_features=2)
    want = orc.supervised_mpn_forward(sd, cfg, g.x, g.edge_attr, g.edge_index, g.mask_output)
    fused = _forward_in_child(tmp_path, "head_fused", 2)
    separate = _forward_in_child(tmp_path, "head_separate", 2, AMBER_MPN_NO_FUSED_HEAD="1")
    assert close(fused, want) and close(separate, want) and close(fused, separate)
    assert not torch.equal(fused, separate)
