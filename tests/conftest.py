import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and not f.startswith(("glue_", "graph_")))


def glue_cases():
    return sorted(f[5:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("glue_"))


def graph_cases():
    return sorted(f[6:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("graph_") and not f.startswith("graph_hier_"))


def hier_cases():
    """Fixtures of the hierarchical observation graph (inter-level edges), load with load_graph(name)."""
    return sorted(f[6:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("graph_hier_"))


def load_graph(name):
    """One graph-construction fixture written by tests/golden/make_golden_graph.py: (meta dict, {name: ndarray})."""
    z = np.load(os.path.join(GOLDEN_DIR, f"graph_{name}.npz"), allow_pickle=False)
    return json.loads(str(z["meta"])), {k: z[k] for k in z.files if k != "meta"}


def load_glue(name):
    """One pre / post glue fixture written by tests/golden/make_golden_glue.py: (meta dict, {name: tensor})."""
    z = np.load(os.path.join(GOLDEN_DIR, f"glue_{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return meta, {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}


def load_golden(name):
    """Load one golden fixture written by tests/golden/make_golden.py (reference outputs)."""
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"), allow_pickle=False)
    cfg = json.loads(bytes(z["config"]).decode())
    params = {k[len("param/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    data = {k: torch.from_numpy(z[k]) for k in ("x", "edge_attr", "edge_index", "batch", "ptr", "mask_output",
                                                  "out_node", "out_edge", "out_sizing")}
    return cfg, params, data


def load_golden_grads(name):
    """(grad_out, {state_dict name: reference gradient}) of a fixture that holds the reference's autograd gradients,
    else None."""
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"), allow_pickle=False)
    if "grad_out" not in z.files:
        return None
    grads = {k[len("grad/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad/")}
    return torch.from_numpy(z["grad_out"]), grads


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a - b| / max |b|: error relative to the output's scale (the tolerance form used throughout)."""
    if b.numel() == 0:
        return 0.0 if a.numel() == 0 else float("inf")
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
