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
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and not f.startswith(("glue_", "graph_", "mesh_")))


def glue_cases():
    return sorted(f[5:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("glue_"))


def graph_cases():
    return sorted(f[6:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("graph_") and not f.startswith("graph_hier_"))


def hier_cases():
    """Fixtures of the hierarchical observation graph (inter-level edges), load with load_graph(name)."""
    return sorted(f[6:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("graph_hier_"))


def mesh_cases():
    return sorted(f[5:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("mesh_"))


def load_mesh(name):
    """One fixture of the step after the model (tests/golden/make_golden_mesh.py): (meta dict, {name: ndarray})."""
    z = np.load(os.path.join(GOLDEN_DIR, f"mesh_{name}.npz"), allow_pickle=False)
    return json.loads(str(z["meta"])), {k: z[k] for k in z.files if k != "meta"}


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


# ---- the parity bar: norm-wise AND element-wise -------------------------------------------------------------------
# north_star: "within 1e-5 relative error in fp32".  rel_err alone (max |a - b| / max |b|) would let small-magnitude
# outputs - sizing values far below the batch maximum - carry a large relative error unseen, so every parity check also
# holds each element to |a - b| <= tol |b| + (tol / 10) max |b| (the second term is the floor below which fp32 rounding
# of the sums that produce an element, which scale with the tensor and not with the element, cannot be relative).
_worst = {}  # test id -> worst (norm-wise error, element-wise ratio) seen


def elem_ratio(a: torch.Tensor, b: torch.Tensor, tol: float = 1e-5) -> float:
    """max over elements of |a - b| / (tol |b| + tol / 10 max |b|): <= 1 passes."""
    if b.numel() == 0:
        return 0.0 if a.numel() == 0 else float("inf")
    a, b = a.double(), b.double()
    bound = tol * b.abs() + 0.1 * tol * b.abs().max().clamp_min(1e-30)
    return float(((a - b).abs() / bound).max())


def close(a: torch.Tensor, b: torch.Tensor, tol: float = 1e-5) -> bool:
    """The parity check of the GPU tests: norm-wise within `tol` and element-wise within the bound of elem_ratio."""
    e, r = rel_err(a, b), elem_ratio(a, b, tol)
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    w = _worst.get(test, (0.0, 0.0))
    _worst[test] = (max(w[0], e), max(w[1], r))
    return e <= tol and r <= 1.0


def pytest_terminal_summary(terminalreporter):
    if not _worst:
        return
    top = sorted(_worst.items(), key=lambda kv: -kv[1][1])[:8]
    terminalreporter.write_line("parity margins (norm-wise error, element-wise |a-b| / (1e-5 |b| + 1e-6 max|b|)), worst tests:")
    for test, (e, r) in top:
        terminalreporter.write_line(f"  {e:.2e}  {r:.3f}  {test}")
    terminalreporter.write_line(f"  over {len(_worst)} tests: worst norm-wise {max(v[0] for v in _worst.values()):.2e}, "
                                f"worst element-wise ratio {max(v[1] for v in _worst.values()):.3f}")


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
