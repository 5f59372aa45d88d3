"""Pin the CPU oracle (oracle/mpn_oracle.py) against outputs of the reference's own classes
(tests/golden/*.npz, written by tests/golden/make_golden.py in the build container)."""
import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden, rel_err
from oracle import mpn_oracle as orc

# Same arithmetic, same library (torch CPU): differences come only from op fusion order, so the pin is tight.
ORACLE_TOL = 2e-6


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_matches_reference(name):
    cfg, params, d = load_golden(name)
    x, e = orc.mpn_forward(params, cfg, d["x"], d["edge_attr"], d["edge_index"], prefix="mpn.")
    assert x.shape == d["out_node"].shape and e.shape == d["out_edge"].shape
    assert rel_err(x, d["out_node"]) <= ORACLE_TOL
    assert rel_err(e, d["out_edge"]) <= ORACLE_TOL
    out = orc.supervised_mpn_forward(params, cfg, d["x"], d["edge_attr"], d["edge_index"], d["mask_output"])
    assert out.shape == d["out_sizing"].shape
    assert rel_err(out, d["out_sizing"]) <= ORACLE_TOL


def test_golden_covers_edge_cases():
    """The fixtures must exercise zero in-degree rows, hub nodes, partial masks and constant embeddings."""
    seen_isolated = seen_hub = seen_mask = seen_f0 = False
    for name in golden_cases():
        cfg, _, d = load_golden(name)
        deg = np.bincount(d["edge_index"][1].numpy(), minlength=d["x"].shape[0])
        seen_isolated |= bool((deg == 0).any())
        seen_hub |= bool(deg.max() >= 12)
        seen_mask |= bool((~d["mask_output"]).any())
        seen_f0 |= cfg["in_node_features"] == 0 or cfg["in_edge_features"] == 0
    assert seen_isolated and seen_hub and seen_mask and seen_f0


def test_scatter_mean_zero_in_degree():
    src = torch.arange(12, dtype=torch.float32).view(4, 3)
    idx = torch.tensor([2, 0, 2, 2])
    out = orc.scatter_mean(src, idx, 4)
    assert torch.equal(out[1], torch.zeros(3)) and torch.equal(out[3], torch.zeros(3))
    assert torch.allclose(out[2], (src[0] + src[2] + src[3]) / 3)
    assert torch.equal(out[0], src[1])


def test_csr_oracle_is_stable_sort():
    rng = np.random.default_rng(0)
    ei = np.stack([rng.integers(0, 50, 400), rng.integers(0, 40, 400)])
    rowptr, perm, src_s, dst_s = orc.csr_oracle(ei, 50)
    t = torch.sort(torch.from_numpy(ei[1]), stable=True)
    assert np.array_equal(perm, t.indices.numpy().astype(np.int32))
    assert np.array_equal(dst_s, t.values.numpy().astype(np.int32))
    assert rowptr[0] == 0 and rowptr[-1] == 400 and np.all(np.diff(rowptr) >= 0)
    assert np.array_equal(np.diff(rowptr), np.bincount(ei[1], minlength=50))
    assert np.array_equal(src_s, ei[0][perm].astype(np.int32))


def test_segment_mean_oracle_equals_scatter_mean():
    rng = np.random.default_rng(1)
    ei = np.stack([rng.integers(0, 30, 200), rng.integers(0, 25, 200)])
    rowptr, perm, _, _ = orc.csr_oracle(ei, 30)
    vals = torch.randn(200, 8)
    a = orc.scatter_mean(vals, torch.from_numpy(ei[1]), 30)
    b = orc.segment_mean_oracle(vals[torch.from_numpy(perm.astype(np.int64))], rowptr)
    assert torch.equal(a, b)  # stable sort keeps the summation order, so this is bit-exact
