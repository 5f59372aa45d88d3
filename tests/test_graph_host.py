"""Graph construction, CPU side: the oracle against the reference-generated golden vectors (bit for bit) and the
product's refusal to build on the CPU."""
import numpy as np
import pytest
import torch

from amber_amd import graph_build
from conftest import graph_cases, hier_cases, load_graph
from oracle import graph_oracle as gro

CASES = graph_cases()


def test_fixtures_exist():
    assert len(CASES) >= 5


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    meta, d = load_graph(name)
    for g in range(meta["num_graphs"]):
        pairs, pos = d[f"g{g}_pairs"], d[f"g{g}_pos"]
        curv = None
        if "edge_curvature" in meta["names"] and meta["node_type"] == "vertex":
            curv = gro.curvature_per_pair(pairs.shape[1], d[f"g{g}_boundary"], d[f"g{g}_curv"])
        ea, ei = gro.mesh_edges(pairs, pos, pos.shape[0], meta["names"], meta["add_self_edges"], curv)
        assert np.array_equal(ei.numpy(), d[f"g{g}_edge_index"])
        assert np.array_equal(ea.numpy(), d[f"g{g}_edge_attr"])
        if f"g{g}_degree" in d:
            assert np.array_equal(gro.degree(pairs).astype(np.float32)[:, None], d[f"g{g}_degree"])


@pytest.mark.parametrize("name", hier_cases())
def test_inter_level_oracle_matches_reference_golden(name):
    meta, d = load_graph(name)
    ea, ei = gro.inter_graph_edges(d["fine_pos"], d["coarse_pos"], d["nearest"], meta["names"], meta["node_type"])
    assert np.array_equal(ei.numpy(), d["inter_edge_index"]) and np.array_equal(ea.numpy(), d["inter_edge_attr"])
    assert len(hier_cases()) >= 3


def test_collation_offsets():
    ea, ei, batch, ptr = gro.collate([torch.ones(3, 1), torch.zeros(2, 1)], [torch.tensor([[0, 1, 2], [1, 2, 0]]), torch.tensor([[0, 1], [1, 0]])],
                                     [3, 2])
    assert ei.tolist() == [[0, 1, 2, 3, 4], [1, 2, 0, 4, 3]] and batch.tolist() == [0, 0, 0, 1, 1] and ptr.tolist() == [0, 3, 5]
    assert ea.shape == (5, 1)


def test_cpu_device_fails_loudly():
    class M:
        mesh_edges = np.array([[0], [1]])
        vertex_positions = np.zeros((2, 2))
        num_vertices = 2

    with pytest.raises(RuntimeError, match="no CPU path"):
        graph_build.build_graph_batch([M()], ["euclidean_distance"], device="cpu")
    with pytest.raises(ValueError, match="not supported"):
        graph_build._mesh_arrays(M(), "pixel", [])
