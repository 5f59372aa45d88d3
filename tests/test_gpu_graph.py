"""Graph construction on the GPU (amber_mpn_build_graph through the C ABI) against the reference-generated golden
vectors and the oracle.  Needs an MI355X: run with `-m gpu`.

Bars: edge_index, batch, ptr, degree, distance_vector and edge_curvature are BIT-EXACT.  euclidean_distance is
sqrt(dx*dx + dy*dy (+ dz*dz)) with every operation rounded once on the GPU; torch's CPU norm kernel may fuse the
multiply-add, so that column is held to 1 ulp (1.2e-7 relative) instead."""
import numpy as np
import pytest
import torch

from amber_amd import csr as csr_mod, graph_build, synthetic as syn
from amber_amd.architecture.supervised_mpn import SupervisedMPN
from amber_amd.config import default_architecture
from conftest import graph_cases, hier_cases, load_graph, rel_err
from oracle import graph_oracle as gro
from oracle import mpn_oracle as orc

pytestmark = pytest.mark.gpu
CASES = graph_cases()


class Mesh:
    def __init__(self, pairs, pos, boundary, curv, element):
        if element:
            self.element_neighbors, self.element_midpoints, self.num_elements = pairs, pos, pos.shape[0]
        else:
            self.mesh_edges, self.vertex_positions, self.num_vertices = pairs, pos, pos.shape[0]
            self.boundary_edges, self.boundary_edge_curvatures = boundary, curv


def _meshes(meta, d):
    el = meta["node_type"] == "element"
    return [Mesh(d[f"g{g}_pairs"], d[f"g{g}_pos"], d[f"g{g}_boundary"], d[f"g{g}_curv"], el) for g in range(meta["num_graphs"])]


def _check_attr(got: torch.Tensor, want: torch.Tensor, names, dim):
    cols, c = [], 0
    for nme in ("distance_vector", "euclidean_distance", "edge_curvature"):  # the reference's column order
        if nme in names:
            w = dim if nme == "distance_vector" else 1
            cols.append((nme, c, c + w))
            c += w
    assert got.shape == want.shape and c == want.shape[1]
    for nme, a, b in cols:
        if nme == "euclidean_distance":
            err = (got[:, a:b].double() - want[:, a:b].double()).abs()
            assert (err <= 1.2e-7 * want[:, a:b].double().abs()).all(), nme
        else:
            assert torch.equal(got[:, a:b], want[:, a:b]), nme


@pytest.mark.parametrize("name", CASES)
def test_batch_matches_reference_per_graph_and_collated(gpu_device, name):
    meta, d = load_graph(name)
    names = meta["names"] if meta["node_type"] == "vertex" else [n for n in meta["names"] if n != "edge_curvature"]
    g = graph_build.build_graph_batch(_meshes(meta, d), meta["names"], node_type=meta["node_type"],
                                      add_self_edges=meta["add_self_edges"], device=gpu_device, with_degree=True)
    ea, ei, batch, ptr = gro.collate([torch.from_numpy(d[f"g{k}_edge_attr"]) for k in range(meta["num_graphs"])],
                                     [torch.from_numpy(d[f"g{k}_edge_index"]) for k in range(meta["num_graphs"])], meta["node_counts"])
    assert g.edge_index.dtype == torch.int64 and torch.equal(g.edge_index.cpu(), ei)
    assert torch.equal(g.batch.cpu(), batch) and torch.equal(g.ptr.cpu(), ptr)
    _check_attr(g.edge_attr.cpu(), ea, names, meta["dim"])
    if "g0_degree" in d:
        want_deg = torch.cat([torch.from_numpy(d[f"g{k}_degree"]).flatten() for k in range(meta["num_graphs"])])
        assert torch.equal(g.degree.cpu(), want_deg)
    # the CSR built in the same call equals the one built from the reference's edge_index
    ref_csr = csr_mod.build_csr(ei.to(gpu_device), int(ptr[-1]))
    for a, b in ((g.amber_csr.rowptr, ref_csr.rowptr), (g.amber_csr.perm, ref_csr.perm), (g.amber_csr.src_sorted, ref_csr.src_sorted)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", CASES)
def test_single_mesh_functions_mirror_the_reference(gpu_device, name):
    meta, d = load_graph(name)
    mesh = _meshes(meta, d)[0]
    fn = graph_build.get_mesh_element_edges if meta["node_type"] == "element" else graph_build.get_mesh_vertex_edges
    edge_attr, edge_index = fn(mesh, edge_feature_names=meta["names"], add_self_edges=meta["add_self_edges"], device=gpu_device)
    names = meta["names"] if meta["node_type"] == "vertex" else [n for n in meta["names"] if n != "edge_curvature"]
    assert torch.equal(edge_index.cpu(), torch.from_numpy(d["g0_edge_index"]))
    _check_attr(edge_attr.cpu(), torch.from_numpy(d["g0_edge_attr"]), names, meta["dim"])


def test_large_batch_vs_oracle_and_into_the_model(gpu_device):
    """200 ragged meshes (1.0 M directed edges) against the oracle; then the built batch goes straight into the model."""
    rng = np.random.default_rng(3)
    meshes, eas, eis, counts = [], [], [], []
    for _ in range(200):
        nx, ny = int(rng.integers(3, 60)), int(rng.integers(3, 60))
        pairs = syn.grid_triangulation_edges(nx, ny)
        pos = rng.normal(size=(nx * ny, 2))
        nb = np.sort(rng.choice(pairs.shape[1], size=5, replace=False))
        cv = rng.normal(size=5)
        meshes.append(Mesh(pairs, pos, nb, cv, False))
        ea, ei = gro.mesh_edges(pairs, pos, nx * ny, ["euclidean_distance", "edge_curvature"], True,
                                gro.curvature_per_pair(pairs.shape[1], nb, cv))
        eas.append(ea), eis.append(ei), counts.append(nx * ny)
    ea, ei, batch, ptr = gro.collate(eas, eis, counts)
    g = graph_build.build_graph_batch(meshes, ["euclidean_distance", "edge_curvature"], device=gpu_device, with_degree=True)
    assert torch.equal(g.edge_index.cpu(), ei) and torch.equal(g.batch.cpu(), batch)
    _check_attr(g.edge_attr.cpu(), ea, ["euclidean_distance", "edge_curvature"], 2)
    deg = torch.cat([torch.from_numpy(gro.degree(m.mesh_edges).astype(np.float32)) for m in meshes])
    assert torch.equal(g.degree.cpu(), deg)
    # node features: degree (device-built) + three host features, as the Laplace vertex graph has them
    torch.manual_seed(0)
    extra = torch.randn(int(ptr[-1]), 3)
    g.x = torch.cat([g.degree.unsqueeze(1) / 6.0, extra.to(gpu_device)], dim=1)
    cfg = default_architecture(latent_dimension=32, num_steps=2)
    model = SupervisedMPN(cfg, syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, 2))).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        got = model.to(gpu_device)(g).cpu()
    want = orc.supervised_mpn_forward(sd, dict(cfg, in_node_features=4, in_edge_features=2, embedding=None), g.x.cpu(), ea, ei, None)
    assert rel_err(got, want) <= 1e-5


@pytest.mark.parametrize("name", hier_cases())
def test_hierarchical_graph_matches_reference(gpu_device, name):
    """[current mesh | initial mesh + n | inter-level edges] as AmberData._extend_to_hierarchical_graph assembles it, from
    the reference's own per-part outputs; inter-level features (float64 arithmetic) bit-exact."""
    meta, d = load_graph(name)
    el = meta["node_type"] == "element"
    fine = Mesh(d["fine_pairs"], d["fine_pos"], d["fine_boundary"], d["fine_curv"], el)
    coarse = Mesh(d["coarse_pairs"], d["coarse_pos"], d["coarse_boundary"], d["coarse_curv"], el)
    nf, nc = meta["n_fine"], meta["n_coarse"]
    x, xi = torch.randn(nf, 3), torch.randn(nc, 3)
    g = graph_build.build_hierarchical_graph(fine, coarse, d["nearest"], meta["names"], node_type=meta["node_type"], device=gpu_device,
                                             x=x, initial_x=xi, topology_only=True)
    ea, ei, mask = gro.hierarchical_graph(torch.from_numpy(d["fine_edge_attr"]), torch.from_numpy(d["fine_edge_index"]), nf,
                                          torch.from_numpy(d["coarse_edge_attr"]), torch.from_numpy(d["coarse_edge_index"]), nc,
                                          torch.from_numpy(d["inter_edge_attr"]), torch.from_numpy(d["inter_edge_index"]))
    assert torch.equal(g.edge_index.cpu(), ei) and torch.equal(g.mask_output.cpu(), mask)
    n_inter = 2 * nf
    names = meta["names"] if not el else [n for n in meta["names"] if n != "edge_curvature"]
    _check_attr(g.edge_attr.cpu()[:-n_inter], ea[:-n_inter], names, meta["dim"])
    assert torch.equal(g.edge_attr.cpu()[-n_inter:], ea[-n_inter:])  # float64 features, one rounding: bit-exact
    want_x = torch.cat([torch.cat([x, torch.zeros(nf, 1)], 1), torch.cat([torch.zeros_like(xi), torch.ones(nc, 1)], 1)], 0)
    assert torch.equal(g.x.cpu(), want_x)
    ref_csr = csr_mod.build_csr(ei.to(gpu_device), nf + nc)
    assert torch.equal(g.amber_csr.rowptr, ref_csr.rowptr) and torch.equal(g.amber_csr.perm, ref_csr.perm)
    with pytest.raises(IndexError, match="outside its mesh"):
        graph_build.build_hierarchical_graph(fine, coarse, np.full(nf, nc), meta["names"], node_type=meta["node_type"], device=gpu_device)
    with pytest.raises(ValueError, match="one partner per current node"):
        graph_build.build_hierarchical_graph(fine, coarse, d["nearest"][:-1], meta["names"], node_type=meta["node_type"], device=gpu_device)


def test_edge_cases(gpu_device):
    empty = graph_build.build_graph_batch([], ["euclidean_distance"], device=gpu_device)
    assert empty.edge_index.shape == (2, 0) and empty.batch.numel() == 0
    lone = Mesh(np.zeros((2, 0), dtype=np.int64), np.zeros((3, 2)), np.zeros(0, dtype=np.int64), np.zeros(0), False)
    g = graph_build.build_graph_batch([lone], ["euclidean_distance", "edge_curvature"], device=gpu_device, with_degree=True)
    assert g.edge_index.cpu().tolist() == [[0, 1, 2], [0, 1, 2]] and float(g.edge_attr.abs().sum()) == 0 and g.degree.cpu().tolist() == [0, 0, 0]
    bad = Mesh(np.array([[0], [5]]), np.zeros((2, 2)), np.zeros(0, dtype=np.int64), np.zeros(0), False)
    with pytest.raises(IndexError, match="outside its mesh"):
        graph_build.build_graph_batch([bad], ["euclidean_distance"], device=gpu_device)
    with pytest.raises(ValueError, match="different dimension"):
        graph_build.build_graph_batch([lone, Mesh(np.zeros((2, 0), dtype=np.int64), np.zeros((2, 3)), None, None, True)][:1]
                                      + [Mesh(np.array([[0], [1]]), np.zeros((2, 3)), np.zeros(0, dtype=np.int64), np.zeros(0), False)],
                                      ["euclidean_distance"], device=gpu_device)
