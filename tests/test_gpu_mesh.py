"""The step after the model on the device (amber_amd/sizing_field.py -> amber_mpn_project_elements_to_vertices /
amber_mpn_estimate_num_elements, through the C ABI) against the reference-generated fixtures and, at the size of a real
refinement step, against the oracle.  Needs an MI355X: run with `-m gpu`."""
import numpy as np
import pytest
import torch

from amber_amd import sizing_field as sf
from conftest import load_mesh, mesh_cases
from oracle import mesh_oracle as mo

pytestmark = pytest.mark.gpu


class _Mesh:
    def __init__(self, t, positions, volumes, edges):
        self.t, self.vertex_positions, self.simplex_volumes, self.mesh_edges = t, positions, volumes, edges
        self.nvertices = positions.shape[0]

    def dim(self):
        return self.vertex_positions.shape[1]


@pytest.mark.parametrize("name", mesh_cases())
def test_device_matches_reference_fixture(gpu_device, name):
    meta, d = load_mesh(name)
    mesh = _Mesh(d["t"], d["positions"], d["volumes"], d["edges"])
    got = sf.project_elements_to_vertices(mesh, d["element_values"], device=gpu_device)
    assert got.dtype == np.float64 and np.array_equal(got, d["projected"], equal_nan=True)  # bit for bit
    nv = sf.mesh_sizing_field_to_num_elements(mesh, d["vertex_field"], node_type="vertex", gradation=meta["gradation"], device=gpu_device)
    ne = sf.mesh_sizing_field_to_num_elements(mesh, d["element_values"], node_type="element", device=gpu_device)
    assert abs(nv - float(d["n_vertex"])) <= 1e-12 * abs(float(d["n_vertex"]))
    assert abs(ne - float(d["n_element"])) <= 1e-12 * abs(float(d["n_element"]))
    with pytest.raises(ValueError, match="Unsupported interpolation_type"):
        sf.mesh_sizing_field_to_num_elements(mesh, d["element_values"], node_type=None, device=gpu_device)


def test_refinement_step_sized_mesh_and_errors(gpu_device):
    """100k triangles (the size AMBER's meshes reach): projection bit-equal to the oracle, estimates within 1e-12, two runs
    bit-equal; out-of-range vertex ids are reported."""
    rng = np.random.default_rng(7)
    nx = ny = 225
    idx = np.arange(nx * ny).reshape(ny, nx)
    a, b, c, dd = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    t = np.concatenate([np.stack([a, b, dd]), np.stack([a, dd, c])], axis=1)
    t = t[:, rng.permutation(t.shape[1])]
    pos = np.stack(np.meshgrid(np.arange(nx), np.arange(ny)), -1).reshape(-1, 2).astype(float) + 0.2 * rng.random((nx * ny, 2))
    p = pos[t]
    vol = 0.5 * np.abs((p[1, :, 0] - p[0, :, 0]) * (p[2, :, 1] - p[0, :, 1]) - (p[2, :, 0] - p[0, :, 0]) * (p[1, :, 1] - p[0, :, 1]))
    e = np.unique(np.sort(np.concatenate([t[[0, 1]], t[[1, 2]], t[[2, 0]]], axis=1), axis=0), axis=1)
    mesh = _Mesh(t, pos, vol, e)
    vals = rng.lognormal(0.0, 0.6, t.shape[1])
    field = rng.lognormal(0.0, 0.8, pos.shape[0])
    got = sf.project_elements_to_vertices(mesh, vals, device=gpu_device)
    assert np.array_equal(got, mo.project_elements_to_vertices(t, vol, vals, pos.shape[0]))
    assert np.array_equal(got, sf.project_elements_to_vertices(mesh, vals, device=gpu_device))
    for node_type, f in (("vertex", field), ("element", vals)):
        want = mo.mesh_sizing_field_to_num_elements(t, vol, f, node_type, 2, e, pos)
        n1 = sf.mesh_sizing_field_to_num_elements(mesh, f, node_type=node_type, device=gpu_device)
        assert abs(n1 - want) <= 1e-12 * abs(want)
        assert n1 == sf.mesh_sizing_field_to_num_elements(mesh, f, node_type=node_type, device=gpu_device)  # deterministic
    bad = _Mesh(np.array([[0], [1], [9]]), np.zeros((3, 2)), np.ones(1), np.zeros((2, 0), dtype=np.int64))
    with pytest.raises(IndexError):
        sf.project_elements_to_vertices(bad, np.ones(1), device=gpu_device)
    with pytest.raises(ValueError, match="one value and one volume per element"):
        sf.project_elements_to_vertices(mesh, vals[:-1], device=gpu_device)
    # the estimate validates its vertex ids too: a bad id in mesh.t or in mesh.mesh_edges raises, as numpy would
    with pytest.raises(IndexError):
        sf.mesh_sizing_field_to_num_elements(bad, np.ones(3), node_type="vertex", device=gpu_device)
    e_bad = e.copy()
    e_bad[1, 3] = pos.shape[0] + 4
    with pytest.raises(IndexError):
        sf.mesh_sizing_field_to_num_elements(_Mesh(t, pos, vol, e_bad), field, node_type="vertex", device=gpu_device)
    e_neg = e.copy()
    e_neg[0, 0] = -1
    with pytest.raises(IndexError):
        sf.mesh_sizing_field_to_num_elements(_Mesh(t, pos, vol, e_neg), field, node_type="vertex", device=gpu_device)
    # a nan in the vertex field propagates through the gradation maximum (np.maximum.at) into the estimate
    f_nan = field.copy()
    f_nan[5] = np.nan
    assert np.isnan(mo.mesh_sizing_field_to_num_elements(t, vol, f_nan, "vertex", 2, e, pos))
    assert np.isnan(sf.mesh_sizing_field_to_num_elements(mesh, f_nan, node_type="vertex", device=gpu_device))
    # tensors already on the GPU are used where they are
    tm = _Mesh(torch.from_numpy(t).to(gpu_device), pos, torch.from_numpy(vol).to(gpu_device), e)
    out = sf.project_elements_to_vertices(tm, torch.from_numpy(vals).to(gpu_device), device=gpu_device, as_tensor=True)
    assert out.is_cuda and np.array_equal(out.cpu().numpy(), got)
