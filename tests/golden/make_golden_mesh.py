"""Golden vectors for the step after the model, from the REFERENCE's own functions (build container only).

Executes, unmodified, /root/reference/src/mesh_util/sizing_field_util.py (project_elements_to_vertices,
_mesh_sizing_field_to_num_elements) on duck-typed meshes carrying the MeshWrapper attributes those functions read.  The
module imports MeshWrapper for annotations only (its module pulls in scikit-fem, absent): registered as an empty module,
as in make_golden_graph.py.  torch_scatter.scatter_add (third party, absent, version unpinned) is the stand-in below:
out.scatter_add_(0, index, src) on a zero tensor - its published algorithm.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mesh.py   ->  tests/golden/mesh_*.npz
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

import src  # noqa: E402,F401
import torch_scatter  # noqa: E402  (the stand-in package)


def _scatter_add(src, index, dim=0, dim_size=None):
    assert dim == 0 and src.dim() == 1
    out = torch.zeros(int(dim_size), dtype=src.dtype)
    return out.scatter_add_(0, index, src)


torch_scatter.scatter_add = _scatter_add

for _name, _path in (("src.mesh_util", "/root/reference/src/mesh_util"), ("src.tasks", "/root/reference/src/tasks"),
                     ("src.tasks.domains", "/root/reference/src/tasks/domains")):
    _m = types.ModuleType(_name)
    _m.__path__ = [_path]
    sys.modules[_name] = _m
_m = types.ModuleType("src.tasks.domains.mesh_wrapper")
_m.MeshWrapper = type("MeshWrapper", (), {})  # annotation target only
sys.modules["src.tasks.domains.mesh_wrapper"] = _m

from src.mesh_util.sizing_field_util import _mesh_sizing_field_to_num_elements, project_elements_to_vertices  # noqa: E402  (REFERENCE)


class Mesh:
    def __init__(self, t, positions, volumes, edges):
        self.t, self.vertex_positions, self.simplex_volumes, self.mesh_edges = t, positions, volumes, edges
        self.nvertices = positions.shape[0]

    def dim(self):
        return self.vertex_positions.shape[1]


def tri_mesh(nx, ny, rng, isolated=0):
    idx = np.arange(nx * ny).reshape(ny, nx)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    t = np.concatenate([np.stack([a, b, d]), np.stack([a, d, c])], axis=1)
    t = t[:, rng.permutation(t.shape[1])]  # element order as a mesher leaves it: not sorted by vertex
    pos = np.stack(np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1.4, ny)), -1).reshape(-1, 2) + rng.normal(0, 0.004, (nx * ny, 2))
    if isolated:
        pos = np.concatenate([pos, rng.random((isolated, 2))])  # vertices no element uses: 0 / 0
    p = pos[t]
    vol = 0.5 * np.abs((p[1, :, 0] - p[0, :, 0]) * (p[2, :, 1] - p[0, :, 1]) - (p[2, :, 0] - p[0, :, 0]) * (p[1, :, 1] - p[0, :, 1]))
    e = np.concatenate([t[[0, 1]], t[[1, 2]], t[[2, 0]]], axis=1)
    return Mesh(t, pos, vol, np.unique(np.sort(e, axis=0), axis=1))


def tet_mesh(n, rng):
    idx = np.arange(n ** 3).reshape(n, n, n)
    c = [idx[:-1, :-1, :-1], idx[:-1, :-1, 1:], idx[:-1, 1:, :-1], idx[:-1, 1:, 1:], idx[1:, :-1, :-1], idx[1:, :-1, 1:],
         idx[1:, 1:, :-1], idx[1:, 1:, 1:]]
    c = [v.ravel() for v in c]
    tets = [(0, 1, 3, 7), (0, 1, 5, 7), (0, 2, 3, 7), (0, 2, 6, 7), (0, 4, 5, 7), (0, 4, 6, 7)]  # Kuhn split of a cube
    t = np.concatenate([np.stack([c[i] for i in q]) for q in tets], axis=1)
    t = t[:, rng.permutation(t.shape[1])]
    g = np.linspace(0, 1, n)
    pos = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + rng.normal(0, 0.01, (n ** 3, 3))
    p = pos[t]
    vol = np.abs(np.einsum("ij,ij->i", p[1] - p[0], np.cross(p[2] - p[0], p[3] - p[0]))) / 6.0
    e = np.concatenate([t[[i, j]] for i in range(4) for j in range(i + 1, 4)], axis=1)
    return Mesh(t, pos, vol, np.unique(np.sort(e, axis=0), axis=1))


def main():
    rng = np.random.default_rng(31)
    for name, mesh in (("tri_24x19", tri_mesh(24, 19, rng)), ("tri_iso_9x7", tri_mesh(9, 7, rng, isolated=3)), ("tet_7", tet_mesh(7, rng))):
        m, v = mesh.t.shape[1], mesh.nvertices
        element_values = rng.lognormal(-2.0, 0.7, m)
        with np.errstate(invalid="ignore"):
            projected = project_elements_to_vertices(mesh, element_values)  # REFERENCE
        vertex_field = rng.lognormal(-2.0, 0.9, v)  # rough on purpose: the gradation passes have something to smooth
        n_vertex = _mesh_sizing_field_to_num_elements(mesh, vertex_field, node_type="vertex", gradation=1.3)  # REFERENCE
        n_element = _mesh_sizing_field_to_num_elements(mesh, element_values, node_type="element")  # REFERENCE
        out = os.path.join(HERE, f"mesh_{name}.npz")
        np.savez_compressed(out, meta=json.dumps({"dim": mesh.dim(), "gradation": 1.3}), t=mesh.t.astype(np.int64),
                            positions=mesh.vertex_positions, volumes=mesh.simplex_volumes, edges=mesh.mesh_edges.astype(np.int64),
                            element_values=element_values, projected=np.asarray(projected), vertex_field=vertex_field,
                            n_vertex=np.float64(n_vertex), n_element=np.float64(n_element))
        print(f"{name}: M={m} V={v} edges={mesh.mesh_edges.shape[1]} n_vertex={n_vertex:.3f} n_element={n_element:.3f} -> "
              f"{os.path.getsize(out) // 1024} KiB")


if __name__ == "__main__":
    main()
