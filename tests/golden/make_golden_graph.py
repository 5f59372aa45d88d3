"""Golden vectors for graph construction, from the REFERENCE's own mesh_to_graph functions (build container only).

Executes, unmodified, /root/reference/src/mesh_util/transforms/mesh_to_graph.py (get_mesh_vertex_edges,
get_mesh_element_edges, get_mesh_vertex_features) on duck-typed meshes: plain objects carrying the MeshWrapper
attributes those functions read.  mesh_to_graph.py imports MeshWrapper and FeatureProvider for type annotations only;
their modules pull in scikit-fem / gmsh (absent), so `src.tasks.domains.mesh_wrapper` and
`src.tasks.features.feature_provider` are registered as empty modules exposing those two names, and `src.mesh_util`,
`src.tasks` as bare namespace packages.  torch_geometric.data.Data comes from tests/golden/_shims.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_graph.py   ->  tests/golden/graph_*.npz
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

import src  # noqa: E402,F401

for _name, _path in (("src.mesh_util", "/root/reference/src/mesh_util"), ("src.mesh_util.transforms", "/root/reference/src/mesh_util/transforms"),
                     ("src.tasks", "/root/reference/src/tasks"), ("src.tasks.domains", "/root/reference/src/tasks/domains"),
                     ("src.tasks.features", "/root/reference/src/tasks/features")):
    _m = types.ModuleType(_name)
    _m.__path__ = [_path]
    sys.modules[_name] = _m
for _name, _attr in (("src.tasks.domains.mesh_wrapper", "MeshWrapper"), ("src.tasks.features.feature_provider", "FeatureProvider")):
    _m = types.ModuleType(_name)
    setattr(_m, _attr, type(_attr, (), {}))  # annotation target only
    sys.modules[_name] = _m

from src.mesh_util.transforms.mesh_to_graph import (  # noqa: E402  (REFERENCE)
    get_inter_graph_edges, get_mesh_element_edges, get_mesh_vertex_edges, get_mesh_vertex_features)

from amber_amd import synthetic as syn  # noqa: E402


class Mesh:
    """The MeshWrapper attributes mesh_to_graph.py reads."""

    def __init__(self, pairs, positions, boundary=None, curvatures=None, element=False):
        if element:
            self.element_neighbors, self.element_midpoints, self.num_elements = pairs, positions, positions.shape[0]
        else:
            self.mesh_edges, self.vertex_positions, self.num_vertices = pairs, positions, positions.shape[0]
            self.p = positions.T
            self.boundary_edges, self.boundary_edge_curvatures = boundary, curvatures

    def dim(self):
        return (self.element_midpoints if hasattr(self, "element_midpoints") else self.vertex_positions).shape[1]


def grid_mesh(nx, ny, dim, rng):
    pairs = syn.grid_triangulation_edges(nx, ny)
    n = nx * ny
    xs, ys = np.meshgrid(np.linspace(0, 1, nx), np.linspace(0, 1.3, ny))
    cols = [xs.ravel() + rng.normal(0, 0.01, n), ys.ravel() + rng.normal(0, 0.01, n)]
    if dim == 3:
        cols.append(rng.normal(0, 0.2, n))
    pos = np.stack(cols, axis=1)  # float64, as scikit-fem holds it
    nb = rng.choice(pairs.shape[1], size=max(3, pairs.shape[1] // 9), replace=False)
    return pairs, pos, np.sort(nb), rng.normal(0, 2.0, nb.shape[0])


def main():
    rng = np.random.default_rng(7)
    cases = {
        "vertex2d_all": dict(dim=2, names=["distance_vector", "euclidean_distance", "edge_curvature"], self_edges=True, grids=[(7, 5), (4, 9), (3, 3)]),
        "vertex2d_default": dict(dim=2, names=["euclidean_distance", "edge_curvature"], self_edges=True, grids=[(12, 11), (5, 4)]),
        "vertex3d_noself": dict(dim=3, names=["distance_vector", "euclidean_distance"], self_edges=False, grids=[(6, 6), (5, 8)]),
        "element3d": dict(dim=3, names=["euclidean_distance", "distance_vector"], self_edges=True, grids=[(9, 4)], element=True),
        "vertex2d_large": dict(dim=2, names=["euclidean_distance", "edge_curvature"], self_edges=True, grids=[(90, 70)]),
    }
    for name, c in cases.items():
        arrays, counts = {}, []
        for g, (nx, ny) in enumerate(c["grids"]):
            pairs, pos, nb, curv = grid_mesh(nx, ny, c["dim"], rng)
            element = c.get("element", False)
            mesh = Mesh(pairs, pos, nb, curv, element=element)
            fn = get_mesh_element_edges if element else get_mesh_vertex_edges
            edge_attr, edge_index = fn(mesh, edge_feature_names=c["names"], add_self_edges=c["self_edges"])
            arrays[f"g{g}_pairs"], arrays[f"g{g}_pos"] = pairs, pos
            arrays[f"g{g}_boundary"], arrays[f"g{g}_curv"] = nb, curv
            arrays[f"g{g}_edge_attr"], arrays[f"g{g}_edge_index"] = edge_attr.numpy(), edge_index.numpy()
            if not element:
                deg = get_mesh_vertex_features(mesh, feature_provider=None, node_feature_names=["degree"])
                arrays[f"g{g}_degree"] = deg.numpy()
            counts.append(pos.shape[0])
        meta = dict(names=c["names"], add_self_edges=c["self_edges"], node_type="element" if c.get("element") else "vertex",
                    num_graphs=len(c["grids"]), dim=c["dim"], node_counts=counts)
        np.savez_compressed(os.path.join(HERE, f"graph_{name}.npz"), meta=np.array(json.dumps(meta)), **arrays)
        print(name, "ok", counts, arrays["g0_edge_attr"].shape)


class _Tree:
    """pykdtree.kdtree.KDTree stand-in with the one call the reference makes: query(points, k=1) -> (dist, index)."""

    def __init__(self, pts):
        from scipy.spatial import cKDTree

        self.t = cKDTree(pts)

    def query(self, pts, k=1):
        return self.t.query(pts, k=k)


def hierarchical():
    """Inter-level edges (get_inter_graph_edges) between a fine and a coarse mesh, vertex and element flavours; the
    nearest-partner search runs on the host (KD-tree stand-in: scipy's cKDTree for the absent pykdtree; for elements the
    reference's find_closest_elements needs scikit-fem, so the fixture supplies that method from the same tree)."""
    rng = np.random.default_rng(11)
    for name, dim, node_type, names in (("hier_vertex2d", 2, "vertex", ["euclidean_distance", "edge_curvature"]),
                                        ("hier_vertex3d", 3, "vertex", ["distance_vector", "euclidean_distance", "edge_curvature"]),
                                        ("hier_element2d", 2, "element", ["distance_vector", "euclidean_distance"])):
        element = node_type == "element"
        fp, fpos, fnb, fcv = grid_mesh(17, 13, dim, rng)
        cp, cpos, cnb, ccv = grid_mesh(5, 4, dim, rng)
        fine, coarse = Mesh(fp, fpos, fnb, fcv, element=element), Mesh(cp, cpos, cnb, ccv, element=element)
        tree = _Tree(cpos)
        if element:
            coarse.find_closest_elements = lambda pts, _t=tree: _t.query(pts, k=1)[1]
        else:
            coarse.vertex_tree = tree
        inter_ea, inter_ei = get_inter_graph_edges(src_mesh=fine, dest_mesh=coarse, node_type=node_type, edge_feature_names=names)
        fn = get_mesh_element_edges if element else get_mesh_vertex_edges
        f_ea, f_ei = fn(fine, edge_feature_names=names, add_self_edges=True)
        c_ea, c_ei = fn(coarse, edge_feature_names=names, add_self_edges=True)
        meta = dict(names=names, node_type=node_type, dim=dim, n_fine=int(fpos.shape[0]), n_coarse=int(cpos.shape[0]))
        np.savez_compressed(os.path.join(HERE, f"graph_{name}.npz"), meta=np.array(json.dumps(meta)), fine_pairs=fp, fine_pos=fpos,
                            fine_boundary=fnb, fine_curv=fcv, coarse_pairs=cp, coarse_pos=cpos, coarse_boundary=cnb, coarse_curv=ccv,
                            nearest=tree.query(fpos, k=1)[1].astype(np.int64), inter_edge_attr=inter_ea.numpy(),
                            inter_edge_index=inter_ei.numpy(), fine_edge_attr=f_ea.numpy(), fine_edge_index=f_ei.numpy(),
                            coarse_edge_attr=c_ea.numpy(), coarse_edge_index=c_ei.numpy())
        print(name, "ok", inter_ea.shape, inter_ei.shape)


if __name__ == "__main__":
    main()
    hierarchical()
