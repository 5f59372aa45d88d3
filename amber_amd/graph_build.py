"""Graph construction on the GPU: mirror of get_mesh_vertex_edges / get_mesh_element_edges
(/root/reference/src/mesh_util/transforms/mesh_to_graph.py:58-145), the "degree" vertex feature (:224-225) and the
collation the reference gets from torch_geometric's Batch.from_data_list through make_batch
(/root/reference/src/helpers/torch_util.py:34-55), as ONE call of amber_mpn_build_graph for a whole list of meshes.

The mesh itself (connectivity from gmsh / scikit-fem, vertex positions, boundary curvature) is host data and stays the
caller's business: a mesh is anything with the MeshWrapper attributes read here (`mesh_edges`, `vertex_positions`,
`num_vertices`, `boundary_edges`, `boundary_edge_curvatures` for vertex graphs; `element_neighbors`,
`element_midpoints`, `num_elements` for element graphs).  Problem-specific node features (sizing field, FEM solution)
are computed on the host as before and passed through."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .csr import build_csr
from .synthetic import GraphBatch


def _mesh_arrays(mesh, node_type: str, edge_feature_names: Sequence[str]) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray], int]:
    """(pairs [2, M] int64, positions [n, dim] float32, curvature [M] float32 | None, n) of one mesh."""
    if node_type == "vertex":
        pairs, pos, n = np.asarray(mesh.mesh_edges), np.asarray(mesh.vertex_positions), int(mesh.num_vertices)
    elif node_type == "element":
        pairs, pos, n = np.asarray(mesh.element_neighbors), np.asarray(mesh.element_midpoints), int(mesh.num_elements)
    else:
        raise ValueError(f"Node type {node_type=} not supported")
    curv = None
    if node_type == "vertex" and "edge_curvature" in edge_feature_names:
        c = np.zeros(pairs.shape[1])  # float64 like the reference, cast to float32 at the end (mesh_to_graph.py:127-138)
        c[np.asarray(mesh.boundary_edges)] = np.asarray(mesh.boundary_edge_curvatures)
        curv = c.astype(np.float32)
    return pairs.astype(np.int64), pos.astype(np.float32), curv, n


def _build(meshes, edge_feature_names, node_type, add_self_edges, dev, with_degree, extra_edges=0):
    """Intra-mesh part: uploads the meshes' flat arrays once and fills the first 2 M + N columns of edge_index / rows of
    edge_attr, which are allocated with `extra_edges` more columns for a caller that appends inter-level edges."""
    lib = _lib.load()
    if node_type == "element" and "edge_curvature" in edge_feature_names:
        edge_feature_names = [f for f in edge_feature_names if f != "edge_curvature"]  # element graphs have none (:81-90)
    parts = [_mesh_arrays(m, node_type, edge_feature_names) for m in meshes]
    dims = {p[1].shape[1] for p in parts}
    if len(dims) > 1:
        raise ValueError(f"amber_amd: meshes of different dimension in one batch: {sorted(dims)}")
    dim = dims.pop() if dims else 2
    node_ptr = np.zeros(len(parts) + 1, dtype=np.int64)
    edge_ptr = np.zeros(len(parts) + 1, dtype=np.int64)
    np.cumsum([p[3] for p in parts], out=node_ptr[1:])
    np.cumsum([p[0].shape[1] for p in parts], out=edge_ptr[1:])
    n, m = int(node_ptr[-1]), int(edge_ptr[-1])
    e_intra = 2 * m + (n if add_self_edges else 0)
    e = e_intra + extra_edges
    gdesc = _lib.AmberGraphDesc(dim=dim, add_self_edges=int(bool(add_self_edges)),
                                distance_vector=int("distance_vector" in edge_feature_names),
                                euclidean_distance=int("euclidean_distance" in edge_feature_names),
                                edge_curvature=int("edge_curvature" in edge_feature_names),
                                edge_capacity=e if extra_edges else 0, edge_offset=0)
    f = lib.amber_mpn_graph_edge_features(C.byref(gdesc))
    if f < 0:
        raise ValueError(f"amber_amd: positions must have 2 or 3 columns, given {dim}")

    def up(arrs, dtype, width=None):  # host staging: ONE upload per array for the whole batch
        cat = np.concatenate(arrs, axis=0) if arrs else np.zeros((0,) if width is None else (0, width), dtype=dtype)
        return torch.from_numpy(np.ascontiguousarray(cat, dtype=dtype)).to(dev)

    nbr_a, nbr_b = up([p[0][0] for p in parts], np.int64), up([p[0][1] for p in parts], np.int64)
    pos = up([p[1] for p in parts], np.float32, dim)
    curv = up([p[2] for p in parts], np.float32) if gdesc.edge_curvature else None
    nptr, eptr = torch.from_numpy(node_ptr).to(dev), torch.from_numpy(edge_ptr).to(dev)
    edge_index = torch.empty(2, e, dtype=torch.int64, device=dev)
    edge_attr = torch.empty(e, f, dtype=torch.float32, device=dev)
    batch = torch.empty(n, dtype=torch.int64, device=dev)
    degree = torch.empty(n, dtype=torch.float32, device=dev) if with_degree else None
    status = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.amber_mpn_build_graph(C.byref(gdesc), len(parts), nptr.data_ptr(), eptr.data_ptr(), nbr_a.data_ptr(), nbr_b.data_ptr(),
                                       pos.data_ptr(), _lib.ptr(curv), n, m, edge_index.data_ptr(), edge_attr.data_ptr(),
                                       batch.data_ptr(), _lib.ptr(degree), status.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "build_graph")
    return dict(gdesc=gdesc, edge_index=edge_index, edge_attr=edge_attr, batch=batch, degree=degree, status=status, node_ptr=nptr,
                node_counts=[p[3] for p in parts], e_intra=e_intra, n=n)


def _device(device) -> torch.device:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("amber_amd: graph construction runs on the GPU; there is no CPU path")
    return dev


def build_graph_batch(meshes: Sequence, edge_feature_names: Sequence[str], node_type: str = "vertex", add_self_edges: bool = True,
                      device="cuda", with_degree: bool = False, with_csr: bool = True, validate: bool = True) -> GraphBatch:
    """Batch of observation graphs for `meshes`, built on `device`: edge_index [2, E] (int64), edge_attr [E, F], batch,
    ptr, optionally `degree` [N] (float32) and the destination-sorted CSR (`amber_csr`, so the first forward skips the
    sort).  Node features `x` are the caller's to set."""
    dev = _device(device)
    r = _build(meshes, edge_feature_names, node_type, add_self_edges, dev, with_degree)
    if validate and int(r["status"].item()) != 0:
        raise IndexError("amber_amd: a mesh edge refers to a node outside its mesh")
    out = GraphBatch(edge_index=r["edge_index"], edge_attr=r["edge_attr"], batch=r["batch"], ptr=r["node_ptr"], num_graphs=len(meshes))
    if r["degree"] is not None:
        out.degree = r["degree"]
    if with_csr:
        out.amber_csr = build_csr(r["edge_index"], r["n"], validate=False)
    return out


def build_hierarchical_graph(mesh, initial_mesh, nearest, edge_feature_names: Sequence[str], node_type: str = "vertex",
                             add_self_edges: bool = True, device="cuda", x: Optional[torch.Tensor] = None,
                             initial_x: Optional[torch.Tensor] = None, topology_only: bool = False, with_csr: bool = True,
                             validate: bool = True) -> GraphBatch:
    """AMBER's hierarchical observation graph (AmberData._extend_to_hierarchical_graph,
    /root/reference/src/algorithm/dataloader/amber_data.py:68-115, refinement_depth > 0) on the device:

        edge_index = [ current mesh | initial mesh + n_current | inter-level edges ]   (same order for edge_attr)
        mask_output = [ True x n_current | False x n_initial ]
        x           = [ x | 0 ; initial_x (zeros if topology_only) | 1 ]               (when x / initial_x are given)

    `nearest[k]`: the partner of current node k in the initial mesh - `initial_mesh.vertex_tree.query(positions, k=1)[1]`
    for vertex graphs, `initial_mesh.find_closest_elements(midpoints)` for element graphs (mesh_to_graph.py:254-272):
    that search stays on the host.  The inter-level features are computed in float64 from the meshes' float64 positions,
    as the reference's numpy code does."""
    lib = _lib.load()
    dev = _device(device)
    nearest = np.ascontiguousarray(np.asarray(nearest), dtype=np.int64)
    pos_attr = "vertex_positions" if node_type == "vertex" else "element_midpoints"
    if node_type not in ("vertex", "element"):
        raise ValueError(f"Node type {node_type=} not supported")
    ps = np.ascontiguousarray(np.asarray(getattr(mesh, pos_attr)), dtype=np.float64)
    pd = np.ascontiguousarray(np.asarray(getattr(initial_mesh, pos_attr)), dtype=np.float64)
    n_src, n_dst = ps.shape[0], pd.shape[0]
    if nearest.shape != (n_src,):
        raise ValueError(f"amber_amd: one partner per current node expected, given {nearest.shape} for {n_src} nodes")
    r = _build([mesh, initial_mesh], edge_feature_names, node_type, add_self_edges, dev, False, extra_edges=2 * n_src)
    g = r["gdesc"]
    g.edge_offset = r["e_intra"]
    near_d, ps_d, pd_d = torch.from_numpy(nearest).to(dev), torch.from_numpy(ps).to(dev), torch.from_numpy(pd).to(dev)
    status2 = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.amber_mpn_build_inter_edges(C.byref(g), n_src, n_dst, n_src, near_d.data_ptr(), ps_d.data_ptr(), pd_d.data_ptr(),
                                             r["edge_index"].data_ptr(), r["edge_attr"].data_ptr(), status2.data_ptr(),
                                             _lib.stream_ptr(dev))
    _lib.check(rc, "build_inter_edges")
    if validate and (int(r["status"].item()) != 0 or int(status2.item()) != 0):
        raise IndexError("amber_amd: a mesh edge or inter-level partner refers to a node outside its mesh")
    mask = torch.zeros(n_src + n_dst, dtype=torch.bool, device=dev)
    mask[:n_src] = True
    out = GraphBatch(edge_index=r["edge_index"], edge_attr=r["edge_attr"], mask_output=mask,
                     batch=torch.zeros(n_src + n_dst, dtype=torch.int64, device=dev))
    if x is not None and initial_x is not None:
        xi = torch.zeros_like(initial_x) if topology_only else initial_x
        out.x = torch.cat([torch.cat([x.to(dev), torch.zeros(n_src, 1, device=dev, dtype=x.dtype)], dim=1),
                           torch.cat([xi.to(dev), torch.ones(n_dst, 1, device=dev, dtype=x.dtype)], dim=1)], dim=0)
    if with_csr:
        out.amber_csr = build_csr(r["edge_index"], n_src + n_dst, validate=False)
    return out


def _single(mesh, edge_feature_names, node_type, add_self_edges, device):
    g = build_graph_batch([mesh], edge_feature_names, node_type=node_type, add_self_edges=add_self_edges, device=device, with_csr=False)
    if g.edge_attr.shape[1] == 0:
        raise RuntimeError("stack expects a non-empty TensorList")  # torch.stack([]) in the reference (mesh_to_graph.py:92,143)
    return g.edge_attr, g.edge_index


def get_mesh_vertex_edges(wrapped_mesh, edge_feature_names: List[str], add_self_edges: bool = True, device="cuda"):
    """(edge_attr [E, F], edge_index [2, E]) of one mesh's vertex graph, on the GPU (mesh_to_graph.py:97-145)."""
    return _single(wrapped_mesh, edge_feature_names, "vertex", add_self_edges, device)


def get_mesh_element_edges(wrapped_mesh, edge_feature_names: List[str], add_self_edges: bool = True, device="cuda"):
    """(edge_attr, edge_index) of one mesh's element graph (mesh_to_graph.py:58-94)."""
    return _single(wrapped_mesh, edge_feature_names, "element", add_self_edges, device)
