"""Synthetic mesh-graph batches shaped like the graphs AMBER feeds its MPN.

The real graphs come from gmsh meshes through `mesh_to_graph`
(/root/reference/src/mesh_util/transforms/mesh_to_graph.py:97-145), which is host-side code outside the
hot path.  gmsh is not available, so benchmarks and tests use structured triangulations built with
the same edge-ordering rule: for undirected mesh edges (a, b) of a graph with n vertices the directed
list is `src = [a, b, arange(n)]`, `dst = [b, a, arange(n)]` (both directions, then self edges,
mesh_to_graph.py:111-117), and graphs are concatenated with node-id offsets exactly as PyG's
`Batch.from_data_list` does.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


class GraphBatch:
    """Duck-typed stand-in for `torch_geometric.data.Batch`: a plain attribute bag.

    The MPN only reads `x, edge_attr, edge_index, batch, mask_output` (and `ptr` for sharding); any other
    attribute passes through untouched.  A real PyG Batch works in its place.
    """

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to(self, device):
        out = GraphBatch()
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device) if torch.is_tensor(v) else v)
        return out

    def keys(self):
        return list(self.__dict__.keys())


def grid_triangulation_edges(nx: int, ny: int) -> np.ndarray:
    """Undirected edges [2, M] of a structured triangulation of an nx x ny vertex grid:
    horizontal, vertical and one diagonal per cell.  M = (nx-1)ny + nx(ny-1) + (nx-1)(ny-1)."""
    idx = np.arange(nx * ny, dtype=np.int64).reshape(ny, nx)
    horiz = np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()])
    vert = np.stack([idx[:-1, :].ravel(), idx[1:, :].ravel()])
    diag = np.stack([idx[:-1, :-1].ravel(), idx[1:, 1:].ravel()])
    return np.concatenate([horiz, vert, diag], axis=1)


def directed_mesh_edges(undirected: np.ndarray, num_nodes: int, add_self_edges: bool = True) -> np.ndarray:
    """Directed edge_index [2, E] in the reference's order (mesh_to_graph.py:111-117)."""
    a, b = undirected[0], undirected[1]
    src = np.concatenate([a, b])
    dst = np.concatenate([b, a])
    if add_self_edges:
        ar = np.arange(num_nodes, dtype=np.int64)
        src = np.concatenate([src, ar])
        dst = np.concatenate([dst, ar])
    return np.stack([src, dst])


def grid3d_edges(nx: int, ny: int, nz: int) -> np.ndarray:
    """Undirected edges of a '3D-like' vertex graph (tetrahedralised brick): axis edges, the three face
    diagonals and the body diagonal per cell -> interior vertex degree 14 (+ self = 15)."""
    idx = np.arange(nx * ny * nz, dtype=np.int64).reshape(nz, ny, nx)
    out = []
    for dz, dy, dx in [(0, 0, 1), (0, 1, 0), (1, 0, 0), (0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1)]:
        a = idx[: nz - dz, : ny - dy, : nx - dx].ravel()
        b = idx[dz:, dy:, dx:].ravel()
        out.append(np.stack([a, b]))
    return np.concatenate(out, axis=1)


def make_batch(
    edge_lists: Sequence[np.ndarray],
    node_counts: Sequence[int],
    in_node_features: int,
    in_edge_features: int,
    seed: int = 1234,
    clip: float = 20.0,
    mask_outputs: Optional[Sequence[np.ndarray]] = None,
) -> GraphBatch:
    """Collate directed per-graph edge lists into one disjoint batch (PyG `Batch.from_data_list`
    semantics: node ids offset per graph, `batch` vector, `ptr` prefix).  Features ~ N(0,1) clipped to
    +-clip (the reference normaliser clips to +-20, config default_normalizer.yaml:3)."""
    ptr = np.zeros(len(node_counts) + 1, dtype=np.int64)
    np.cumsum(np.asarray(node_counts, dtype=np.int64), out=ptr[1:])
    edge_index = np.concatenate([el + ptr[g] for g, el in enumerate(edge_lists)], axis=1)
    edge_ptr = np.zeros(len(node_counts) + 1, dtype=np.int64)
    np.cumsum(np.asarray([el.shape[1] for el in edge_lists], dtype=np.int64), out=edge_ptr[1:])
    n, e = int(ptr[-1]), edge_index.shape[1]
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(n, in_node_features, generator=gen).clamp_(-clip, clip)
    edge_attr = torch.randn(e, in_edge_features, generator=gen).clamp_(-clip, clip)
    batch = torch.from_numpy(np.repeat(np.arange(len(node_counts), dtype=np.int64), node_counts))
    if mask_outputs is None:
        mask = torch.ones(n, dtype=torch.bool)
    else:
        mask = torch.from_numpy(np.concatenate(mask_outputs).astype(bool))
    return GraphBatch(
        x=x,
        edge_attr=edge_attr,
        edge_index=torch.from_numpy(edge_index),
        batch=batch,
        ptr=torch.from_numpy(ptr),
        edge_ptr=torch.from_numpy(edge_ptr),
        mask_output=mask,
    )


def grid_batch(
    num_graphs: int = 100,
    nx: int = 120,
    ny: int = 120,
    in_node_features: int = 4,
    in_edge_features: int = 2,
    seed: int = 1234,
    hetero: bool = False,
) -> GraphBatch:
    """The BASELINE.json config-5 batch: `num_graphs` disjoint triangulated grids.  Defaults give
    N = 1,440,000 and E = 9,984,200 (99,842 directed edges per 120x120 graph incl. self edges).
    `hetero=True` draws nx, ny ~ U{80..160} per graph from numpy's default_rng(seed)."""
    rng = np.random.default_rng(seed)
    edge_lists, counts = [], []
    cache = {}
    for _ in range(num_graphs):
        gx, gy = (int(rng.integers(80, 161)), int(rng.integers(80, 161))) if hetero else (nx, ny)
        if (gx, gy) not in cache:
            cache[(gx, gy)] = directed_mesh_edges(grid_triangulation_edges(gx, gy), gx * gy)
        edge_lists.append(cache[(gx, gy)])
        counts.append(gx * gy)
    return make_batch(edge_lists, counts, in_node_features, in_edge_features, seed=seed)


def brick_batch(num_graphs: int, nx: int, ny: int, nz: int, in_node_features: int = 4, in_edge_features: int = 2,
                seed: int = 1234) -> GraphBatch:
    """'3D-like' variant (in-degree ~15), standing in for the beam/console/mold tetrahedral vertex graphs."""
    el = directed_mesh_edges(grid3d_edges(nx, ny, nz), nx * ny * nz)
    return make_batch([el] * num_graphs, [nx * ny * nz] * num_graphs, in_node_features, in_edge_features, seed=seed)


def hierarchical_graph(nx: int, ny: int, cx: int, cy: int, rng: np.random.Generator) -> Tuple[np.ndarray, int, np.ndarray]:
    """A fine grid mesh plus a coarse 'initial mesh' and inter-mesh edges, shaped like AmberData's
    hierarchical observation graph (/root/reference/src/algorithm/dataloader/amber_data.py:98-122):
    edge list = [fine edges, coarse edges + n_fine, inter edges]; every fine vertex links to one coarse
    vertex (both directions), so coarse vertices are hub nodes; mask_output drops the coarse nodes."""
    n_f, n_c = nx * ny, cx * cy
    fine = directed_mesh_edges(grid_triangulation_edges(nx, ny), n_f)
    coarse = directed_mesh_edges(grid_triangulation_edges(cx, cy), n_c) + n_f
    owner = rng.integers(0, n_c, size=n_f) + n_f
    ar = np.arange(n_f, dtype=np.int64)
    inter = np.stack([np.concatenate([ar, owner]), np.concatenate([owner, ar])])
    el = np.concatenate([fine, coarse, inter], axis=1)
    mask = np.concatenate([np.ones(n_f, dtype=bool), np.zeros(n_c, dtype=bool)])
    return el, n_f + n_c, mask


def random_graph(n: int, e: int, rng: np.random.Generator, isolated: int = 0) -> np.ndarray:
    """Random directed multigraph; the last `isolated` nodes receive no incoming edge
    (zero in-degree rows of scatter_mean)."""
    src = rng.integers(0, n, size=e)
    dst = rng.integers(0, max(n - isolated, 1), size=e)
    return np.stack([src, dst]).astype(np.int64)
