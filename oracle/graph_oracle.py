"""CPU oracle for graph construction (the step before the MPN path).  TEST INFRASTRUCTURE ONLY.

numpy / torch-CPU restatement of get_mesh_vertex_edges / get_mesh_element_edges
(src/mesh_util/transforms/mesh_to_graph.py:58-145), the "degree" vertex feature (:224-225) and the collation of a list
of graphs.  Only `tests/` may import it.

Pinning: tests/golden/make_golden_graph.py runs the reference's own mesh_to_graph functions (unmodified, from
/root/reference) on duck-typed meshes and writes tests/golden/graph_*.npz; tests/test_graph_host.py holds the per-mesh
functions here to those vectors bit for bit.  The COLLATION (torch_geometric.data.Batch.from_data_list, reached through
make_batch, src/helpers/torch_util.py:34-55) lives in the third-party package torch_geometric (pinned 2.6.1 in
env/requirements-pip.txt, not installed, not vendored): `collate` restates its documented behaviour (concatenate in list
order, offset edge_index by the nodes before the graph, batch = graph id, ptr = node prefix sums) -> that link is
"parity unpinned".
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def mesh_edges(pairs: np.ndarray, positions: np.ndarray, num_nodes: int, edge_feature_names: Sequence[str], add_self_edges: bool,
               curvature_per_pair: Optional[np.ndarray] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """get_mesh_vertex_edges (mesh_to_graph.py:97-145) / get_mesh_element_edges (:58-94): (edge_attr, edge_index)."""
    nb = torch.tensor(pairs, dtype=torch.long)
    pos = torch.tensor(positions, dtype=torch.float32)
    src = torch.cat([nb[0], nb[1]], dim=0)
    dst = torch.cat([nb[1], nb[0]], dim=0)
    if add_self_edges:
        src = torch.cat([src, torch.arange(num_nodes)], dim=0)
        dst = torch.cat([dst, torch.arange(num_nodes)], dim=0)
    feats = []
    if "distance_vector" in edge_feature_names:
        feats.extend(list((pos[dst] - pos[src]).T))
    if "euclidean_distance" in edge_feature_names:
        feats.append(torch.norm(pos[dst] - pos[src], dim=1))
    if "edge_curvature" in edge_feature_names and curvature_per_pair is not None:
        c = np.concatenate((curvature_per_pair, curvature_per_pair), axis=0)
        if add_self_edges:
            c = np.concatenate((c, np.zeros(num_nodes)), axis=0)
        feats.append(torch.tensor(c, dtype=torch.float32))
    edge_index = torch.vstack((src, dst)).long()
    edge_attr = torch.stack(feats, dim=1) if feats else torch.zeros(edge_index.shape[1], 0)
    return edge_attr, edge_index


def inter_graph_edges(src_positions: np.ndarray, dest_positions: np.ndarray, nearest: np.ndarray, edge_feature_names: Sequence[str],
                      node_type: str = "vertex") -> Tuple[torch.Tensor, torch.Tensor]:
    """get_inter_graph_edges (mesh_to_graph.py:239-301) after the nearest-partner search (:254-272, host KD-tree /
    element finder: `nearest` is its result): float64 numpy features, cast to float32 at the end."""
    index_offset = src_positions.shape[0]
    src_idx = np.arange(index_offset)
    dest_pos = dest_positions[nearest]
    dest_idx = nearest + index_offset
    positions1 = np.concatenate((src_positions, dest_pos), axis=0)
    positions2 = np.concatenate((dest_pos, src_positions), axis=0)
    indices1 = np.concatenate((src_idx, dest_idx), axis=0)
    indices2 = np.concatenate((dest_idx, src_idx), axis=0)
    feats = []
    if "distance_vector" in edge_feature_names:
        feats.extend(list((positions1 - positions2).T))
    if "euclidean_distance" in edge_feature_names:
        feats.append(np.linalg.norm(positions1 - positions2, axis=1))
    if node_type == "vertex" and "edge_curvature" in edge_feature_names:
        feats.append(np.zeros(len(positions1)))
    edge_index = torch.tensor(np.vstack((indices1, indices2))).long()
    edge_attr = torch.tensor(np.array(feats).T, dtype=torch.float32)
    return edge_attr, edge_index


def hierarchical_graph(graph_ea, graph_ei, n_current, initial_ea, initial_ei, n_initial, inter_ea, inter_ei):
    """AmberData._extend_to_hierarchical_graph, refinement_depth > 0 (src/algorithm/dataloader/amber_data.py:98-115):
    (edge_attr, edge_index, mask_output)."""
    ei = torch.cat([graph_ei, initial_ei + n_current, inter_ei], dim=1)
    ea = torch.cat([graph_ea, initial_ea, inter_ea], dim=0)
    mask = torch.cat([torch.ones(n_current), torch.zeros(n_initial)], dim=0).bool()
    return ea, ei, mask


def curvature_per_pair(num_pairs: int, boundary_edges: np.ndarray, boundary_edge_curvatures: np.ndarray) -> np.ndarray:
    """mesh_to_graph.py:127-129: zeros, boundary pairs overwritten."""
    c = np.zeros(num_pairs)
    c[boundary_edges] = boundary_edge_curvatures
    return c


def degree(pairs: np.ndarray) -> np.ndarray:
    """mesh_to_graph.py:224-225: np.unique(mesh_edges, return_counts=True)[1] (every vertex occurs in a mesh)."""
    return np.unique(pairs, return_counts=True)[1]


def collate(edge_attrs: List[torch.Tensor], edge_indices: List[torch.Tensor], node_counts: Sequence[int]):
    """Batch.from_data_list of torch_geometric (third party, restated): (edge_attr, edge_index, batch, ptr)."""
    ptr = np.concatenate([[0], np.cumsum(node_counts)]).astype(np.int64)
    ei = torch.cat([e + int(o) for e, o in zip(edge_indices, ptr[:-1])], dim=1) if edge_indices else torch.zeros(2, 0, dtype=torch.long)
    ea = torch.cat(edge_attrs, dim=0) if edge_attrs else torch.zeros(0, 0)
    batch = torch.cat([torch.full((int(n),), g, dtype=torch.long) for g, n in enumerate(node_counts)]) if len(node_counts) else \
        torch.zeros(0, dtype=torch.long)
    return ea, ei, batch, torch.from_numpy(ptr)
