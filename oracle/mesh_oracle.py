"""CPU oracle for the step AFTER the model in AMBER's inference step: the element -> vertex
projection of a sizing field and the heuristic element-count estimate.  TEST INFRASTRUCTURE ONLY (tests/, bench legs):
the product path is amber_amd/sizing_field.py on libamber_mpn.so.

numpy restatement of /root/reference/src/mesh_util/sizing_field_util.py; torch_scatter.scatter_add (third party, absent)
is restated as numpy's np.add.at, which like scatter_add_ on the CPU adds the entries in index order.  Pinned against the
reference's own functions by tests/golden/mesh_*.npz (tests/golden/make_golden_mesh.py).
"""
import numpy as np


def edge_length_to_volume(sizing_field: np.ndarray, dim: int) -> np.ndarray:
    """src/tasks/domains/geometry_util.py:30-47."""
    if dim == 2:
        return sizing_field ** 2 * np.sqrt(3) / 4
    if dim == 3:
        return sizing_field ** 3 * np.sqrt(2) / 12
    raise ValueError(f"Dimension {dim} not supported")


def project_elements_to_vertices(t: np.ndarray, volumes: np.ndarray, element_values: np.ndarray, num_vertices: int) -> np.ndarray:
    """sizing_field_util.py:43-70.  t: [dim + 1, M] vertex ids of the simplices (mesh.t); volume-weighted mean of the
    values of the elements around every vertex (float64; a vertex without elements gives 0 / 0 = nan, as there)."""
    k = t.shape[0]
    vol = np.repeat(volumes, k)            # :61
    val = np.repeat(element_values, k)     # :62
    index = t.T.flatten()                  # :63
    sums = np.zeros(num_vertices, dtype=np.float64)
    weights = np.zeros(num_vertices, dtype=np.float64)
    np.add.at(sums, index, val * vol)      # scatter_add :64 (entries added in index order)
    np.add.at(weights, index, vol)         # :65
    with np.errstate(invalid="ignore", divide="ignore"):
        return sums / weights              # :66


def mesh_sizing_field_to_num_elements(t: np.ndarray, volumes: np.ndarray, sizing_field: np.ndarray, node_type: str, dim: int,
                                      edges: np.ndarray = None, positions: np.ndarray = None, gradation: float = 1.3) -> float:
    """sizing_field_util.py:128-200 (_mesh_sizing_field_to_num_elements).  edges: [2, M_e] mesh.mesh_edges, positions:
    [V, dim] (vertex sizing fields only)."""
    if node_type == "vertex":
        pos0, pos1 = positions[edges[0]], positions[edges[1]]
        edge_distances = np.linalg.norm(pos0 - pos1, axis=1)          # :166
        sizing_field = sizing_field.copy()
        for _ in range(2):                                            # :171
            sizing0, sizing1 = sizing_field[edges[0]], sizing_field[edges[1]]
            min_sizing1 = sizing0 * gradation ** (-(edge_distances / sizing0))   # :176-177
            min_sizing0 = sizing1 * gradation ** (-(edge_distances / sizing1))   # :180-181
            np.maximum.at(sizing_field, edges[0], min_sizing0)        # :184
            np.maximum.at(sizing_field, edges[1], min_sizing1)        # :185
        element_sizing_field = sizing_field[t].mean(axis=0)           # :187
        n = (volumes / edge_length_to_volume(element_sizing_field, dim)).sum()   # :188-191
        return n * 0.618 if dim == 3 else n                           # :192-194
    if node_type == "element":
        return (volumes / edge_length_to_volume(sizing_field, dim)).sum()        # :196-199
    raise ValueError(f"Unsupported interpolation_type: {node_type}")
