"""The step after the model in AMBER's inference step, on the device: mirrors of
`project_elements_to_vertices` and `_mesh_sizing_field_to_num_elements`
(/root/reference/src/mesh_util/sizing_field_util.py:43-70, :128-200) on libamber_mpn.so.

`mesh` is duck-typed: the MeshWrapper attributes the reference reads - `t` [dim + 1, M] vertex ids, `simplex_volumes` [M],
`nvertices` / `num_vertices`, and for vertex sizing fields `mesh_edges` [2, Me] and `vertex_positions` [V, dim].  numpy
arrays are uploaded; tensors already on the GPU are used where they are.  Results come back as the reference returns
them: a float64 numpy array / a Python float.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib


def _dev(a, dtype, device) -> torch.Tensor:
    t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device=device, dtype=dtype).contiguous()


def _nverts(mesh) -> int:
    for name in ("nvertices", "num_vertices"):
        if hasattr(mesh, name):
            return int(getattr(mesh, name))
    return int(mesh.vertex_positions.shape[0])


def project_elements_to_vertices(mesh, element_values, device="cuda", as_tensor: bool = False):
    """Volume-weighted mean of the element values around every vertex; bit-identical to the reference's float64 result
    (same additions in the same order).  Returns a float64 numpy array [num_vertices] (or the device tensor)."""
    lib = _lib.load()
    device = torch.device(device)
    t = _dev(mesh.t, torch.int64, device)
    k, m = int(t.shape[0]), int(t.shape[1])
    v = _nverts(mesh)
    vol = _dev(mesh.simplex_volumes, torch.float64, device)
    val = _dev(element_values, torch.float64, device)
    if val.numel() != m or vol.numel() != m:
        raise ValueError(f"one value and one volume per element expected ({m}), given {val.numel()} and {vol.numel()}")
    out = torch.empty(v, dtype=torch.float64, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    ws = torch.empty(lib.amber_mpn_project_workspace_bytes(k, m, v), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        rc = lib.amber_mpn_project_elements_to_vertices(t.data_ptr(), k, m, v, val.data_ptr(), vol.data_ptr(), out.data_ptr(),
                                                        status.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(device))
    _lib.check(rc, "project_elements_to_vertices")
    if as_tensor:
        return out
    res = out.cpu().numpy()
    if int(status.item()) != 0:
        raise IndexError(f"mesh.t contains vertex ids outside [0, {v})")
    return res


def mesh_sizing_field_to_num_elements(mesh, sizing_field, node_type: Optional[str] = None, gradation: Optional[float] = 1.3,
                                      device="cuda") -> float:
    """Heuristic number of elements of the mesh `sizing_field` would produce (`_mesh_sizing_field_to_num_elements`)."""
    if node_type not in ("vertex", "element"):
        raise ValueError(f"Unsupported interpolation_type: {node_type}")
    lib = _lib.load()
    device = torch.device(device)
    t = _dev(mesh.t, torch.int64, device)
    k, m = int(t.shape[0]), int(t.shape[1])
    dim = int(mesh.dim()) if hasattr(mesh, "dim") else k - 1
    vol = _dev(mesh.simplex_volumes, torch.float64, device)
    sf = _dev(sizing_field, torch.float64, device)
    vertex = node_type == "vertex"
    v = _nverts(mesh) if vertex else 0
    edges = pos = None
    me = 0
    if vertex:
        edges = _dev(mesh.mesh_edges, torch.int64, device)
        pos = _dev(mesh.vertex_positions, torch.float64, device)
        me = int(edges.shape[1])
        if sf.numel() != v:
            raise ValueError(f"one sizing value per vertex expected ({v}), given {sf.numel()}")
    elif sf.numel() != m:
        raise ValueError(f"one sizing value per element expected ({m}), given {sf.numel()}")
    out = torch.empty(2, dtype=torch.float64, device=device)  # [estimate, "an id was out of range" flag]
    ws = torch.empty(lib.amber_mpn_estimate_workspace_bytes(v, me), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        rc = lib.amber_mpn_estimate_num_elements(t.data_ptr(), k, m, v, vol.data_ptr(), sf.data_ptr(), 1 if vertex else 0, dim,
                                                 _lib.ptr(edges), me, _lib.ptr(pos), float(gradation), out.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), _lib.stream_ptr(device))
    _lib.check(rc, "estimate_num_elements")
    res = out.cpu()
    if float(res[1]) != 0.0:
        raise IndexError(f"mesh.t / mesh.mesh_edges contain vertex ids outside [0, {v})")
    return float(res[0])
