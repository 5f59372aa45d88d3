#!/usr/bin/env python3
"""Times the step after the model on a 100k-triangle mesh: the numpy restatement of the reference (oracle, host cores) beside
the device entry points.  Evidence for SURVEY.md section 8f rank 4 ("only if profiling shows it dominates the loop")."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amber_amd import sizing_field as sf  # noqa: E402
from oracle import mesh_oracle as mo  # noqa: E402


class Mesh:
    pass


rng = np.random.default_rng(0)
nx = ny = 225
idx = np.arange(nx * ny).reshape(ny, nx)
a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
t = np.concatenate([np.stack([a, b, d]), np.stack([a, d, c])], axis=1)
pos = np.stack(np.meshgrid(np.arange(nx), np.arange(ny)), -1).reshape(-1, 2).astype(float) + 0.2 * rng.random((nx * ny, 2))
p = pos[t]
vol = 0.5 * np.abs((p[1, :, 0] - p[0, :, 0]) * (p[2, :, 1] - p[0, :, 1]) - (p[2, :, 0] - p[0, :, 0]) * (p[1, :, 1] - p[0, :, 1]))
e = np.unique(np.sort(np.concatenate([t[[0, 1]], t[[1, 2]], t[[2, 0]]], axis=1), axis=0), axis=1)
vals, field = rng.random(t.shape[1]) + 0.5, rng.random(pos.shape[0]) + 0.5
m = Mesh()
m.t, m.vertex_positions, m.simplex_volumes, m.mesh_edges, m.nvertices = t, pos, vol, e, pos.shape[0]
dev = torch.device("cuda:0")
md = Mesh()  # the mesh already on the device (uploaded once per mesh)
md.t, md.vertex_positions = torch.from_numpy(t).to(dev), torch.from_numpy(pos).to(dev)
md.simplex_volumes, md.mesh_edges, md.nvertices = torch.from_numpy(vol).to(dev), torch.from_numpy(e).to(dev), pos.shape[0]
vd, fd = torch.from_numpy(vals).to(dev), torch.from_numpy(field).to(dev)


def tm(f, n=10):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"mesh: {t.shape[1]} triangles, {pos.shape[0]} vertices, {e.shape[1]} edges")
print(f"project_elements_to_vertices   host (numpy restatement) {tm(lambda: mo.project_elements_to_vertices(t, vol, vals, pos.shape[0]), 3):8.2f} ms"
      f"   device, mesh resident {tm(lambda: sf.project_elements_to_vertices(md, vd, device=dev, as_tensor=True)):6.3f} ms"
      f"   device incl. upload + download {tm(lambda: sf.project_elements_to_vertices(m, vals, device=dev)):6.3f} ms")
print(f"num_elements (vertex field)    host {tm(lambda: mo.mesh_sizing_field_to_num_elements(t, vol, field, 'vertex', 2, e, pos), 3):8.2f} ms"
      f"   device, mesh resident {tm(lambda: sf.mesh_sizing_field_to_num_elements(md, fd, node_type='vertex', device=dev)):6.3f} ms")
print(f"num_elements (element field)   host {tm(lambda: mo.mesh_sizing_field_to_num_elements(t, vol, vals, 'element', 2), 3):8.2f} ms"
      f"   device, mesh resident {tm(lambda: sf.mesh_sizing_field_to_num_elements(md, vd, node_type='element', device=dev)):6.3f} ms")
