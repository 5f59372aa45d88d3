import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_mesh
from amber_amd import sizing_field as sf
class M:
    pass
meta, d = load_mesh("tri_24x19")
m = M(); m.t = d["t"]; m.vertex_positions = d["positions"]; m.simplex_volumes = d["volumes"]; m.mesh_edges = d["edges"]; m.nvertices = d["positions"].shape[0]
got = sf.project_elements_to_vertices(m, d["element_values"], device="cuda:0")
want = d["projected"]
bad = np.nonzero(got != want)[0]
print("mismatch", len(bad), "of", len(want), "max rel", np.max(np.abs(got - want) / np.abs(want)))
print(got[bad[:5]], want[bad[:5]])
