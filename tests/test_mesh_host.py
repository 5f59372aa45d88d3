"""CPU: the oracle of the step after the model (oracle/mesh_oracle.py) against the fixtures the reference's own
project_elements_to_vertices / _mesh_sizing_field_to_num_elements produced (tests/golden/make_golden_mesh.py)."""
import numpy as np
import pytest

from conftest import load_mesh, mesh_cases
from oracle import mesh_oracle as mo


def test_fixtures_exist():
    assert len(mesh_cases()) >= 3


@pytest.mark.parametrize("name", mesh_cases())
def test_oracle_matches_reference_fixture(name):
    meta, d = load_mesh(name)
    v = d["positions"].shape[0]
    got = mo.project_elements_to_vertices(d["t"], d["volumes"], d["element_values"], v)
    assert np.array_equal(got, d["projected"], equal_nan=True)  # same additions in the same order: bit for bit
    if name.startswith("tri_iso"):
        assert np.isnan(d["projected"]).sum() == 3  # vertices no element uses: 0 / 0, as in the reference
    nv = mo.mesh_sizing_field_to_num_elements(d["t"], d["volumes"], d["vertex_field"], "vertex", meta["dim"], d["edges"],
                                              d["positions"], meta["gradation"])
    ne = mo.mesh_sizing_field_to_num_elements(d["t"], d["volumes"], d["element_values"], "element", meta["dim"])
    assert abs(nv - float(d["n_vertex"])) <= 1e-13 * abs(float(d["n_vertex"]))
    assert abs(ne - float(d["n_element"])) <= 1e-13 * abs(float(d["n_element"]))
    with pytest.raises(ValueError, match="Unsupported interpolation_type"):
        mo.mesh_sizing_field_to_num_elements(d["t"], d["volumes"], d["element_values"], "pixel", meta["dim"])
