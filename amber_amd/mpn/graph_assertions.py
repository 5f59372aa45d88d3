"""Opt-in shape checks (mirror of /root/reference/src/mpn/graph_assertions.py:6-48; same messages)."""


class MessagePassingGraphAssertions:
    def __init__(self, *, in_node_features: int, in_edge_features: int):
        self._expected = {"in_node_features": in_node_features, "in_edge_features": in_edge_features}

    def __call__(self, tensor):
        ei, ea, x = tensor.edge_index, tensor.edge_attr, tensor.x
        assert ei.shape[0] == 2, f"Edge index must have shape (2, num_edges), given '{ei.shape}' instead."
        assert ei.shape[1] == ea.shape[0], (
            f"Must provide one edge index per edge feature vector, given '{ei.shape}' and '{ea.shape}' instead."
        )
        want_e = self._expected["in_edge_features"]
        assert ea.shape[1] == want_e, f"Feature dimensions of edges do not match. Given '{ea.shape[1]}', expected '{want_e}"
        want_n = self._expected["in_node_features"]
        assert x.shape[1] == want_n, f"Feature dimensions of nodes do not match. Given '{x.shape[1]}', expected '{want_n}"
