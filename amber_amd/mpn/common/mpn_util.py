"""Small helpers of the MPN interface (mirror of /root/reference/src/mpn/common/mpn_util.py:9-46,75-85)."""
import copy

import torch


def noop(*args, **kwargs):
    return None


def unpack_features(graph, agent_node_type: str = "node"):
    """Output packing of MessagePassingBase.forward: ({type: x}, {"edges": {edge_index, edge_attr}}, {type: batch}).
    edge_index is cast to int64; a missing `batch` becomes zeros (reference mpn_util.py:22-46)."""
    node_features = graph.x if graph.x is not None else graph.pos
    batch = getattr(graph, "batch", None)
    if batch is None:
        batch = torch.zeros(node_features.shape[0]).long()
    return (
        {agent_node_type: node_features},
        {"edges": {"edge_index": graph.edge_index.long(), "edge_attr": graph.edge_attr}},
        {agent_node_type: batch},
    )


def get_create_copy(create_graph_copy: bool):
    """The reference deep-copies the whole batch per call when create_graph_copy is set (mpn_util.py:82-83).
    The HIP path never writes the caller's tensors, so the guarantee holds without a copy; this helper is kept
    for hosts that call it directly."""
    return (lambda g: copy.deepcopy(g)) if create_graph_copy else (lambda g: g)
