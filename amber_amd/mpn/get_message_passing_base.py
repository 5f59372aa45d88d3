"""Factories (mirror of /root/reference/src/mpn/get_message_passing_base.py:8-68)."""
from typing import Any, Dict, Optional, Tuple, Union

from amber_amd.mpn.message_passing_base import MessagePassingBase


def get_mpn_from_graph(*, example_graph, latent_dimension: int, base_config: Dict[str, Any], node_name: str = "node",
                       device: Optional = None) -> MessagePassingBase:
    """Feature widths are read off an example graph (x.shape[1], edge_attr.shape[1])."""
    return get_mpn(in_node_features=example_graph.x.shape[1], in_edge_features=example_graph.edge_attr.shape[1],
                   latent_dimension=latent_dimension, base_config=base_config, node_name=node_name, device=device)


def get_mpn(*, in_node_features: Union[int, Dict[str, int]], in_edge_features: Union[int, Dict[Tuple[str, str, str], int]],
            latent_dimension: int, base_config: Dict[str, Any], node_name: str = "node",
            device: Optional = None) -> MessagePassingBase:
    """Reads create_graph_copy, assert_graph_shapes, stack, embedding, edge_dropout from `base_config` (only `.get`
    is used, so a dict or a DictConfig works) and moves the module to `device`."""
    assert type(in_node_features) == type(in_edge_features), (
        f"May either provide feature dimensions as int or Dict, but not both. Given '{in_node_features}', '{in_edge_features}'"
    )
    if not isinstance(in_node_features, int):
        raise NotImplementedError("amber_amd: heterogeneous (dict-typed) feature dimensions are not part of AMBER's MPN path")
    edge_dropout = base_config.get("edge_dropout")
    base = MessagePassingBase(
        in_node_features=in_node_features, in_edge_features=in_edge_features, latent_dimension=latent_dimension,
        stack_config=base_config.get("stack"), embedding_config=base_config.get("embedding"),
        edge_dropout=0.0 if edge_dropout is None else edge_dropout, create_graph_copy=base_config.get("create_graph_copy"),
        assert_graph_shapes=base_config.get("assert_graph_shapes"), node_type=node_name)
    return base.to(device)
