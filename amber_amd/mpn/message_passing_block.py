"""One message passing step (mirror of /root/reference/src/mpn/message_passing_block.py:14-146)."""
from typing import Any, Dict, List, Optional

import torch
from torch import nn as nn

from amber_amd import engine
from amber_amd.csr import CSRCache, csr_of, permute_rows
from amber_amd.mpn.message_passing_modules import MessagePassingEdgeModule, MessagePassingNodeModule


def _mode(value: Optional[str]) -> Optional[str]:
    value = value.lower() if value is not None else None
    return value if value in ("inner", "outer") else None


class MessagePassingBlock(nn.Module):
    """edge update -> (inner residual, inner LN) -> aggregate -> node update -> (inner residual, inner LN) ->
    (outer residual, outer LN); executed by amber_mpn_block (edge_kernel + node_kernel)."""

    def __init__(self, stack_config: Dict[str, Any], latent_dimension: int):
        super().__init__()
        self._latent_dimension = latent_dimension
        self.residual_connections = _mode(stack_config.get("residual_connections"))
        self.layer_norm = _mode(stack_config.get("layer_norm"))
        self.use_layer_norm = self.layer_norm is not None
        self.edge_module = MessagePassingEdgeModule(in_features=3 * latent_dimension, latent_dimension=latent_dimension,
                                                    stack_config=stack_config)
        self.node_module = MessagePassingNodeModule(in_features=2 * latent_dimension, latent_dimension=latent_dimension,
                                                    stack_config=stack_config)
        if self.use_layer_norm:
            self._node_layer_norms = nn.LayerNorm(normalized_shape=latent_dimension)
            self._edge_layer_norms = nn.LayerNorm(normalized_shape=latent_dimension)
        else:
            self._node_layer_norms = None
            self._edge_layer_norms = None
        mlp = stack_config.get("mlp")
        self._mlp_layers = int(mlp.get("num_layers"))
        self._activation = mlp.get("activation_function").lower()
        self._packed = engine.PackedParams()
        self._csr_cache = CSRCache()

    def param_list(self) -> List[torch.Tensor]:
        """edge MLP, node MLP, then edge LN (gamma, beta), node LN (gamma, beta): include/amber_mpn.h order."""
        out = self.edge_module.param_list() + self.node_module.param_list()
        if self.use_layer_norm:
            out += [self._edge_layer_norms.weight, self._edge_layer_norms.bias,
                    self._node_layer_norms.weight, self._node_layer_norms.bias]
        return out

    def forward(self, graph) -> None:
        """In place on a graph whose x / edge_attr are LATENT features (edge_attr in edge_index order)."""
        L = self._latent_dimension
        desc = engine.make_desc(latent=L, num_steps=1, num_step_repeats=1, mlp_layers=self._mlp_layers,
                                activation=self._activation, residual=self.residual_connections, layer_norm=self.layer_norm,
                                in_node_features=0, in_edge_features=0)
        dev = graph.x.device
        zeros = torch.zeros(L, device=dev)
        flat, packed = self._packed.get(desc, [zeros, zeros] + self.param_list(), key_skip=2)
        csr = csr_of(graph, int(graph.x.shape[0]), self._csr_cache)
        x = graph.x.to(torch.float32).clone()
        e = permute_rows(graph.edge_attr, csr.perm, gather=True)
        engine.block(desc, flat, packed, 0, x, e, csr)
        graph.__setattr__("x", x)
        graph.__setattr__("edge_attr", permute_rows(e, csr.perm, gather=False))

    def __repr__(self):
        return f"{self.__class__.__name__}(\n edge_module={self.edge_module},\nnode_module={self.node_module},\n "
