"""Edge and node update modules (mirror of /root/reference/src/mpn/message_passing_modules.py:11-72).

Parameter containers: the edge module's first Linear is [L, 3L] over cat[x[src], x[dst], e] and the node module's
is [L, 2L] over cat[x, scatter_mean(e, dst)].  Both run inside the fused HIP kernels (edge_kernel / node_kernel),
which split the first layer by column blocks instead of materialising the concatenation.
"""
from typing import Dict

import torch

from amber_amd.mpn.common.latent_mlp import LatentMLP


class MessagePassingMetaModule(torch.nn.Module):
    def __init__(self, *, in_features: int, latent_dimension: int, stack_config: Dict):
        super().__init__()
        self._mlp = LatentMLP(in_features=in_features, latent_dimension=latent_dimension, config=stack_config.get("mlp"))
        self.in_features = in_features
        self.latent_dimension = latent_dimension

    def param_list(self):
        return self._mlp.param_list()

    def forward(self, graph):
        raise RuntimeError(f"amber_amd.{type(self).__name__} runs fused inside MessagePassingBlock.forward (HIP); "
                           "it cannot be called on its own.")


class MessagePassingEdgeModule(MessagePassingMetaModule):
    """K2 + K3: MLP_e(cat[x[src], x[dst], e])."""


class MessagePassingNodeModule(MessagePassingMetaModule):
    """K6 + K7: MLP_n(cat[x, scatter_mean(e, dst, N)])."""
