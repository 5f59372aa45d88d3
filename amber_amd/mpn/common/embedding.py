"""Input embedding container (mirror of /root/reference/src/mpn/common/embedding.py:10-61)."""
from typing import Dict, List, Optional

import torch
from torch import nn

from amber_amd.mpn.common.empty_linear_layer import EmptyLinear
from amber_amd.mpn.common.latent_mlp import LatentMLP


class Embedding(nn.Module):
    """`embedding` is an EmptyLinear (no raw features), an nn.Linear (no embedding config) or a LatentMLP; the
    attribute name matches the reference so checkpoints load.  Executed by embed_kernel (K0/K1)."""

    def __init__(self, in_features: int, latent_dimension: int, embedding_config: Optional[Dict], device=None):
        super().__init__()
        if in_features == 0:
            self.embedding = EmptyLinear(out_features=latent_dimension, device=device)
            self.is_constant_embedding = True
            self.embed_layers = 0
            self.embed_activation = "leakyrelu"
        else:
            if embedding_config is None:
                self.embedding = nn.Linear(in_features=in_features, out_features=latent_dimension, device=device)
                self.embed_layers = 0
                self.embed_activation = "leakyrelu"
            else:
                mlp_config = embedding_config.get("mlp")
                self.embedding = LatentMLP(in_features=in_features, latent_dimension=latent_dimension, config=mlp_config)
                self.embed_layers = self.embedding.num_layers
                self.embed_activation = self.embedding.activation_function
            self.is_constant_embedding = False
        self._out_features = latent_dimension
        self.in_features = in_features

    def param_list(self) -> List[torch.Tensor]:
        if self.is_constant_embedding:
            return [self.embedding.bias]
        if isinstance(self.embedding, LatentMLP):
            return self.embedding.param_list()
        return [self.embedding.weight, self.embedding.bias]

    @property
    def out_features(self) -> int:
        return self._out_features

    def forward(self, tensor: torch.Tensor) -> torch.Tensor:
        raise RuntimeError("amber_amd.Embedding holds parameters only; it runs inside MessagePassingInputEmbedding.forward.")

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, is_constant={self.is_constant_embedding}"
