"""K0/K1 input embeddings (mirror of /root/reference/src/mpn/input_embedding.py:9-42)."""
from typing import Dict, List, Optional

import torch
from torch import nn

from amber_amd import engine
from amber_amd.csr import CSRCache, csr_of, permute_rows
from amber_amd.mpn.common.embedding import Embedding


class MessagePassingInputEmbedding(nn.Module):
    def __init__(self, *, in_node_features: int, in_edge_features: int, embedding_config: Optional[Dict], latent_dimension: int,
                 device: Optional = None):
        super().__init__()
        self.node_input_embedding = Embedding(in_features=in_node_features, latent_dimension=latent_dimension,
                                              embedding_config=embedding_config, device=device)
        self.edge_input_embedding = Embedding(in_features=in_edge_features, latent_dimension=latent_dimension,
                                              embedding_config=embedding_config, device=device)
        self._latent_dimension = latent_dimension
        self._packed = engine.PackedParams()
        self._csr_cache = CSRCache()

    def param_list(self) -> List[torch.Tensor]:
        return self.node_input_embedding.param_list() + self.edge_input_embedding.param_list()

    def _desc(self):
        n, e = self.node_input_embedding, self.edge_input_embedding
        emb = n if not n.is_constant_embedding else e
        return engine.make_desc(latent=self._latent_dimension, num_steps=0, num_step_repeats=1, mlp_layers=1, activation="relu",
                                residual=None, layer_norm=None, in_node_features=n.in_features, in_edge_features=e.in_features,
                                embed_layers=emb.embed_layers, embed_activation=emb.embed_activation)

    def forward(self, graph) -> None:
        """In place on `graph` like the reference: x and edge_attr become latent (edge_index order)."""
        desc = self._desc()
        flat, packed = self._packed.get(desc, self.param_list())
        csr = csr_of(graph, int(graph.x.shape[0]), self._csr_cache)
        xl, el = engine.embed(desc, flat, packed, graph.x, graph.edge_attr, csr)
        graph.__setattr__("x", xl)
        graph.__setattr__("edge_attr", permute_rows(el, csr.perm, gather=False))
