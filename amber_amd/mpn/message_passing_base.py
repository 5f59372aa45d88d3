"""The message passing network (mirror of /root/reference/src/mpn/message_passing_base.py:13-146).

forward(graph) keeps the reference contract -- assertions, (no) copy, edge dropout (identity in eval), input
embeddings, the stack, output packing -- but executes K0-K8 as ONE call into libamber_mpn.so
(amber_mpn_forward): destination-sorted CSR built once per mesh, fused edge and node kernels, no float atomics.
"""
from typing import Callable, Dict, List, Optional, Type

import torch
from torch import nn

from amber_amd import engine
from amber_amd.csr import CSRCache, build_csr, csr_of
from amber_amd.mpn.common.mpn_util import get_create_copy, noop, unpack_features
from amber_amd.mpn.graph_assertions import MessagePassingGraphAssertions
from amber_amd.mpn.input_embedding import MessagePassingInputEmbedding
from amber_amd.mpn.message_passing_stack import MessagePassingStack


class _LatentGraph:
    """What the reference hands to unpack_features: the (copied) graph with latent x / edge_attr."""

    def __init__(self, source, x, edge_attr):
        self.__dict__.update({k: v for k, v in getattr(source, "__dict__", {}).items()})
        self.__dict__["_amber_source"] = source
        self.x, self.edge_attr, self.edge_index = x, edge_attr, source.edge_index
        batch = getattr(source, "batch", None)
        if batch is not None:
            self.batch = batch

    def __getattr__(self, name):  # anything else (mask_output, ptr, y, ... of a PyG Batch) is read through
        source = self.__dict__.get("_amber_source")
        if source is None or name.startswith("__"):
            raise AttributeError(name)
        return getattr(source, name)


class MessagePassingBase(nn.Module):
    def __init__(self, *, in_node_features: int, in_edge_features: int, latent_dimension: int, stack_config: Dict,
                 embedding_config: Dict, output_type: str = "features", edge_dropout: float = 0.0,
                 create_graph_copy: bool = True, assert_graph_shapes: bool = True, device: Optional = None,
                 node_type: str = "node"):
        super().__init__()
        self._latent_dimension = latent_dimension
        self._node_type = node_type
        self._in_node_features = in_node_features
        self._in_edge_features = in_edge_features
        if assert_graph_shapes:
            self.maybe_assertions = self._get_assertions()(in_node_features=in_node_features, in_edge_features=in_edge_features)
        else:
            self.maybe_assertions = noop
        if edge_dropout is not None and edge_dropout > 0.0:
            from amber_amd.mpn.common.edge_dropout import EdgeDropout

            self.maybe_edge_dropout = EdgeDropout(dropout_prob=edge_dropout)
        else:
            self.maybe_edge_dropout = noop
        self._create_graph_copy = bool(create_graph_copy)
        self.maybe_create_copy: Callable = get_create_copy(create_graph_copy=create_graph_copy)
        if output_type not in ("features", "graph"):
            raise ValueError(f"Unknown output_type '{output_type}'")
        self._output_type = output_type
        self.input_embeddings = MessagePassingInputEmbedding(
            in_node_features=in_node_features, in_edge_features=in_edge_features, latent_dimension=latent_dimension,
            embedding_config=embedding_config, device=device)
        self.message_passing_stack = MessagePassingStack(stack_config=stack_config, latent_dimension=latent_dimension)
        mlp = stack_config.get("mlp")
        self._stack_kwargs = dict(
            num_steps=self.message_passing_stack.num_steps, num_step_repeats=self.message_passing_stack.num_step_repeats,
            mlp_layers=int(mlp.get("num_layers")), activation=mlp.get("activation_function"),
            residual=_inner_outer(stack_config.get("residual_connections")), layer_norm=_inner_outer(stack_config.get("layer_norm")))
        self._packed = engine.PackedParams()
        self._csr_cache = CSRCache()

    def _get_assertions(self) -> Type[MessagePassingGraphAssertions]:
        return MessagePassingGraphAssertions

    # ---- pieces shared with SupervisedMPN ----------------------------------------------------------
    def desc_kwargs(self) -> dict:
        n, e = self.input_embeddings.node_input_embedding, self.input_embeddings.edge_input_embedding
        emb = n if not n.is_constant_embedding else e
        return dict(latent=self._latent_dimension, in_node_features=self._in_node_features,
                    in_edge_features=self._in_edge_features, embed_layers=emb.embed_layers,
                    embed_activation=emb.embed_activation, **self._stack_kwargs)

    def param_list(self) -> List[torch.Tensor]:
        return self.input_embeddings.param_list() + self.message_passing_stack.param_list()

    def prepare(self, graph):
        """The reference's pre-amble (message_passing_base.py:141-143: assertions, copy, edge dropout) and the CSR.
        Returns (graph to compute on, csr).  Eval: the caller's graph and the per-mesh cached CSR.  Training with edge
        dropout: a private shallow view holding the kept edges (the caller's graph is only touched when
        create_graph_copy=False, as in the reference) and a CSR of its own, since every draw changes the edge list."""
        self.maybe_assertions(graph)
        if self.training and self.maybe_edge_dropout is not noop and self.maybe_edge_dropout.dropout_prob > 0.0:
            work = graph if not self._create_graph_copy else _LatentGraph(graph, graph.x, graph.edge_attr)
            self.maybe_edge_dropout(work)
            return work, build_csr(work.edge_index, int(work.x.shape[0]), validate="deferred")
        return graph, csr_of(graph, int(graph.x.shape[0]), self._csr_cache)

    def transform_to_features(self, graph):
        return unpack_features(graph, agent_node_type=self._node_type)

    def forward(self, graph):
        """graph: PyG `Batch` (or any attribute bag) with x[N,F_n], edge_attr[E,F_e], edge_index[2,E] on the GPU.
        Returns ({node_type: x[N,L]}, {"edges": {edge_index, edge_attr[E,L]}}, {node_type: batch}) for
        output_type="features", None for "graph".  The caller's graph is only mutated when
        create_graph_copy=False (then x / edge_attr become the latent features, as in the reference)."""
        caller_graph = graph
        graph, csr = self.prepare(graph)
        if self._output_type == "graph" and self._create_graph_copy:
            return None  # the reference computes on a private copy and returns noop(graph) == None
        desc = engine.make_desc(**self.desc_kwargs())
        params = self.param_list()
        if engine.wants_grad(params):  # training step: autograd through amber_mpn_forward_train / amber_mpn_backward
            node, edge, _ = engine.MpnFunction.apply(desc, self._packed, graph.x, graph.edge_attr, csr, None, None, True, True,
                                                     None, *params)
        else:
            flat, packed = self._packed.get(desc, params)
            node, edge, _ = engine.forward(desc, flat, packed, graph.x, graph.edge_attr, csr, want_node=True, want_edge=True)
        if not self._create_graph_copy:
            graph = caller_graph
            graph.__setattr__("x", node)
            graph.__setattr__("edge_attr", edge)
        if self._output_type == "graph":
            return None
        return self.transform_to_features(_LatentGraph(graph, node, edge))


def _inner_outer(value: Optional[str]) -> Optional[str]:
    value = value.lower() if value is not None else None
    return value if value in ("inner", "outer") else None
