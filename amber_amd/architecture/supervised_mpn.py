"""The model AMBER calls (mirror of /root/reference/src/algorithm/architecture/supervised_mpn.py:10-56):
MPN -> x[mask_output] -> decoder MLP -> readout Linear(L, 1), as one call into libamber_mpn.so."""
import torch
import torch.nn as nn

from amber_amd import engine
from amber_amd.architecture.mlp import MLP
from amber_amd.mpn.get_message_passing_base import get_mpn_from_graph


class SupervisedMPN(nn.Module):
    def __init__(self, architecture_config, example_graph):
        """architecture_config: `.latent_dimension`, `.decoder` and `.get(...)` for create_graph_copy,
        assert_graph_shapes, stack, embedding, edge_dropout (an omegaconf DictConfig in the reference; any object
        with attribute access and `.get` works).  example_graph fixes the raw feature widths."""
        super().__init__()
        self._node_type = "node"
        latent_dimension = architecture_config.latent_dimension
        self.mpn = get_mpn_from_graph(example_graph=example_graph, latent_dimension=latent_dimension, node_name=self._node_type,
                                      base_config=architecture_config)
        mlp_config = architecture_config.decoder
        self.decoder_mlp = MLP(in_features=latent_dimension, mlp_config=mlp_config, latent_dimension=latent_dimension)
        self.readout = nn.Linear(latent_dimension, 1)
        self._packed = engine.PackedParams()

    def forward(self, observations, **kwargs) -> torch.Tensor:
        """observations: (batch of) observation graph(s) on the GPU.  Returns one scalar per kept node,
        shape [N_masked, 1] (all nodes when the batch has no `mask_output`)."""
        observations, csr = self.mpn.prepare(observations)
        n = csr.num_nodes
        out_index, n_out = None, n
        if hasattr(observations, "mask_output"):
            out_index, n_out = engine.out_index_from_mask(observations.mask_output)
        dec = self.decoder_mlp
        desc = engine.make_desc(has_head=True, decoder_layers=dec.num_layers, decoder_activation=dec.activation_function,
                                decoder_norm=dec.latent_normalization, **self.mpn.desc_kwargs())
        dec_pieces, dec_sources = dec.param_list(single_row=(n_out == 1))
        base = self.mpn.param_list()
        tail = [self.readout.weight, self.readout.bias]
        if engine.wants_grad(base + dec_sources + tail):  # training step (src/algorithm/core/amber.py:62-82)
            if self.training and any(isinstance(m, nn.Dropout) and m.p > 0 for m in dec.mlp_layers):
                raise NotImplementedError("amber_amd: decoder dropout is not built for training")
            if dec.latent_normalization is not None:
                raise NotImplementedError("amber_amd: decoder latent_normalization is not built for training")
            return engine.MpnFunction.apply(desc, self._packed, observations.x, observations.edge_attr, csr, out_index, n_out,
                                            False, False, *(base + dec_pieces + tail))[2]
        flat, packed = self._packed.get(desc, base + dec_pieces + tail, key_tensors=base + dec_sources + tail
                                        + ([self.readout.bias] if n_out == 1 else []))
        _, _, out = engine.forward(desc, flat, packed, observations.x, observations.edge_attr, csr, want_node=False,
                                   want_edge=False, out_index=out_index, n_out=n_out)
        return out
