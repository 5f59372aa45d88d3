"""The model AMBER calls (mirror of /root/reference/src/algorithm/architecture/supervised_mpn.py:10-56):
MPN -> x[mask_output] -> decoder MLP -> readout Linear(L, 1), as one call into libamber_mpn.so."""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from amber_amd import _lib, engine
from amber_amd.architecture.mlp import MLP
from amber_amd.mpn.get_message_passing_base import get_mpn_from_graph


# Every registration of a Parameter on ANY module (construction, `module.weight = nn.Parameter(...)`) bumps this counter;
# a cached plan (the module walk's tensor lists) is only valid for the generation it was made in.  One integer compare per
# forward instead of a walk over ~250 parameters.
_param_generation = [0]


def _note_parameter_registration(module, name, param):
    _param_generation[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_note_parameter_registration)


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
        self._plans = {}          # per call shape: descriptor + the module walk's tensor lists (see _plan)
        self._mask_cache = None   # (weakref(mask), version, out_index, n_out)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._plans.clear())

    # ---- host-side caches: one mesh per forward (AMBER's refinement loop) is 42 launches of ~30 us; walking ~250
    # parameter tensors through nn.Module.__getattr__ (0.4 ms) and a device->host read of the mask count per call would
    # show up next to that ----
    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda() / .float(): parameters are re-created
        self._plans.clear()
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode: bool = True):
        self._plans.clear()
        return super().train(mode)

    def _plan(self, single_row: bool):
        """(desc, flat-buffer pieces, key tensors).  The lists are cached while the decoder's pieces ARE its parameters
        (no BatchNorm folding, no spectral norm: those derive temporaries per call); `_apply`, `train()` and
        `load_state_dict` drop the cache, and so does the registration of a new Parameter object on any module
        (`model.readout.weight = nn.Parameter(...)`: `_param_generation`).  PackedParams still checks every tensor's data
        pointer and version per call."""
        if self._plans.get("generation") != _param_generation[0]:
            self._plans.clear()
            self._plans["generation"] = _param_generation[0]
        key = (single_row, engine.default_precision)
        plan = self._plans.get(key)
        if plan is not None:
            return plan
        dec = self.decoder_mlp
        desc = engine.make_desc(has_head=True, decoder_layers=dec.num_layers, decoder_activation=dec.activation_function,
                                decoder_norm=dec.latent_normalization, **self.mpn.desc_kwargs())
        dec_pieces, dec_sources = dec.param_list(single_row=single_row)
        base = self.mpn.param_list()
        tail = [self.readout.weight, self.readout.bias]
        plan = (desc, base, dec_pieces, dec_sources, tail)
        if len(dec_pieces) == len(dec_sources) and all(a is b for a, b in zip(dec_pieces, dec_sources)):
            self._plans[key] = plan
        return plan

    def _out_index(self, observations, n):
        """x[mask_output] as (out_index, n_out); remembered per mask tensor (identity + version), so that a repeated
        forward on one graph does not read the kept-node count back from the device each time."""
        if not hasattr(observations, "mask_output"):
            return None, n
        mask = observations.mask_output
        hit = self._mask_cache
        if hit is not None and hit[0]() is mask and hit[1] == mask._version:
            return hit[2], hit[3]
        out_index, n_out = engine.out_index_from_mask(mask)
        self._mask_cache = (weakref.ref(mask), mask._version, out_index, n_out)
        return out_index, n_out

    def _decoder_runs_as_modules(self) -> bool:
        """In train() mode the reference's decoder applies dropout and BatchNorm batch statistics (and updates the running
        buffers) whatever the autograd mode (mlp.py:128-129); the fused head implements the eval-mode forms only.  For
        those configurations - off AMBER's defaults (config/algorithm/architecture/decoder/mlp.yaml: dropout 0, no latent
        normalisation) - the message-passing stack runs through the HIP kernels and the decoder's own torch modules are
        applied to its latents (N_masked x L rows: torch's dropout generator, batch statistics and buffer updates are then
        the reference's by construction).  The same for BatchNorm in eval mode with gradients asked for (its backward is not
        in the C library)."""
        mods = list(self.decoder_mlp.mlp_layers)
        has_bn = any(isinstance(m, nn.BatchNorm1d) for m in mods)
        if self.training:
            return has_bn or any(isinstance(m, nn.Dropout) and m.p > 0 for m in mods)
        return has_bn and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def _forward_module_decoder(self, observations) -> torch.Tensor:
        """supervised_mpn.py:48-54 with the stack on the HIP path (autograd through amber_mpn_backward for the latents)."""
        node_features, _, _ = self.mpn(observations)
        node_features = node_features.get(self._node_type)
        if hasattr(observations, "mask_output"):
            node_features = node_features[observations.mask_output]
        for layer in self.decoder_mlp.mlp_layers:
            node_features = layer(node_features)
        return self.readout(node_features)

    def _check_train_mode_decoder(self):
        """fused_predict is an inference call: in train() mode a decoder with dropout or BatchNorm would silently get
        eval-mode numbers there, so it raises (forward() handles those configurations, see _decoder_runs_as_modules)."""
        if not self.training:
            return
        for m in self.decoder_mlp.mlp_layers:
            if isinstance(m, nn.Dropout) and m.p > 0:
                raise NotImplementedError("amber_amd: decoder dropout is not built for train() mode (call model.eval() for inference)")
            if isinstance(m, nn.BatchNorm1d):
                raise NotImplementedError("amber_amd: decoder BatchNorm (batch statistics, running-buffer updates) is not built "
                                          "for train() mode (call model.eval() for inference)")

    def forward(self, observations, **kwargs) -> torch.Tensor:
        """observations: (batch of) observation graph(s) on the GPU.  Returns one scalar per kept node,
        shape [N_masked, 1] (all nodes when the batch has no `mask_output`)."""
        if self._decoder_runs_as_modules():
            return self._forward_module_decoder(observations)
        observations, csr = self.mpn.prepare(observations)
        n = csr.num_nodes
        out_index, n_out = self._out_index(observations, n)
        dec = self.decoder_mlp
        desc, base, dec_pieces, dec_sources, tail = self._plan(single_row=(n_out == 1))
        if engine.wants_grad(base + dec_sources + tail):  # training step (src/algorithm/core/amber.py:62-82)
            return engine.MpnFunction.apply(desc, self._packed, observations.x, observations.edge_attr, csr, out_index, n_out,
                                            False, False, base + dec_sources + tail, *(base + dec_pieces + tail))[2]
        flat, packed = self._packed.get(desc, base + dec_pieces + tail, key_tensors=base + dec_sources + tail
                                        + ([self.readout.bias] if n_out == 1 else []))
        _, _, out = engine.forward(desc, flat, packed, observations.x, observations.edge_attr, csr, want_node=False,
                                   want_edge=False, out_index=out_index, n_out=n_out)
        return out

    def fused_predict(self, observations, glue_desc, node_normalizer, edge_normalizer, prediction_normalizer, baseline) -> torch.Tensor:
        """amber_mpn_predict: input normalisation + forward + post-processing of the readout in one C call (inference;
        see amber_amd.predict.predict).  The *_normalizer arguments are TorchRunningMeanStd modules or None."""
        self._check_train_mode_decoder()
        observations, csr = self.mpn.prepare(observations)
        n, e = csr.num_nodes, csr.num_edges
        out_index, n_out = self._out_index(observations, n)
        desc, base, dec_pieces, dec_sources, tail = self._plan(single_row=(n_out == 1))
        flat, packed = self._packed.get(desc, base + dec_pieces + tail, key_tensors=base + dec_sources + tail
                                        + ([self.readout.bias] if n_out == 1 else []))
        x, edge_attr = observations.x, observations.edge_attr
        engine._require_cuda(x, "x")
        engine._require_cuda(edge_attr, "edge_attr")
        dev = x.device
        x, edge_attr = engine._f32c(x), engine._f32c(edge_attr)
        if x.shape[0] != n or edge_attr.shape[0] != e:
            raise ValueError(f"Must provide one edge index per edge feature vector, given '{(2, e)}' and '{tuple(edge_attr.shape)}' instead.")

        def stats(m, width, what):
            if m is None or width == 0:
                return None, None
            if m.mean.numel() != width:
                raise ValueError(f"amber_amd: {what} statistics of {m.mean.numel()} features, expected {width}")
            return m.mean.to(dev).contiguous(), m.var.to(dev).contiguous()

        nm, nv = stats(node_normalizer, desc.in_node_features, "node")
        em, ev = stats(edge_normalizer, desc.in_edge_features, "edge")
        pm, pv = stats(prediction_normalizer, 1, "prediction")
        if baseline is not None:
            baseline = baseline.to(device=dev, dtype=torch.float32).contiguous().view(-1)
            if baseline.numel() != n_out:
                raise ValueError(f"amber_amd: {n_out} predictions but a baseline (current_sizing_field) of {baseline.numel()}")
        lib = _lib.load()
        ws = torch.empty(lib.amber_mpn_predict_workspace_bytes(C.byref(desc), n, e), dtype=torch.uint8, device=dev)
        out = torch.empty(n_out, 1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.amber_mpn_predict(
                C.byref(desc), C.byref(glue_desc), flat.data_ptr(), packed.data_ptr(), n, e, _lib.ptr(x), _lib.ptr(edge_attr),
                csr.rowptr.data_ptr(), _lib.ptr(csr.src_sorted), _lib.ptr(csr.dst_sorted), _lib.ptr(csr.perm), _lib.ptr(out_index),
                _lib.ptr(nm), _lib.ptr(nv), _lib.ptr(em), _lib.ptr(ev), _lib.ptr(pm), _lib.ptr(pv), _lib.ptr(baseline), out.data_ptr(),
                ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, "predict")
        csr.check()
        return out
