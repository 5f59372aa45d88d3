"""Decoder MLP container (mirror of /root/reference/src/algorithm/architecture/mlp.py:29-130).

Per layer: Linear -> activation [-> LayerNorm | BatchNorm1d] [-> Dropout], with the same ModuleList indices as the
reference so `decoder_mlp.mlp_layers.{i}.*` checkpoint keys load.  Runs fused in the last node kernel (K10).
"""
from typing import List, Tuple

import torch
from torch import nn as nn
from torch.nn.utils import spectral_norm as apply_spectral_norm
from torch.nn.utils.spectral_norm import SpectralNorm

_ACTIVATIONS = {"relu": nn.ReLU, "leakyrelu": nn.LeakyReLU, "elu": nn.ELU, "swish": nn.SiLU, "silu": nn.SiLU,
                "mish": nn.Mish, "gelu": nn.GELU, "tanh": nn.Tanh}


class SaveBatchnorm1d(nn.BatchNorm1d):
    """BatchNorm that passes 1-sample inputs through unchanged (reference mlp.py:7-26)."""

    def forward(self, tensor: torch.Tensor) -> torch.Tensor:
        if tensor.shape[0] == 1 or len(tensor.shape) == 1:
            return tensor
        return super().forward(input=tensor)


def build_mlp(*, in_features: int, mlp_config, latent_dimension: int) -> nn.ModuleList:
    name = mlp_config.get("activation_function").lower()
    if name not in _ACTIVATIONS:
        raise ValueError("Unknown activation function '{}'".format(name))
    reg = mlp_config.get("regularization")
    spectral = bool(reg.get("spectral_norm", False))
    dropout = reg.get("dropout", 0.0)
    norm = reg.get("latent_normalization", None)
    layers = nn.ModuleList()
    fan_in = in_features
    for _ in range(mlp_config.get("num_layers")):
        lin = nn.Linear(in_features=fan_in, out_features=latent_dimension)
        layers.append(apply_spectral_norm(lin) if spectral else lin)  # reference mlp.py:62-65, same torch utility
        layers.append(_ACTIVATIONS[name]())
        if norm in ["batch", "batch_norm"]:
            layers.append(SaveBatchnorm1d(num_features=latent_dimension))
        elif norm in ["layer", "layer_norm"]:
            layers.append(nn.LayerNorm(normalized_shape=latent_dimension))
        if dropout:
            layers.append(nn.Dropout(p=dropout))
        fan_in = latent_dimension
    return layers


class MLP(nn.Module):
    def __init__(self, *, in_features: int, latent_dimension: int, mlp_config):
        super().__init__()
        self.mlp_layers = build_mlp(in_features=in_features, mlp_config=mlp_config, latent_dimension=latent_dimension)
        self.in_features = in_features
        self.activation_function = mlp_config.get("activation_function").lower()
        self.num_layers = int(mlp_config.get("num_layers"))
        norm = mlp_config.get("regularization").get("latent_normalization", None)
        self.latent_normalization = {"batch_norm": "batch", "layer_norm": "layer"}.get(norm, norm)
        self._spectral_cache = {}

    def _spectral_weight(self, lin: nn.Linear, hook: SpectralNorm) -> torch.Tensor:
        """The weight a spectrally normalised Linear uses in this forward: torch's own hook arithmetic
        (weight_orig / (u . W v); in training mode after one power iteration that updates u and v in place), which the
        reference runs as the layer's forward pre-hook.  In eval mode without autograd the result is cached until
        weight_orig / u / v change."""
        orig, u, v = lin.weight_orig, lin.weight_u, lin.weight_v
        if lin.training or (torch.is_grad_enabled() and orig.requires_grad):
            return hook.compute_weight(lin, do_power_iteration=lin.training)
        key = (orig.data_ptr(), orig._version, u._version, v._version)
        hit = self._spectral_cache.get(id(lin))
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, hook.compute_weight(lin, do_power_iteration=False))
            self._spectral_cache[id(lin)] = hit
        return hit[1]

    def param_list(self, single_row: bool) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        """(flat-buffer pieces, source tensors for the cache key).  Eval-mode BatchNorm is folded to a per-feature
        scale / shift; with a single input row SaveBatchnorm1d is the identity.  Spectral norm is a re-parametrisation of
        the weight: the kernels get the effective weight."""
        pieces, sources = [], []
        mods = list(self.mlp_layers)
        i = 0
        while i < len(mods):
            lin = mods[i]
            assert isinstance(lin, nn.Linear)
            hook = next((h for h in lin._forward_pre_hooks.values() if isinstance(h, SpectralNorm)), None)
            if hook is not None:
                pieces += [self._spectral_weight(lin, hook), lin.bias]
                sources += [lin.weight_orig, lin.weight_u, lin.weight_v, lin.bias]
            else:
                pieces += [lin.weight, lin.bias]
                sources += [lin.weight, lin.bias]
            i += 2  # Linear, activation
            if i < len(mods) and isinstance(mods[i], nn.LayerNorm):
                pieces += [mods[i].weight, mods[i].bias]
                sources += [mods[i].weight, mods[i].bias]
                i += 1
            elif i < len(mods) and isinstance(mods[i], nn.BatchNorm1d):
                bn = mods[i]
                if single_row:
                    scale, shift = torch.ones_like(bn.weight), torch.zeros_like(bn.bias)
                else:
                    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    shift = bn.bias - bn.running_mean * scale
                pieces += [scale.detach(), shift.detach()]
                sources += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
                i += 1
            if i < len(mods) and isinstance(mods[i], nn.Dropout):
                i += 1  # identity in eval mode
        return pieces, sources

    def forward(self, tensor: torch.Tensor) -> torch.Tensor:
        raise RuntimeError("amber_amd.MLP holds parameters only; the decoder runs fused inside SupervisedMPN.forward (HIP).")
