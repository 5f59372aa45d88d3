"""LatentMLP parameter container (mirror of /root/reference/src/mpn/common/latent_mlp.py:49-119).

`num_layers` x (Linear -> activation), activation after every layer.  The ModuleList interleaves Linear and
activation modules exactly like the reference, so state_dict keys (`mlp_layers.{0,2,..}.weight/bias`) match.
The arithmetic runs inside the fused HIP kernels (amber_amd/csrc/mpn_kernels.hip: tile_gemm + tile_act).
"""
from typing import Dict, List

import torch
from torch import nn

_ACTIVATIONS = {"relu": nn.ReLU, "leakyrelu": nn.LeakyReLU, "tanh": nn.Tanh}


def build_latent_mlp(*, in_features: int, config: Dict, latent_dimension: int) -> nn.ModuleList:
    name = config.get("activation_function").lower()
    if name not in _ACTIVATIONS:
        raise ValueError("Unknown activation function '{}'".format(name))
    layers = nn.ModuleList()
    fan_in = in_features
    for _ in range(int(config.get("num_layers"))):
        layers.append(nn.Linear(in_features=fan_in, out_features=latent_dimension))
        layers.append(_ACTIVATIONS[name]())
        fan_in = latent_dimension
    return layers


class LatentMLP(nn.Module):
    def __init__(self, *, in_features: int, latent_dimension: int, config: Dict):
        super().__init__()
        self.mlp_layers = build_latent_mlp(in_features=in_features, config=config, latent_dimension=latent_dimension)
        self.in_features = in_features
        self.latent_dimension = latent_dimension
        self.activation_function = config.get("activation_function").lower()
        self.num_layers = int(config.get("num_layers"))

    def linears(self) -> List[nn.Linear]:
        return [m for m in self.mlp_layers if isinstance(m, nn.Linear)]

    def param_list(self) -> List[torch.Tensor]:
        """[W0, b0, W1, b1, ...] in the flat-buffer order of include/amber_mpn.h."""
        out = []
        for lin in self.linears():
            out += [lin.weight, lin.bias]
        return out

    def forward(self, tensor: torch.Tensor) -> torch.Tensor:
        raise RuntimeError(
            "amber_amd.LatentMLP holds parameters only: its layers execute fused inside the HIP edge / node kernels. "
            "Call MessagePassingBase / MessagePassingBlock / SupervisedMPN instead."
        )
