"""Learned constant embedding for graphs without raw features
(mirror of /root/reference/src/mpn/common/empty_linear_layer.py:6-60)."""
import torch
from torch import nn


class EmptyLinear(nn.Module):
    """Holds `bias[out_features]` ~ U(-1, 1); the embedding kernel broadcasts it to every row (K0/K1 with F = 0)."""

    def __init__(self, out_features: int, bias: bool = True, device=None, dtype=None, **kwargs) -> None:
        super().__init__()
        self.out_features = out_features
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features, device=device, dtype=dtype))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        if self.bias is not None:
            nn.init.uniform_(self.bias, -1, 1)

    def forward(self, tensor: torch.Tensor) -> torch.Tensor:
        # trivial enough to state directly; the fused path broadcasts inside the embedding kernel
        if self.bias is None:
            return torch.zeros(tensor.shape[0], self.out_features, device=tensor.device)
        return self.bias.unsqueeze(0).expand(tensor.shape[0], -1).contiguous()

    def extra_repr(self) -> str:
        return f"out_features={self.out_features}, bias={self.bias is not None}"
