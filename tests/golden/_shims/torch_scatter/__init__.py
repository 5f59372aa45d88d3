"""Stand-in for the absent third-party package `torch_scatter` (version unpinned by the reference).
Restates scatter_mean's published algorithm: scatter_add_ sum, scatter_add count of ones,
count clamped to >= 1, true division."""
import torch


def scatter_mean(src, index, dim=0, dim_size=None):
    assert dim == 0 and src.dim() == 2
    n = int(dim_size) if dim_size is not None else int(index.max()) + 1
    out = torch.zeros(n, src.shape[1], dtype=src.dtype, device=src.device)
    out.scatter_add_(0, index.view(-1, 1).expand_as(src), src)
    count = torch.zeros(n, dtype=src.dtype, device=src.device)
    count.scatter_add_(0, index, torch.ones_like(index, dtype=src.dtype))
    count.clamp_(min=1)
    return out.true_divide_(count.view(-1, 1))
