"""Running mean / variance buffers (mirror of /root/reference/src/algorithm/normalizer/torch_running_mean_std.py:7-57).

Same buffer names, shapes and dtype (float64 `mean`, `var`, `count`), so a reference state_dict loads.  The statistics
are updated on the host side of the training loop from new data (src/algorithm/core/amber.py:335); that bookkeeping is
not on the model-call path and stays plain torch.  What the path READS from here (mean, var) goes to the kernels as
device pointers."""
from typing import Tuple

import torch
from torch import Tensor, nn


class TorchRunningMeanStd(nn.Module):
    def __init__(self, epsilon, shape: Tuple[int, ...] = (), dtype: torch.dtype = torch.float64):
        super().__init__()
        self.register_buffer("mean", torch.zeros(*shape, dtype=dtype))
        self.register_buffer("var", torch.ones(*shape, dtype=dtype))
        self.register_buffer("count", torch.tensor(epsilon, dtype=dtype))

    def update(self, arr: Tensor) -> None:
        """Fold a batch [rows, *shape] into the statistics (parallel-variance merge of Chan et al.)."""
        rows = arr.shape[0]
        b_mean = arr.mean(dim=0).to(self.mean)
        b_var = torch.nan_to_num(arr.var(dim=0), nan=0.0).to(self.var)  # one row: unbiased variance is NaN -> 0
        total = self.count + rows
        shift = b_mean - self.mean
        second = self.var * self.count + b_var * rows + torch.square(shift) * self.count * rows / total
        self.mean = self.mean + shift * rows / total
        self.var = second / total
        self.count = total

    def __repr__(self):
        return f"TorchRunningMeanStd(mean={self.mean}, var={self.var}, count={self.count})"
