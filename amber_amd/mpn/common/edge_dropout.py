"""Edge dropout (mirror of /root/reference/src/mpn/common/edge_dropout.py:6-28).

Eval mode is the identity, which is all the forward path needs.  Training-mode dropout changes E every step and
needs the (not yet built) backward kernels: SURVEY.md section 8f rank 1."""
from torch import nn


class EdgeDropout(nn.Module):
    def __init__(self, dropout_prob: float):
        super().__init__()
        self.dropout_prob = dropout_prob

    def forward(self, graph):
        if self.training and self.dropout_prob > 0.0:
            raise NotImplementedError("amber_amd: training-mode edge dropout needs the backward path (not built yet)")
        return None

    def __repr__(self):
        return f"EdgeDropout(p={self.dropout_prob})"
