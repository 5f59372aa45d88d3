"""Edge dropout (mirror of /root/reference/src/mpn/common/edge_dropout.py:6-28).

Eval mode is the identity.  In training every edge is kept with probability 1 - p (torch_geometric.utils.dropout_edge:
`mask = rand(E) >= p`, drawn from torch's generator of the edge_index device) and edge_index / edge_attr are filtered
IN PLACE on the graph object handed in, as in the reference; MessagePassingBase hands in a private shallow view when
create_graph_copy is set.  The kept edges then get their own CSR (the sort cannot be cached across dropout draws)."""
import torch
from torch import nn


def dropout_edge(edge_index: torch.Tensor, p: float = 0.5, training: bool = True):
    """Restatement of torch_geometric.utils.dropout_edge (third party, not installed) for the undirected=False case the
    reference uses: returns (edge_index[:, mask], mask)."""
    if p < 0.0 or p > 1.0:
        raise ValueError(f"Dropout probability has to be between 0 and 1 (got {p}")
    num_edges = edge_index.shape[1]
    if not training or p == 0.0:
        return edge_index, edge_index.new_ones(num_edges, dtype=torch.bool)
    edge_mask = torch.rand(num_edges, device=edge_index.device) >= p
    return edge_index[:, edge_mask], edge_mask


class EdgeDropout(nn.Module):
    def __init__(self, dropout_prob: float):
        super().__init__()
        self.dropout_prob = dropout_prob

    def forward(self, graph):
        if not self.training or self.dropout_prob == 0.0:
            return  # identity (the reference re-indexes with an all-true mask here: same values)
        graph.edge_index, edge_mask = dropout_edge(edge_index=graph.edge_index, p=self.dropout_prob, training=self.training)
        graph.edge_attr = graph.edge_attr[edge_mask]

    def __repr__(self):
        return f"EdgeDropout(p={self.dropout_prob})"
