import torch


def dropout_edge(edge_index, p=0.5, force_undirected=False, training=True):
    """Eval-mode identity / training-mode Bernoulli edge mask (semantics of PyG's dropout_edge)."""
    if not training or p == 0.0:
        return edge_index, edge_index.new_ones(edge_index.size(1), dtype=torch.bool)
    mask = torch.rand(edge_index.size(1), device=edge_index.device) >= p
    return edge_index[:, mask], mask
