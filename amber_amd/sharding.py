"""Graph-wise data parallelism for disjoint mesh-graph batches.

A PyG batch is a disjoint union: graphs never exchange messages, so the forward shards at graph boundaries with
NO collective.  Each rank takes a contiguous group of graphs balanced on sum(N_g + E_g) (the size measure of the
reference's batcher, /root/reference/src/algorithm/dataloader/graph_batcher.py:34-41), re-bases node ids, builds
its own CSR and runs the full forward; outputs concatenate in graph order.  A single graph is never split
("replicas only" for that case).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .synthetic import GraphBatch


def partition_contiguous(weights: Sequence[int], parts: int) -> List[Tuple[int, int]]:
    """Split `weights` (one per graph) into `parts` contiguous groups [lo, hi) with balanced sums: boundary k is
    placed where the prefix sum is nearest k/parts of the total.  Groups may be empty when there are fewer graphs
    than parts."""
    w = np.asarray(weights, dtype=np.float64)
    prefix = np.concatenate([[0.0], np.cumsum(w)])
    total = prefix[-1]
    bounds = [0]
    for k in range(1, parts):
        target = total * k / parts
        i = int(np.searchsorted(prefix, target))
        if i > 0 and (i >= len(prefix) or abs(prefix[i - 1] - target) <= abs(prefix[i] - target)):
            i -= 1
        bounds.append(min(max(i, bounds[-1]), len(w)))
    bounds.append(len(w))
    return [(bounds[k], bounds[k + 1]) for k in range(parts)]


def graph_sizes(batch) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """(ptr[G+1], edge_ptr[G+1], nodes per graph, edges per graph) of a batch.  Uses `ptr` / `edge_ptr` when present,
    otherwise derives them from the `batch` vector (edges are grouped by graph in a PyG batch)."""
    if getattr(batch, "ptr", None) is not None:
        ptr = batch.ptr.cpu().numpy().astype(np.int64)
    else:
        counts = np.bincount(batch.batch.cpu().numpy())
        ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    if getattr(batch, "edge_ptr", None) is not None:
        eptr = batch.edge_ptr.cpu().numpy().astype(np.int64)
    else:
        src = batch.edge_index[0].cpu().numpy()
        graph_of_edge = np.searchsorted(ptr, src, side="right") - 1
        if np.any(np.diff(graph_of_edge) < 0):
            raise ValueError("edges of the batch are not grouped by graph; cannot shard without re-ordering")
        eptr = np.concatenate([[0], np.cumsum(np.bincount(graph_of_edge, minlength=len(ptr) - 1))]).astype(np.int64)
    return ptr, eptr, np.diff(ptr), np.diff(eptr)


def shard_batch(batch, rank: int, world_size: int) -> GraphBatch:
    """The sub-batch of `rank`: its graphs' nodes and edges with node ids re-based to start at 0."""
    ptr, eptr, n_g, e_g = graph_sizes(batch)
    lo, hi = partition_contiguous(n_g + e_g, world_size)[rank]
    n0, n1, e0, e1 = int(ptr[lo]), int(ptr[hi]), int(eptr[lo]), int(eptr[hi])
    out = GraphBatch(
        x=batch.x[n0:n1], edge_attr=batch.edge_attr[e0:e1], edge_index=batch.edge_index[:, e0:e1] - n0,
        ptr=torch.from_numpy(ptr[lo:hi + 1] - n0), edge_ptr=torch.from_numpy(eptr[lo:hi + 1] - e0),
        graph_range=(lo, hi), node_range=(n0, n1))
    if getattr(batch, "batch", None) is not None:
        out.batch = batch.batch[n0:n1] - lo
    if getattr(batch, "mask_output", None) is not None:
        out.mask_output = batch.mask_output[n0:n1]
    return out


def sharded_forward(forward_fn: Callable, batch, rank: int, world_size: int, gather: bool = True,
                    group=None) -> Optional[torch.Tensor]:
    """Run `forward_fn(sub_batch) -> [n_kept_local, 1]` on this rank's shard.  With gather=True the per-node outputs
    of all ranks are concatenated in graph order on every rank (an all_gather of a few bytes per node, OFF the
    data path: the forward itself needs no communication)."""
    import torch.distributed as dist

    local = forward_fn(shard_batch(batch, rank, world_size))
    if not gather or world_size == 1:
        return local
    # gloo is a host backend (CPU tests, one-GPU rehearsals): stage device tensors through the host ourselves - its own
    # staging was seen to hang on ROCm while the stream still had kernels queued.  RCCL ("nccl") takes them as they are.
    dev = local.device
    comm = torch.device("cpu") if (dev.type == "cuda" and dist.get_backend(group) == "gloo") else dev
    sizes = [torch.zeros(1, dtype=torch.int64, device=comm) for _ in range(world_size)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=comm), group=group)
    sizes = [int(s.item()) for s in sizes]
    pad = max(sizes)
    buf = torch.zeros(pad, *local.shape[1:], dtype=local.dtype, device=comm)
    buf[: local.shape[0]] = local.to(comm)
    outs = [torch.empty_like(buf) for _ in range(world_size)]
    dist.all_gather(outs, buf, group=group)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], dim=0).to(dev)
