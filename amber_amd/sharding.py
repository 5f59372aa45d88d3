"""Graph-wise data parallelism for disjoint mesh-graph batches.

A PyG batch is a disjoint union: graphs never exchange messages, so the forward shards at graph boundaries with
NO collective.  Each rank takes a group of graphs balanced on sum(N_g + E_g) (the size measure of the
reference's batcher, /root/reference/src/algorithm/dataloader/graph_batcher.py:34-41), re-bases node ids, builds
its own CSR and runs the full forward; outputs are put back in graph order.  The groups are contiguous cuts of
the batch unless a longest-processing-time assignment balances better (ragged graph sizes); a single graph is
never split ("replicas only" for that case).
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


def partition_lpt(weights: Sequence[int], parts: int) -> List[List[int]]:
    """Longest-processing-time assignment: graphs by descending size, each to the least loaded part (ties: the
    lower part).  Returns the graph ids of every part in ascending order.  Deterministic: every rank computes the same."""
    w = np.asarray(weights, dtype=np.float64)
    order = np.argsort(-w, kind="stable")
    loads = np.zeros(parts)
    groups: List[List[int]] = [[] for _ in range(parts)]
    for g in order:
        k = int(np.argmin(loads))
        groups[k].append(int(g))
        loads[k] += w[g]
    return [sorted(g) for g in groups]


def imbalance(weights: Sequence[int], groups: Sequence[Sequence[int]]) -> float:
    """max part load / mean part load - 1 (0 = perfectly balanced; the ideal speed-up is parts / (1 + imbalance))."""
    w = np.asarray(weights, dtype=np.float64)
    loads = np.array([w[list(g)].sum() for g in groups])
    mean = w.sum() / max(len(groups), 1)
    return float(loads.max() / mean - 1.0) if mean > 0 else 0.0


def assign_graphs(weights: Sequence[int], parts: int) -> List[List[int]]:
    """Graph ids per rank: the contiguous cut (slices of the batch, no gather of rows) unless the LPT assignment is
    better balanced by more than one percent of the mean load."""
    cont = [list(range(lo, hi)) for lo, hi in partition_contiguous(weights, parts)]
    lpt = partition_lpt(weights, parts)
    return lpt if imbalance(weights, lpt) + 0.01 < imbalance(weights, cont) else cont


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


def shard_batch(batch, rank: int, world_size: int, balance: str = "auto") -> GraphBatch:
    """The sub-batch of `rank`: its graphs' nodes and edges with node ids re-based to start at 0.
    balance: "contiguous" = slices of the batch; "lpt"; "auto" = assign_graphs' choice.  `graph_ids` of the result
    lists the graphs it holds (ascending)."""
    ptr, eptr, n_g, e_g = graph_sizes(batch)
    w = n_g + e_g
    if balance == "contiguous":
        ids = [list(range(lo, hi)) for lo, hi in partition_contiguous(w, world_size)][rank]
    elif balance == "lpt":
        ids = partition_lpt(w, world_size)[rank]
    else:
        ids = assign_graphs(w, world_size)[rank]
    contiguous = len(ids) == 0 or ids == list(range(ids[0], ids[-1] + 1))
    if contiguous:
        lo, hi = (ids[0], ids[-1] + 1) if ids else (0, 0)
        n0, n1, e0, e1 = int(ptr[lo]), int(ptr[hi]), int(eptr[lo]), int(eptr[hi])
        out = GraphBatch(
            x=batch.x[n0:n1], edge_attr=batch.edge_attr[e0:e1], edge_index=batch.edge_index[:, e0:e1] - n0,
            ptr=torch.from_numpy(ptr[lo:hi + 1] - n0), edge_ptr=torch.from_numpy(eptr[lo:hi + 1] - e0),
            graph_range=(lo, hi), node_range=(n0, n1), graph_ids=ids)
        if getattr(batch, "batch", None) is not None:
            out.batch = batch.batch[n0:n1] - lo
        if getattr(batch, "mask_output", None) is not None:
            out.mask_output = batch.mask_output[n0:n1]
        return out
    # a scattered set of graphs: gather their rows, shift every graph's node ids from its old to its new offset
    dev = batch.x.device
    new_ptr = np.concatenate([[0], np.cumsum(n_g[ids])]).astype(np.int64)
    new_eptr = np.concatenate([[0], np.cumsum(e_g[ids])]).astype(np.int64)
    node_idx = torch.from_numpy(np.concatenate([np.arange(ptr[g], ptr[g + 1]) for g in ids])).to(dev)
    edge_idx = torch.from_numpy(np.concatenate([np.arange(eptr[g], eptr[g + 1]) for g in ids])).to(dev)
    shift = torch.from_numpy(np.repeat(new_ptr[:-1] - ptr[ids], e_g[ids])).to(dev)
    out = GraphBatch(
        x=batch.x[node_idx], edge_attr=batch.edge_attr[edge_idx], edge_index=batch.edge_index[:, edge_idx] + shift,
        ptr=torch.from_numpy(new_ptr), edge_ptr=torch.from_numpy(new_eptr), graph_ids=ids)
    out.batch = torch.from_numpy(np.repeat(np.arange(len(ids)), n_g[ids])).to(dev)
    if getattr(batch, "mask_output", None) is not None:
        out.mask_output = batch.mask_output[node_idx]
    return out


def shard_stats(batch, world_size: int, balance: str = "auto") -> dict:
    """What every rank gets: nodes / edges per rank, imbalance of the N + E measure, ideal speed-up of the partition."""
    ptr, eptr, n_g, e_g = graph_sizes(batch)
    w = n_g + e_g
    groups = {"contiguous": lambda: [list(range(lo, hi)) for lo, hi in partition_contiguous(w, world_size)],
              "lpt": lambda: partition_lpt(w, world_size)}.get(balance, lambda: assign_graphs(w, world_size))()
    imb = imbalance(w, groups)
    return {"nodes_per_rank": [int(n_g[g].sum()) for g in groups], "edges_per_rank": [int(e_g[g].sum()) for g in groups],
            "graphs_per_rank": [len(g) for g in groups], "imbalance_pct": 100.0 * imb,
            "ideal_speedup": world_size / (1.0 + imb)}


def sharded_forward(forward_fn: Callable, batch, rank: int, world_size: int, gather: bool = True,
                    group=None) -> Optional[torch.Tensor]:
    """Run `forward_fn(sub_batch) -> [n_kept_local, 1]` on this rank's shard.  With gather=True the per-node outputs
    of all ranks are concatenated in graph order on every rank (an all_gather of a few bytes per node, OFF the
    data path: the forward itself needs no communication)."""
    import torch.distributed as dist

    shard = shard_batch(batch, rank, world_size)
    local = forward_fn(shard)
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
    return reassemble_in_graph_order(batch, [o[:s] for o, s in zip(outs, sizes)], world_size).to(dev)


def reassemble_in_graph_order(batch, outs: Sequence[torch.Tensor], world_size: int) -> torch.Tensor:
    """Per-rank outputs (rank k's rows = the kept nodes of its graphs, ascending graph id) -> rows of the whole batch in
    graph order.  Rank order equals graph order only when the flattened assignment is already ascending (the contiguous
    cut); an LPT assignment whose groups happen to be contiguous ranges can still hand a LATER range to an EARLIER rank
    (weights [5, 17, 20] on 3 ranks -> [[2], [1], [0]]), so everything else goes through the per-graph placement."""
    ptr, eptr, n_g, e_g = graph_sizes(batch)
    groups = assign_graphs(n_g + e_g, world_size)
    flat = [g for ids in groups for g in ids]
    if flat == sorted(flat):
        return torch.cat(list(outs), dim=0)
    mask = getattr(batch, "mask_output", None)
    kept = np.asarray([int(mask[ptr[g]:ptr[g + 1]].sum()) if mask is not None else int(n_g[g]) for g in range(len(n_g))])
    pieces = [None] * len(n_g)
    for o, g_ids in zip(outs, groups):
        off = 0
        for g in g_ids:
            pieces[g] = o[off:off + kept[g]]
            off += kept[g]
    return torch.cat([p for p in pieces if p is not None], dim=0)


def broadcast_buffers(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make rank `src`'s buffers of `module` (the running statistics of the normalisers, a decoder's BatchNorm running
    buffers) the buffers of every rank, in place.  The statistics are updated on the host from each rank's own new data
    (/root/reference/src/algorithm/core/amber.py:335); with graph-wise data parallelism one rank stays authoritative and
    the others take its numbers, so that every rank normalises its shard identically.  A no-op
    without an initialised process group or with one rank.  Device tensors under gloo are staged through the host (see
    engine._sync_flat_gradient); RCCL takes them as they are."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    gloo = dist.get_backend(group) == "gloo"
    for buf in module.buffers():
        if buf.is_cuda and gloo:
            host = buf.detach().cpu()
            dist.broadcast(host, src=src, group=group)
            buf.copy_(host)
        else:
            dist.broadcast(buf, src=src, group=group)
