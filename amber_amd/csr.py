"""Destination-sorted CSR of a PyG-style edge_index, built on the GPU once per mesh.

Replaces the per-call index handling of `scatter_mean(graph.edge_attr, dest_indices, ...)`
(/root/reference/src/mpn/message_passing_modules.py:46-48,67-68).  The sort is stable, so inside a
destination's row the edges keep the reference's edge order.
"""
from __future__ import annotations

import os
import weakref
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib


@dataclass
class CSR:
    num_nodes: int
    num_edges: int
    rowptr: torch.Tensor      # int32 [N+1]
    src_sorted: torch.Tensor  # int32 [E]
    dst_sorted: torch.Tensor  # int32 [E]
    perm: torch.Tensor        # int32 [E]: original edge id at sorted position p

    def check(self, wait: bool = False) -> None:
        """Deferred index validation (build_csr(validate="deferred")): raises IndexError once the status word of the build
        has reached the host and says that edge_index held node ids outside [0, N).  wait=False only looks (no
        synchronisation); wait=True blocks until the word is there."""
        pending = getattr(self, "_pending", None)
        if pending is None:
            return
        ring, slot, generation = pending
        verdict = ring.verdict(slot, generation, wait)
        if verdict is None:
            return  # not there yet
        self._pending = None
        if verdict:
            raise IndexError(f"edge_index contains node ids outside [0, {self.num_nodes})")


class _StatusRing:
    """Pinned host words for the deferred status read-backs (allocating pinned memory per mesh would cost more than the
    synchronisation it avoids).  The slot itself holds what the verdict needs - the event behind the copy and the mesh's
    node count - so it does not depend on the CSR object staying alive: a slot is handed out again only after its previous
    verdict has been read, and a bad one that nobody has collected is raised there (naming the earlier mesh) rather than
    dropped."""

    def __init__(self, slots: int = 64):
        self.slots = slots
        self.host = None
        self.pending = [None] * slots   # (event, num_nodes) of a verdict nobody has read yet
        self.generation = [0] * slots
        self.next = 0

    def verdict(self, slot: int, generation: int, wait: bool):
        """True (bad ids) / False (fine) once the word is there, None while it is not.  A slot whose generation has moved
        on was already read at hand-over (and raised there if it was bad)."""
        if self.generation[slot] != generation or self.pending[slot] is None:
            return False
        event, _ = self.pending[slot]
        if wait:
            event.synchronize()
        elif not event.query():
            return None
        self.pending[slot] = None
        return int(self.host[slot]) != 0

    def take(self):
        """(slot index, generation, pinned word).  The slot's previous verdict, if still unread, is waited for first."""
        if self.host is None:
            self.host = torch.zeros(self.slots, dtype=torch.int32).pin_memory()
        i = self.next
        old = self.pending[i]
        if old is not None:
            event, num_nodes = old
            event.synchronize()
            self.pending[i] = None  # read now; `next` stays, so the coming build takes this very slot
            if int(self.host[i]) != 0:
                raise IndexError(f"an edge_index built {self.slots} meshes ago contained node ids outside [0, {num_nodes}) "
                                 f"and its CSR was never used again, so this is the first chance to report it")
        self.next = (i + 1) % self.slots
        self.generation[i] += 1
        return i, self.generation[i], self.host[i:i + 1]

    def arm(self, slot: int, event, num_nodes: int) -> None:
        self.pending[slot] = (event, num_nodes)


_status_ring = _StatusRing()
_STRICT = bool(os.environ.get("AMBER_CSR_STRICT"))  # diagnostic: validate every build at once (a synchronisation per mesh)


def source_order(csr: CSR):
    """(rowptr_src[N+1], perm_src[E]): the CSR positions regrouped by SOURCE node, stable (perm_src[q] = position in the
    destination-sorted order of the q-th edge in source order).  The backward pass sums the gradients of the gathered
    P_s[src] terms over it, deterministically.  Built with the same integer sort as the CSR itself; cached on `csr`."""
    cached = getattr(csr, "_source_order", None)
    if cached is not None:
        return cached
    if csr.num_edges == 0:
        dev = csr.rowptr.device
        out = (torch.zeros(csr.num_nodes + 1, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev))
    else:
        by_src = build_csr(torch.stack([csr.dst_sorted.long(), csr.src_sorted.long()]), csr.num_nodes, validate=False)
        out = (by_src.rowptr, by_src.perm)
    csr._source_order = out
    return out


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"amber_amd: `{name}` is on {t.device}; the MPN path runs on MI355X only (move the batch to "
            f"'cuda'). There is no CPU path."
        )


def build_csr(edge_index: torch.Tensor, num_nodes: int, validate=True) -> CSR:
    """Build the CSR for edge_index[2, E] (row 0 = source, row 1 = destination).

    `validate=True` reads back one status word (a device synchronisation, once per mesh) and raises
    IndexError when an index lies outside [0, num_nodes), the error torch indexing would raise.
    `validate="deferred"` (the model's own path: AMBER's refinement loop brings a new mesh every iteration) copies the
    word to pinned host memory behind the build WITHOUT waiting; `CSR.check()` - called by every forward before and after
    it queues its kernels - raises the same IndexError as soon as the word has arrived, at the latest on the next use of
    the CSR or after 64 further builds.  That is also when torch's own GPU indexing would report it (a device-side assert
    surfaces at the next synchronisation).  The CSR of an invalid edge_index is memory-safe either way: the builder clamps
    the ids it stores.
    """
    lib = _lib.load()
    _require_cuda(edge_index, "edge_index")
    if validate == "deferred" and _STRICT:
        validate = True
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"Edge index must have shape (2, num_edges), given '{tuple(edge_index.shape)}' instead.")
    ei = edge_index.long().contiguous()
    dev = ei.device
    E, N = int(ei.shape[1]), int(num_nodes)
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    src_sorted = torch.empty(E, dtype=torch.int32, device=dev)
    dst_sorted = torch.empty(E, dtype=torch.int32, device=dev)
    perm = torch.empty(E, dtype=torch.int32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.amber_mpn_csr_workspace_bytes(N, E)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.amber_mpn_csr_build(ei.data_ptr(), E, N, rowptr.data_ptr(), src_sorted.data_ptr(), dst_sorted.data_ptr(),
                                     perm.data_ptr(), status.data_ptr(), ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev))
    _lib.check(rc, "csr_build")
    csr = CSR(N, E, rowptr, src_sorted, dst_sorted, perm)
    if validate == "deferred":
        slot, generation, host = _status_ring.take()
        host.copy_(status, non_blocking=True)
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(dev))
        _status_ring.arm(slot, event, N)
        csr._pending = (_status_ring, slot, generation)
    elif validate and int(status.item()) != 0:
        raise IndexError(f"edge_index contains node ids outside [0, {N})")
    return csr


class CSRCache:
    """Remembers the CSR of the last few edge_index tensors ("sorted once per mesh").

    A hit needs the SAME tensor object, unmodified (`_version`), so a recycled address cannot alias."""

    def __init__(self, capacity: int = 4):
        self.capacity = capacity
        self._entries = []  # (weakref(tensor), version, num_nodes, CSR)

    def get(self, edge_index: torch.Tensor, num_nodes: int) -> CSR:
        for ref, version, n, csr in self._entries:
            if ref() is edge_index and version == edge_index._version and n == num_nodes:
                csr.check()
                return csr
        csr = build_csr(edge_index, num_nodes, validate="deferred")
        self._entries = [e for e in self._entries if e[0]() is not None][-(self.capacity - 1):] if self.capacity > 1 else []
        self._entries.append((weakref.ref(edge_index), edge_index._version, num_nodes, csr))
        return csr


def csr_of(graph, num_nodes: int, cache: Optional[CSRCache]) -> CSR:
    """CSR for a graph: an `amber_csr` attribute set by the caller wins, else the cache, else a fresh build."""
    pre = getattr(graph, "amber_csr", None)
    if isinstance(pre, CSR) and pre.num_nodes == num_nodes and pre.num_edges == int(graph.edge_index.shape[1]):
        return pre
    if cache is not None:
        return cache.get(graph.edge_index, num_nodes)
    return build_csr(graph.edge_index, num_nodes, validate="deferred")


def segment_mean(values_sorted: torch.Tensor, rowptr: torch.Tensor) -> torch.Tensor:
    """Deterministic CSR segmented mean (scatter_mean on destination-sorted rows; zero-degree rows -> 0)."""
    lib = _lib.load()
    _require_cuda(values_sorted, "values")
    v = values_sorted.float().contiguous()
    n = int(rowptr.shape[0]) - 1
    out = torch.empty(n, v.shape[1], dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        rc = lib.amber_mpn_segment_mean(v.data_ptr(), rowptr.data_ptr(), n, int(v.shape[0]), int(v.shape[1]), out.data_ptr(),
                                        _lib.stream_ptr(v.device))
    _lib.check(rc, "segment_mean")
    return out


def permute_rows(rows: torch.Tensor, perm: torch.Tensor, gather: bool) -> torch.Tensor:
    """gather: out[p] = rows[perm[p]] (edge_index order -> CSR order); else out[perm[p]] = rows[p]."""
    lib = _lib.load()
    _require_cuda(rows, "rows")
    r = rows.float().contiguous()
    out = torch.empty_like(r)
    with torch.cuda.device(r.device):
        rc = lib.amber_mpn_permute_rows(r.data_ptr(), perm.data_ptr(), int(r.shape[0]), int(r.shape[1]), 1 if gather else 0,
                                        out.data_ptr(), _lib.stream_ptr(r.device))
    _lib.check(rc, "permute_rows")
    return out
