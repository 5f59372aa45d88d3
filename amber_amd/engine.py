"""Host driver between the nn.Module mirrors and the C ABI: network descriptor, parameter
flattening + MFMA re-tiling (cached), workspace, and the compute calls.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from .csr import CSR, _require_cuda, source_order


PRECISIONS = {"fp32": 0, "f16x2": 1, "f16": 2}
# Precision of the MLP contractions used by the nn.Module mirrors: "fp32" = exact fp32 MFMA everywhere; "f16x2" =
# fp32-grade split precision in the edge and node kernels (include/amber_mpn.h, AMBER_PREC_F16X2); both meet the 1e-5
# bar.  "f16" = plain fp16 operands in the same kernels (tolerance 1e-2, opt-in only).
default_precision = os.environ.get("AMBER_MPN_PRECISION", "f16x2")


def make_desc(*, latent: int, num_steps: int, num_step_repeats: int, mlp_layers: int, activation: str,
              residual: Optional[str], layer_norm: Optional[str], in_node_features: int, in_edge_features: int,
              embed_layers: int = 0, embed_activation: str = "leakyrelu", has_head: bool = False,
              decoder_layers: int = 0, decoder_activation: str = "leakyrelu", decoder_norm: Optional[str] = None,
              ln_eps: float = 1e-5, precision: Optional[str] = None) -> _lib.AmberMpnDesc:
    def lut(table, key, what):
        k = key.lower() if isinstance(key, str) else key
        if k not in table:
            raise ValueError(f"Unknown {what} '{key}'")
        return table[k]

    return _lib.AmberMpnDesc(
        latent=latent, num_steps=num_steps, num_step_repeats=num_step_repeats, mlp_layers=mlp_layers,
        activation=lut(_lib.ACT, activation, "activation function"), residual=lut(_lib.MODE, residual, "residual_connections"),
        layer_norm=lut(_lib.MODE, layer_norm, "layer_norm"), in_node_features=in_node_features,
        in_edge_features=in_edge_features, embed_layers=embed_layers,
        embed_activation=lut(_lib.ACT, embed_activation, "activation function"), has_head=1 if has_head else 0,
        decoder_layers=decoder_layers, decoder_activation=lut(_lib.ACT, decoder_activation, "activation function"),
        decoder_norm=lut(_lib.NORM, decoder_norm, "latent_normalization"),
        precision=lut(PRECISIONS, precision if precision is not None else default_precision, "precision"), ln_eps=ln_eps)


class PackedParams:
    """Flat fp32 parameter buffer (include/amber_mpn.h order) + its MFMA-tiled copy, rebuilt only when a
    source tensor changes (identity, in-place version or device)."""

    def __init__(self):
        self._key = None
        self.flat: Optional[torch.Tensor] = None
        self.packed: Optional[torch.Tensor] = None

    def get(self, desc: _lib.AmberMpnDesc, tensors: Sequence[torch.Tensor], key_skip: int = 0,
            key_tensors: Optional[Sequence[torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """`tensors`: the flat-buffer pieces in order.  The cache key is built from `key_tensors` (default:
        `tensors[key_skip:]`), i.e. from the module's own parameters / buffers, not from derived temporaries."""
        src = key_tensors if key_tensors is not None else tensors[key_skip:]
        key = tuple((t.data_ptr(), t._version, t.device) for t in src) + (bytes(desc),)
        if key == self._key:
            return self.flat, self.packed
        lib = _lib.load()
        dev = tensors[0].device
        _require_cuda(tensors[0], "model parameters")
        flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors]).contiguous()
        expect = lib.amber_mpn_param_floats(C.byref(desc))
        if expect < 0:
            _lib.check(-2, "network description")
        if flat.numel() != expect:
            raise RuntimeError(f"amber_amd: parameter count {flat.numel()} does not match the description ({expect})")
        packed = torch.empty(lib.amber_mpn_packed_floats(C.byref(desc)), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.amber_mpn_pack_weights(C.byref(desc), flat.data_ptr(), packed.data_ptr(), _lib.stream_ptr(dev))
        _lib.check(rc, "pack_weights")
        self._key, self.flat, self.packed = key, flat, packed
        return flat, packed


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.to(torch.float32).contiguous()


def out_index_from_mask(mask: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """x[mask_output] as an index map: out_index[n] = rank of n among kept nodes, -1 if dropped
    (/root/reference/src/algorithm/architecture/supervised_mpn.py:51-52)."""
    m = mask.to(torch.bool)
    ranks = torch.cumsum(m.to(torch.int32), 0, dtype=torch.int32) - 1
    out_index = torch.where(m, ranks, torch.full_like(ranks, -1))
    n_kept = int(ranks[-1].item()) + 1 if m.numel() > 0 else 0
    return out_index.contiguous(), n_kept


def workspace(desc: _lib.AmberMpnDesc, n: int, e: int, device) -> torch.Tensor:
    nbytes = _lib.load().amber_mpn_workspace_bytes(C.byref(desc), n, e)
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def forward(desc: _lib.AmberMpnDesc, flat: torch.Tensor, packed: torch.Tensor, x: torch.Tensor, edge_attr: torch.Tensor,
            csr: CSR, *, want_node: bool, want_edge: bool, out_index: Optional[torch.Tensor] = None,
            n_out: Optional[int] = None, ws: Optional[torch.Tensor] = None, timed: bool = False):
    """One fused forward (amber_mpn_forward).  Returns (node[N,L] | None, edge[E,L] | None, sizing[n_out,1] | None)
    and, with timed=True, additionally (kernel_ms, kernel_launches) lists per kernel class."""
    lib = _lib.load()
    _require_cuda(x, "x")
    _require_cuda(edge_attr, "edge_attr")
    dev = x.device
    x, edge_attr = _f32c(x), _f32c(edge_attr)
    n, e, latent = csr.num_nodes, csr.num_edges, desc.latent
    if x.shape[0] != n or edge_attr.shape[0] != e:
        raise ValueError(f"Must provide one edge index per edge feature vector, given '{(2, e)}' and '{tuple(edge_attr.shape)}' instead.")
    if x.shape[1] != desc.in_node_features:
        raise ValueError(f"Feature dimensions of nodes do not match. Given '{x.shape[1]}', expected '{desc.in_node_features}")
    if edge_attr.shape[1] != desc.in_edge_features:
        raise ValueError(f"Feature dimensions of edges do not match. Given '{edge_attr.shape[1]}', expected '{desc.in_edge_features}")
    out_node = torch.empty(n, latent, dtype=torch.float32, device=dev) if want_node else None
    out_edge = torch.empty(e, latent, dtype=torch.float32, device=dev) if want_edge else None
    out_sizing = None
    if desc.has_head and n_out is not None:
        out_sizing = torch.empty(n_out, 1, dtype=torch.float32, device=dev)
    if ws is None:
        ws = workspace(desc, n, e, dev)
    args = [C.byref(desc), flat.data_ptr(), packed.data_ptr(), n, e, _lib.ptr(x), _lib.ptr(edge_attr), csr.rowptr.data_ptr(),
            _lib.ptr(csr.src_sorted), _lib.ptr(csr.dst_sorted), _lib.ptr(csr.perm), _lib.ptr(out_index), _lib.ptr(out_node),
            _lib.ptr(out_edge), _lib.ptr(out_sizing), ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)]
    with torch.cuda.device(dev):
        if timed:
            ms = (C.c_float * _lib.NUM_KERNEL_CLASSES)()
            cnt = (C.c_int32 * _lib.NUM_KERNEL_CLASSES)()
            rc = lib.amber_mpn_forward_timed(*args, ms, cnt)
            _lib.check(rc, "forward")
            csr.check()
            return out_node, out_edge, out_sizing, list(ms), list(cnt)
        rc = lib.amber_mpn_forward(*args)
    _lib.check(rc, "forward")
    csr.check()  # deferred index validation of a freshly built CSR: its status word has usually arrived by now
    return out_node, out_edge, out_sizing


def embed(desc, flat, packed, x, edge_attr, csr: CSR):
    """amber_mpn_embed: (x_latent[N,L], e_latent in CSR order [E,L])."""
    lib = _lib.load()
    dev = x.device
    x, edge_attr = _f32c(x), _f32c(edge_attr)
    xl = torch.empty(csr.num_nodes, desc.latent, dtype=torch.float32, device=dev)
    el = torch.empty(csr.num_edges, desc.latent, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.amber_mpn_embed(C.byref(desc), flat.data_ptr(), packed.data_ptr(), csr.num_nodes, csr.num_edges, _lib.ptr(x),
                                 _lib.ptr(edge_attr), _lib.ptr(csr.perm), xl.data_ptr(), el.data_ptr(), 0, 0, _lib.stream_ptr(dev))
    _lib.check(rc, "embed")
    return xl, el


def block(desc, flat, packed, step: int, x_latent: torch.Tensor, e_latent_sorted: torch.Tensor, csr: CSR,
          ws: Optional[torch.Tensor] = None) -> None:
    """amber_mpn_block: one MessagePassingBlock, in place on latent features (edges in CSR order)."""
    lib = _lib.load()
    dev = x_latent.device
    if ws is None:
        ws = workspace(desc, csr.num_nodes, csr.num_edges, dev)
    with torch.cuda.device(dev):
        rc = lib.amber_mpn_block(C.byref(desc), flat.data_ptr(), packed.data_ptr(), step, csr.num_nodes, csr.num_edges,
                                 csr.rowptr.data_ptr(), _lib.ptr(csr.src_sorted), _lib.ptr(csr.dst_sorted), x_latent.data_ptr(),
                                 _lib.ptr(e_latent_sorted), ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "block")


def project(desc, flat, packed, step: int, x_latent: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """amber_mpn_project: (P_s, P_d) = (W0s x, W0d x + b0) of block `step`'s edge MLP."""
    lib = _lib.load()
    dev = x_latent.device
    x_latent = _f32c(x_latent)
    ps, pd = torch.empty_like(x_latent), torch.empty_like(x_latent)
    with torch.cuda.device(dev):
        rc = lib.amber_mpn_project(C.byref(desc), flat.data_ptr(), packed.data_ptr(), step, int(x_latent.shape[0]),
                                   x_latent.data_ptr(), ps.data_ptr(), pd.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "project")
    return ps, pd


def head(desc, flat, packed, x_latent: torch.Tensor, out_index: Optional[torch.Tensor], n_out: int) -> torch.Tensor:
    """amber_mpn_head: readout(decoder(x_latent[mask]))."""
    lib = _lib.load()
    dev = x_latent.device
    out = torch.empty(n_out, 1, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.amber_mpn_head(C.byref(desc), flat.data_ptr(), packed.data_ptr(), int(x_latent.shape[0]), x_latent.data_ptr(),
                                _lib.ptr(out_index), out.data_ptr(), _lib.stream_ptr(dev))
    _lib.check(rc, "head")
    return out


# ---------------------------------------------------------------------------------------------------
# training: forward that keeps the per-step block inputs on a tape + backward (amber_mpn_backward), as one
# torch.autograd.Function over the module's parameters
# ---------------------------------------------------------------------------------------------------
# How parameter gradients are combined across ranks when torch.distributed is initialised (graph-wise data
# parallelism): one all-reduce of the single flat gradient buffer per backward ("mean" = DistributedDataParallel's
# convention, "sum", or None = leave the gradients local).
grad_sync: Optional[str] = "mean"
debug_keep_backward_workspace: Optional[list] = None  # developer diagnostics: set to a list to receive the workspaces


def _sync_flat_gradient(flat_grad: torch.Tensor) -> None:
    import torch.distributed as dist

    if grad_sync is None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    if flat_grad.is_cuda and dist.get_backend() == "gloo":
        # gloo is a host backend (CPU tests, one-GPU rehearsals of the multi-rank path).  Stage through the host here:
        # handed the device tensor, gloo parked both ranks of a rehearsal in all_reduce (DESIGN.md section 6 records what
        # one diagnostic run with AMBER_GLOO_DEVICE_ALLREDUCE=sync - a stream synchronise, then the device tensor - showed).
        if os.environ.get("AMBER_GLOO_DEVICE_ALLREDUCE") == "sync":
            torch.cuda.current_stream(flat_grad.device).synchronize()
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        else:
            host = flat_grad.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat_grad.copy_(host)
    else:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)  # RCCL over xGMI on the GPU box (backend "nccl"); one bucket
    if grad_sync == "mean":
        flat_grad.div_(dist.get_world_size())


class MpnFunction(torch.autograd.Function):
    """(node[N,L], edge[E,L], sizing[n_out,1]) = f(parameters); unwanted outputs are empty tensors.
    Inputs x / edge_attr are data and get no gradient."""

    @staticmethod
    def forward(ctx, desc, packer: PackedParams, x, edge_attr, csr: CSR, out_index, n_out, want_node, want_edge, key_tensors,
                *params):
        """key_tensors: the module's own parameters / buffers the cache key is built from (None: `params`).  Derived
        temporaries among `params` (a spectrally normalised weight is a fresh tensor every forward, and the allocator
        hands the same address out again) must not key the cache."""
        lib = _lib.load()
        _require_cuda(x, "x")
        _require_cuda(edge_attr, "edge_attr")
        dev = x.device
        x, edge_attr = _f32c(x), _f32c(edge_attr)
        n, e, latent = csr.num_nodes, csr.num_edges, desc.latent
        if x.shape[0] != n or edge_attr.shape[0] != e:
            raise ValueError(f"Must provide one edge index per edge feature vector, given '{(2, e)}' and '{tuple(edge_attr.shape)}' instead.")
        flat, packed = packer.get(desc, params, key_tensors=key_tensors)
        out_node = torch.empty(n, latent, dtype=torch.float32, device=dev) if want_node else None
        out_edge = torch.empty(e, latent, dtype=torch.float32, device=dev) if want_edge else None
        out_sizing = torch.empty(n_out, 1, dtype=torch.float32, device=dev) if (desc.has_head and n_out is not None) else None
        tape_bytes = lib.amber_mpn_tape_bytes(C.byref(desc), n, e)
        if tape_bytes == 0:
            _lib.check(-2, "network description")
        tape = torch.empty(tape_bytes, dtype=torch.uint8, device=dev)
        ws = workspace(desc, n, e, dev)
        with torch.cuda.device(dev):
            rc = lib.amber_mpn_forward_train(
                C.byref(desc), flat.data_ptr(), packed.data_ptr(), n, e, _lib.ptr(x), _lib.ptr(edge_attr), csr.rowptr.data_ptr(),
                _lib.ptr(csr.src_sorted), _lib.ptr(csr.dst_sorted), _lib.ptr(csr.perm), _lib.ptr(out_index), _lib.ptr(out_node),
                _lib.ptr(out_edge), _lib.ptr(out_sizing), tape.data_ptr(), tape.numel(), ws.data_ptr(), ws.numel(),
                _lib.stream_ptr(dev))
        _lib.check(rc, "forward_train")
        csr.check()
        ctx.desc, ctx.csr, ctx.out_index = desc, csr, out_index
        ctx.x, ctx.edge_attr, ctx.flat, ctx.packed, ctx.tape = x, edge_attr, flat, packed, tape
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.has = (want_node, want_edge, out_sizing is not None)
        empty = torch.empty(0, dtype=torch.float32, device=dev)
        outs = (out_node if want_node else empty, out_edge if want_edge else empty, out_sizing if out_sizing is not None else empty)
        ctx.mark_non_differentiable(*[o for o, h in zip(outs, ctx.has) if not h])
        return outs

    @staticmethod
    def backward(ctx, g_node, g_edge, g_sizing):
        lib = _lib.load()
        desc, csr = ctx.desc, ctx.csr
        dev = ctx.x.device
        # A training step's CSR (that step's kept edges) is built outside any cache and dropped after this call: collect its
        # deferred index verdict here.  The wait is for the event behind the CSR build, long past by now - not for the forward.
        csr.check(wait=True)
        n, e = csr.num_nodes, csr.num_edges
        want_node, want_edge, has_sizing = ctx.has
        g_node = _f32c(g_node) if want_node and g_node is not None else None
        g_edge = _f32c(g_edge) if want_edge and g_edge is not None else None
        g_sizing = _f32c(g_sizing) if has_sizing and g_sizing is not None else None
        rowptr_src, perm_src = source_order(csr)
        grad = torch.empty(ctx.flat.numel(), dtype=torch.float32, device=dev)
        ws_bytes = lib.amber_mpn_backward_workspace_bytes(C.byref(desc), n, e)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.amber_mpn_backward(
                C.byref(desc), ctx.flat.data_ptr(), ctx.packed.data_ptr(), n, e, _lib.ptr(ctx.x), _lib.ptr(ctx.edge_attr),
                csr.rowptr.data_ptr(), _lib.ptr(csr.src_sorted), _lib.ptr(csr.dst_sorted), _lib.ptr(csr.perm),
                _lib.ptr(rowptr_src), _lib.ptr(perm_src), _lib.ptr(ctx.out_index), _lib.ptr(g_node), _lib.ptr(g_edge),
                _lib.ptr(g_sizing), ctx.tape.data_ptr(), ctx.tape.numel(), grad.data_ptr(), ws.data_ptr(), ws.numel(),
                _lib.stream_ptr(dev))
        _lib.check(rc, "backward")
        ctx.tape = None
        if debug_keep_backward_workspace is not None:
            debug_keep_backward_workspace.append(ws)
        _sync_flat_gradient(grad)
        grads, off = [], 0
        for shape in ctx.shapes:
            k = 1
            for d in shape:
                k *= d
            grads.append(grad[off:off + k].view(shape))
            off += k
        return (None,) * 10 + tuple(grads)


def wants_grad(params: Sequence[torch.Tensor]) -> bool:
    """The autograd path runs when gradients are being recorded for at least one parameter."""
    return torch.is_grad_enabled() and any(p.requires_grad for p in params)
