"""Host driver between the nn.Module mirrors and the C ABI: network descriptor, parameter
flattening + MFMA re-tiling (cached), workspace, and the compute calls.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from .csr import CSR, _require_cuda


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
            return out_node, out_edge, out_sizing, list(ms), list(cnt)
        rc = lib.amber_mpn_forward(*args)
    _lib.check(rc, "forward")
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
