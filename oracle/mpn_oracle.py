"""CPU oracle for AMBER's MPN forward path.  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-fp32 CPU restatement of the reference algorithm.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it,
and only as the checker / the timed CPU baseline -- never as a product path.  The
product (`amber_amd`) never imports anything from `oracle/`.

Pinning: the reference ships no tests, golden vectors or checkpoints.
This oracle is therefore pinned against outputs of the reference's own classes run in
the build container (tests/golden/make_golden.py imports /root/reference/src/mpn under
stand-ins for the absent third-party packages and writes tests/golden/*.npz).
One link stays "parity unpinned": `torch_scatter.scatter_mean` is a third-party
package that is neither vendored in the reference nor installed here (version not
pinned by the reference's env files); `scatter_mean` below restates its published
algorithm (sum via scatter_add_, count via scatter_add of ones, count clamped to >= 1,
true division).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Parameters are looked up by the reference's own state_dict names.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# configuration helpers
# --------------------------------------------------------------------------------------
def _activation(name: str):
    """Activation zoo of LatentMLP (src/mpn/common/latent_mlp.py:76-83) and of the decoder
    MLP (src/algorithm/architecture/mlp.py:68-83).  All with torch defaults."""
    name = name.lower()
    if name == "relu":
        return F.relu
    if name == "leakyrelu":
        return lambda t: F.leaky_relu(t, 0.01)
    if name == "tanh":
        return torch.tanh
    if name == "elu":
        return F.elu
    if name in ("swish", "silu"):
        return F.silu
    if name == "mish":
        return F.mish
    if name == "gelu":
        return F.gelu
    raise ValueError(f"Unknown activation function '{name}'")


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """torch_scatter.scatter_mean(src, index, dim=0, dim_size=N) restated (third party, absent).
    Call site: src/mpn/message_passing_modules.py:68.  Zero in-degree rows stay 0."""
    out = torch.zeros(dim_size, src.shape[1], dtype=src.dtype, device=src.device)
    out.scatter_add_(0, index.view(-1, 1).expand_as(src), src)
    count = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
    count.scatter_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype, device=src.device))
    count.clamp_(min=1)
    return out.true_divide_(count.view(-1, 1))


def latent_mlp(params: Dict[str, torch.Tensor], prefix: str, t: torch.Tensor, num_layers: int, activation: str):
    """LatentMLP.forward: num_layers x (Linear -> activation), activation after EVERY layer.
    src/mpn/common/latent_mlp.py:71-85 (build), :116-119 (forward).  The ModuleList interleaves
    Linear and activation modules, so Linear i sits at index 2*i."""
    act = _activation(activation)
    for i in range(num_layers):
        t = act(F.linear(t, params[f"{prefix}.mlp_layers.{2 * i}.weight"], params[f"{prefix}.mlp_layers.{2 * i}.bias"]))
    return t


def embedding(params: Dict[str, torch.Tensor], prefix: str, t: torch.Tensor, in_features: int,
              embedding_cfg: Optional[dict]) -> torch.Tensor:
    """Embedding.forward (src/mpn/common/embedding.py:26-54): EmptyLinear for 0 input features
    (broadcast bias, src/mpn/common/empty_linear_layer.py:44-45), plain Linear when no embedding
    config is given, LatentMLP otherwise."""
    if in_features == 0:
        return params[f"{prefix}.embedding.bias"].unsqueeze(0).repeat(t.shape[0], 1)
    if embedding_cfg is None:
        return F.linear(t, params[f"{prefix}.embedding.weight"], params[f"{prefix}.embedding.bias"])
    mlp_cfg = embedding_cfg["mlp"]
    return latent_mlp(params, f"{prefix}.embedding", t, mlp_cfg["num_layers"], mlp_cfg["activation_function"])


# --------------------------------------------------------------------------------------
# the message passing network
# --------------------------------------------------------------------------------------
def mpn_block(params, prefix, x, e, edge_index, *, latent, num_layers, activation, residual, layer_norm, eps=1e-5):
    """One MessagePassingBlock.forward (src/mpn/message_passing_block.py:107-128).

    Order: store old -> edge module -> inner edge residual -> inner edge LN -> node module ->
    inner node residual -> inner node LN -> outer residual (node, edge) -> outer LN (node, edge).
    """
    x_old, e_old = x, e  # :117, :130-134
    src, dst = edge_index[0], edge_index[1]
    # edge module, src/mpn/message_passing_modules.py:46-52: cat order is [x[src], x[dst], e]
    agg_in = torch.cat([x[src], x[dst], e], 1)
    e = latent_mlp(params, f"{prefix}.edge_module._mlp", agg_in, num_layers, activation)
    if residual == "inner":
        e = e + e_old  # :139-140
    if layer_norm == "inner":
        e = F.layer_norm(e, (latent,), params[f"{prefix}._edge_layer_norms.weight"],
                         params[f"{prefix}._edge_layer_norms.bias"], eps)  # :145-146
    # node module, src/mpn/message_passing_modules.py:67-72: cat order is [x, scatter_mean(e, dst)]
    agg = scatter_mean(e, dst, x.shape[0])
    x = latent_mlp(params, f"{prefix}.node_module._mlp", torch.cat([x, agg], 1), num_layers, activation)
    if residual == "inner":
        x = x + x_old  # :136-137
    if layer_norm == "inner":
        x = F.layer_norm(x, (latent,), params[f"{prefix}._node_layer_norms.weight"],
                         params[f"{prefix}._node_layer_norms.bias"], eps)  # :142-143
    if residual == "outer":  # :91-93, node first then edge
        x = x + x_old
        e = e + e_old
    if layer_norm == "outer":  # :95-97
        x = F.layer_norm(x, (latent,), params[f"{prefix}._node_layer_norms.weight"],
                         params[f"{prefix}._node_layer_norms.bias"], eps)
        e = F.layer_norm(e, (latent,), params[f"{prefix}._edge_layer_norms.weight"],
                         params[f"{prefix}._edge_layer_norms.bias"], eps)
    return x, e


def mpn_forward(params: Dict[str, torch.Tensor], cfg: dict, x: torch.Tensor, edge_attr: torch.Tensor,
                edge_index: torch.Tensor, prefix: str = "", collect_steps: bool = False):
    """MessagePassingBase.forward in eval mode (src/mpn/message_passing_base.py:127-146):
    (copy) -> (edge dropout: identity in eval, src/mpn/common/edge_dropout.py:24) -> input embeddings
    (src/mpn/input_embedding.py:41-42) -> stack (src/mpn/message_passing_stack.py:58-60).

    `cfg` keys: latent_dimension, in_node_features, in_edge_features, embedding (None|dict),
    stack{num_steps, num_step_repeats, residual_connections, layer_norm, mlp{activation_function, num_layers}}.
    Returns (x[N,L], e[E,L]) and optionally the per-step list.
    """
    stack = cfg["stack"]
    latent = cfg["latent_dimension"]
    res = stack.get("residual_connections")
    res = res.lower() if res is not None else None
    ln = stack.get("layer_norm")
    ln = ln.lower() if ln is not None else None
    mlp = stack["mlp"]
    edge_index = edge_index.long()
    x = embedding(params, f"{prefix}input_embeddings.node_input_embedding", x, cfg["in_node_features"], cfg.get("embedding"))
    e = embedding(params, f"{prefix}input_embeddings.edge_input_embedding", edge_attr, cfg["in_edge_features"], cfg.get("embedding"))
    steps = [(x, e)]
    for i in range(stack["num_steps"]):
        for _ in range(stack.get("num_step_repeats", 1)):
            x, e = mpn_block(params, f"{prefix}message_passing_stack._message_passing_steps.{i}", x, e, edge_index,
                             latent=latent, num_layers=mlp["num_layers"], activation=mlp["activation_function"],
                             residual=res, layer_norm=ln)
            if collect_steps:
                steps.append((x, e))
    if collect_steps:
        return x, e, steps
    return x, e


def linear_weight(params, prefix):
    """Weight of a decoder Linear.  With regularization.spectral_norm (mlp.py:62-63, torch.nn.utils.spectral_norm) the
    state_dict holds weight_orig and the power-iteration vectors weight_u / weight_v; in eval mode the layer uses
    weight_orig / sigma with sigma = u . (weight_orig v), no iteration (torch/nn/utils/spectral_norm.py, compute_weight)."""
    if f"{prefix}.weight_orig" in params:
        w = params[f"{prefix}.weight_orig"]
        sigma = torch.dot(params[f"{prefix}.weight_u"], torch.mv(w, params[f"{prefix}.weight_v"]))
        return w / sigma
    return params[f"{prefix}.weight"]


def decoder_mlp(params, prefix, t, dec_cfg: dict, latent: int):
    """MLP.forward of the decoder (src/algorithm/architecture/mlp.py:60-95,128-130), eval mode:
    per layer Linear -> activation [-> LayerNorm | BatchNorm(eval, running stats)] [-> Dropout = identity]."""
    act = _activation(dec_cfg["activation_function"])
    reg = dec_cfg.get("regularization") or {}
    norm = reg.get("latent_normalization")
    dropout = reg.get("dropout", 0.0)
    idx = 0
    for _ in range(dec_cfg["num_layers"]):
        t = act(F.linear(t, linear_weight(params, f"{prefix}.mlp_layers.{idx}"), params[f"{prefix}.mlp_layers.{idx}.bias"]))
        idx += 2
        if norm in ("layer", "layer_norm"):
            t = F.layer_norm(t, (latent,), params[f"{prefix}.mlp_layers.{idx}.weight"], params[f"{prefix}.mlp_layers.{idx}.bias"], 1e-5)
            idx += 1
        elif norm in ("batch", "batch_norm"):
            if t.shape[0] > 1:  # SaveBatchnorm1d, mlp.py:22-26
                t = F.batch_norm(t, params[f"{prefix}.mlp_layers.{idx}.running_mean"], params[f"{prefix}.mlp_layers.{idx}.running_var"],
                                 params[f"{prefix}.mlp_layers.{idx}.weight"], params[f"{prefix}.mlp_layers.{idx}.bias"], False, 0.0, 1e-5)
            idx += 1
        if dropout:
            idx += 1
    return t


def supervised_mpn_forward(params: Dict[str, torch.Tensor], cfg: dict, x, edge_attr, edge_index,
                           mask_output: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SupervisedMPN.forward (src/algorithm/architecture/supervised_mpn.py:39-56):
    mpn -> x[mask_output] -> decoder MLP -> readout Linear(L,1).  Returns [N_masked, 1]."""
    xl, _ = mpn_forward(params, cfg, x, edge_attr, edge_index, prefix="mpn.")
    if mask_output is not None:
        xl = xl[mask_output]
    dec = decoder_mlp(params, "decoder_mlp", xl, cfg["decoder"], cfg["latent_dimension"])
    return F.linear(dec, params["readout.weight"], params["readout.bias"])


# --------------------------------------------------------------------------------------
# CSR oracle (integer work: bit-exact bar)
# --------------------------------------------------------------------------------------
def csr_oracle(edge_index: np.ndarray, num_nodes: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Destination-sorted CSR of edge_index[2,E] (row 0 = source, row 1 = destination,
    src/mpn/message_passing_modules.py:46,67).  Stable counting sort, so inside a destination's
    segment the edges keep the reference's edge order (the order scatter_add_ visits them).
    Returns (rowptr[N+1] i32, perm[E] i32, src_sorted[E] i32, dst_sorted[E] i32)."""
    src = np.asarray(edge_index[0], dtype=np.int64)
    dst = np.asarray(edge_index[1], dtype=np.int64)
    perm = np.argsort(dst, kind="stable")
    counts = np.bincount(dst, minlength=num_nodes)
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return (rowptr.astype(np.int32), perm.astype(np.int32), src[perm].astype(np.int32), dst[perm].astype(np.int32))


def segment_mean_oracle(values: torch.Tensor, rowptr: np.ndarray) -> torch.Tensor:
    """Mean over each CSR row of `values` (already destination-sorted): the CSR form of scatter_mean."""
    n = rowptr.shape[0] - 1
    deg = torch.from_numpy(np.diff(rowptr).astype(np.int64))
    idx = torch.repeat_interleave(torch.arange(n), deg)
    return scatter_mean(values, idx, n)


# --------------------------------------------------------------------------------------
# parameter construction with the reference's names and default init
# --------------------------------------------------------------------------------------
def init_params(cfg: dict, seed: int = 0, supervised: bool = True) -> Dict[str, torch.Tensor]:
    """Random parameters with the reference's state_dict names and torch's default
    nn.Linear / nn.LayerNorm initialisation distributions.  Not bit-identical to instantiating the
    reference classes (module creation order differs); used for synthetic workloads only."""
    g = torch.Generator().manual_seed(seed)
    L = cfg["latent_dimension"]
    p: Dict[str, torch.Tensor] = {}

    def linear(name, fin, fout):
        bound = 1.0 / np.sqrt(fin) if fin > 0 else 1.0
        p[f"{name}.weight"] = (torch.rand(fout, fin, generator=g) * 2 - 1) * bound
        p[f"{name}.bias"] = (torch.rand(fout, generator=g) * 2 - 1) * bound

    def layer_norm(name):
        # perturbed affine so that gamma/beta handling is actually exercised
        p[f"{name}.weight"] = 1.0 + 0.1 * (torch.rand(L, generator=g) * 2 - 1)
        p[f"{name}.bias"] = 0.1 * (torch.rand(L, generator=g) * 2 - 1)

    pre = "mpn." if supervised else ""
    emb_cfg = cfg.get("embedding")
    for kind, fin in (("node", cfg["in_node_features"]), ("edge", cfg["in_edge_features"])):
        name = f"{pre}input_embeddings.{kind}_input_embedding.embedding"
        if fin == 0:
            p[f"{name}.bias"] = torch.rand(L, generator=g) * 2 - 1
        elif emb_cfg is None:
            linear(name, fin, L)
        else:
            prev = fin
            for i in range(emb_cfg["mlp"]["num_layers"]):
                linear(f"{name}.mlp_layers.{2 * i}", prev, L)
                prev = L
    stack = cfg["stack"]
    nl = stack["mlp"]["num_layers"]
    for s in range(stack["num_steps"]):
        sp = f"{pre}message_passing_stack._message_passing_steps.{s}"
        for mod, fin in (("edge_module", 3 * L), ("node_module", 2 * L)):
            prev = fin
            for i in range(nl):
                linear(f"{sp}.{mod}._mlp.mlp_layers.{2 * i}", prev, L)
                prev = L
        if (stack.get("layer_norm") or "").lower() in ("inner", "outer"):
            layer_norm(f"{sp}._node_layer_norms")
            layer_norm(f"{sp}._edge_layer_norms")
    if supervised:
        dec = cfg["decoder"]
        reg = dec.get("regularization") or {}
        idx = 0
        for i in range(dec["num_layers"]):
            linear(f"decoder_mlp.mlp_layers.{idx}", L, L)
            idx += 2
            if reg.get("latent_normalization") in ("layer", "layer_norm"):
                layer_norm(f"decoder_mlp.mlp_layers.{idx}")
                idx += 1
            elif reg.get("latent_normalization") in ("batch", "batch_norm"):
                layer_norm(f"decoder_mlp.mlp_layers.{idx}")
                p[f"decoder_mlp.mlp_layers.{idx}.running_mean"] = 0.1 * torch.randn(L, generator=g)
                p[f"decoder_mlp.mlp_layers.{idx}.running_var"] = 1.0 + 0.1 * torch.rand(L, generator=g)
                idx += 1
            if reg.get("dropout", 0.0):
                idx += 1
        linear("readout", L, 1)
    return p
