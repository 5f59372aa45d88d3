#!/usr/bin/env python3
"""Developer diagnostic: WHERE (CSR tile / row / feature) the edge latents of a non-default configuration differ from the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from amber_amd import csr as csr_mod, engine, synthetic as syn  # noqa: E402
from amber_amd.architecture.supervised_mpn import SupervisedMPN  # noqa: E402
from amber_amd.config import AttrDict  # noqa: E402
from conftest import rel_err  # noqa: E402
from oracle import mpn_oracle as orc  # noqa: E402

dev = torch.device("cuda:0")
engine.default_precision = "f16x2"


def run(latent, steps, act, residual, layer_norm, g, tag):
    torch.manual_seed(latent + steps)
    cfg = {"latent_dimension": latent, "create_graph_copy": True, "assert_graph_shapes": False, "edge_dropout": 0.0, "embedding": None,
           "stack": {"layer_norm": layer_norm, "num_steps": steps, "residual_connections": residual,
                     "mlp": {"activation_function": act, "num_layers": 2}},
           "decoder": {"activation_function": "tanh" if act == "tanh" else "leakyrelu", "num_layers": 2,
                       "regularization": {"dropout": 0, "spectral_norm": False, "latent_normalization": None}}}
    model = SupervisedMPN(AttrDict(cfg), syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, 2)))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ocfg = dict(cfg, in_node_features=4, in_edge_features=2)
    wx, we = orc.mpn_forward(sd, ocfg, g.x, g.edge_attr, g.edge_index, prefix="mpn.")
    model = model.to(dev).eval()
    gd = g.to(dev)
    with torch.no_grad():
        nodes, edges, _ = model.mpn(gd)
    ge = edges["edges"]["edge_attr"].cpu()
    gx = nodes["node"].cpu()
    csr = csr_mod.build_csr(gd.edge_index, g.x.shape[0])
    perm = csr.perm.long().cpu()
    err = (ge - we).abs()[perm]  # CSR order
    scale = float(we.abs().max())
    bad = err > 1e-4 * scale
    rows = bad.any(1).nonzero().flatten()
    tiles = torch.unique(rows // 32)
    cols = bad.any(0).nonzero().flatten()
    print(f"{tag}: L{latent} s{steps} {act} {residual} {layer_norm} N={g.x.shape[0]} E={g.edge_index.shape[1]} tiles={(g.edge_index.shape[1] + 31) // 32} "
          f"e err {rel_err(ge, we):.1e} x err {rel_err(gx, wx):.1e} bad rows {rows.numel()} bad tiles {tiles.numel()} bad cols {cols.numel()}")
    if rows.numel():
        print("   bad tiles:", tiles[:40].tolist())
        print("   rows in tile of first bad rows:", (rows[:40] % 32).tolist())
        print("   bad cols:", cols[:64].tolist())
        r = int(rows[0])
        print("   got ", ge[perm][r, :8].tolist())
        print("   want", we[perm][r, :8].tolist())
    if rows.numel():
        gp, wp = ge[perm], we[perm]
        src_s, dst_s = csr.src_sorted.cpu(), csr.dst_sorted.cpu()
        for r in rows[:8].tolist():
            d = (wp - gp[r]).abs().max(1).values
            q = int(d.argmin())
            print(f"   row {r} (tile {r // 32} pos {r % 32}, src {int(src_s[r])} dst {int(dst_s[r])}): nearest oracle row {q} (pos {q % 32}, src {int(src_s[q])} "
                  f"dst {int(dst_s[q])}) dist {float(d[q]):.2e}; own dist {float(d[r]):.2e}; prev-row dst {int(dst_s[r - 1])} next-row dst {int(dst_s[min(r + 1, len(dst_s) - 1)])}")
    xerr = (gx - wx).abs()
    badx = (xerr > 1e-4 * float(wx.abs().max())).any(1).nonzero().flatten()
    print(f"   bad node rows {badx.numel()}: {badx[:20].tolist()}")


run(64, 1, "tanh", None, None, syn.grid_batch(num_graphs=2, nx=20, ny=20), "grid20x20")
run(64, 1, "tanh", None, None, syn.grid_batch(num_graphs=2, nx=20, ny=20), "grid20x20-again")
run(128, 1, "leakyrelu", "outer", "outer", syn.grid_batch(num_graphs=2, nx=40, ny=30), "grid40x30")
