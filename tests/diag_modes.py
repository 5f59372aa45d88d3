#!/usr/bin/env python3
"""Developer diagnostic (not collected by pytest): error of non-default stack configurations against the oracle by graph
kind and depth.  usage (GPU box): python3 tests/diag_modes.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from amber_amd import engine, synthetic as syn  # noqa: E402
from amber_amd.architecture.supervised_mpn import SupervisedMPN  # noqa: E402
from amber_amd.config import AttrDict  # noqa: E402
from conftest import elem_ratio, rel_err  # noqa: E402
from oracle import mpn_oracle as orc  # noqa: E402

dev = torch.device("cuda:0")


def graph(kind):
    rng = np.random.default_rng(17)
    if kind == "grid":
        return syn.grid_batch(num_graphs=2, nx=40, ny=30)
    if kind == "hier_small":
        el, n, mask = syn.hierarchical_graph(18, 14, 2, 2, rng)
        return syn.make_batch([el], [n], 4, 2, seed=5, mask_outputs=[mask])
    parts = [syn.hierarchical_graph(47, 39, 3, 2, rng), syn.hierarchical_graph(31, 28, 2, 2, rng)]
    return syn.make_batch([p[0] for p in parts], [p[1] for p in parts], 4, 2, mask_outputs=[p[2] for p in parts])


def run(latent, steps, act, residual, layer_norm, kind, perturb=True):
    torch.manual_seed(latent + steps)
    cfg = {"latent_dimension": latent, "create_graph_copy": True, "assert_graph_shapes": False, "edge_dropout": 0.0, "embedding": None,
           "stack": {"layer_norm": layer_norm, "num_steps": steps, "residual_connections": residual,
                     "mlp": {"activation_function": act, "num_layers": 2}},
           "decoder": {"activation_function": "tanh" if act == "tanh" else "leakyrelu", "num_layers": 2,
                       "regularization": {"dropout": 0, "spectral_norm": False, "latent_normalization": None}}}
    g = graph(kind)
    model = SupervisedMPN(AttrDict(cfg), syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, 2)))
    if perturb:
        with torch.no_grad():
            for name, prm in model.named_parameters():
                if "layer_norm" in name:
                    prm.add_(0.1 * torch.randn(prm.shape))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ocfg = dict(cfg, in_node_features=4, in_edge_features=2)
    want = orc.supervised_mpn_forward(sd, ocfg, g.x, g.edge_attr, g.edge_index, g.mask_output)
    wx, we = orc.mpn_forward(sd, ocfg, g.x, g.edge_attr, g.edge_index, prefix="mpn.")
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    want64 = orc.supervised_mpn_forward(sd64, ocfg, g.x.double(), g.edge_attr.double(), g.edge_index, g.mask_output)
    model = model.to(dev).eval()
    gd = g.to(dev)
    res = []
    for prec in ("fp32", "f16x2"):
        engine.default_precision = prec
        with torch.no_grad():
            got = model(gd).cpu()
            nodes, edges, _ = model.mpn(gd)
        res.append(f"{prec}: out {rel_err(got, want):.1e} (ratio {elem_ratio(got, want):.1f}) x {rel_err(nodes['node'].cpu(), wx):.1e} "
                   f"e {rel_err(edges['edges']['edge_attr'].cpu(), we):.1e}")
    print(f"L{latent} s{steps} {act} res={residual} ln={layer_norm} {kind} perturb={perturb} N={g.x.shape[0]} E={g.edge_index.shape[1]} "
          f"max|out| {float(want.abs().max()):.2e} oracle32-vs-64 {rel_err(want.double(), want64):.1e} | " + " | ".join(res), flush=True)


for kind in ("grid", "hier_small", "hier_mid"):
    run(64, 6, "tanh", None, None, kind)
for steps in (1, 2, 4, 6):
    run(64, steps, "tanh", None, None, "hier_mid")
run(64, 6, "tanh", "inner", "inner", "hier_mid")
run(64, 6, "leakyrelu", "inner", "inner", "hier_mid")
run(64, 6, "leakyrelu", None, None, "hier_mid")
run(64, 6, "leakyrelu", "outer", "outer", "hier_mid", perturb=False)
run(128, 4, "leakyrelu", "outer", "outer", "hier_mid")
run(128, 4, "leakyrelu", "inner", "inner", "hier_mid")
