#!/usr/bin/env python3
"""Developer diagnostic: run-to-run differences of the edge / node latents of one configuration (race hunting)."""
import os
import sys

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
latent, steps, act = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
fe = int(sys.argv[4]) if len(sys.argv) > 4 else 2
torch.manual_seed(latent + steps)
cfg = {"latent_dimension": latent, "create_graph_copy": True, "assert_graph_shapes": False, "edge_dropout": 0.0, "embedding": None,
       "stack": {"layer_norm": None, "num_steps": steps, "residual_connections": None, "mlp": {"activation_function": act, "num_layers": 2}},
       "decoder": {"activation_function": "leakyrelu", "num_layers": 2,
                   "regularization": {"dropout": 0, "spectral_norm": False, "latent_normalization": None}}}
g = syn.grid_batch(num_graphs=2, nx=40, ny=30, in_edge_features=fe)
model = SupervisedMPN(AttrDict(cfg), syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, fe)))
sd = {k: v.clone() for k, v in model.state_dict().items()}
ocfg = dict(cfg, in_node_features=4, in_edge_features=fe)
wx, we = orc.mpn_forward(sd, ocfg, g.x, g.edge_attr, g.edge_index, prefix="mpn.")
model = model.to(dev).eval()
gd = g.to(dev)
csr = csr_mod.build_csr(gd.edge_index, g.x.shape[0])
inv = torch.empty_like(csr.perm.long().cpu())
inv[csr.perm.long().cpu()] = torch.arange(inv.numel())
runs = []
with torch.no_grad():
    for _ in range(12):
        nodes, edges, _ = model.mpn(gd)
        runs.append((nodes["node"].cpu(), edges["edges"]["edge_attr"].cpu()))
tag = f"L{latent} s{steps} {act} Fe={fe} NO_FUSED_EMBED={os.environ.get('AMBER_MPN_NO_FUSED_EMBED')}"
for k, (gx, ge) in enumerate(runs):
    bad = ((ge - we).abs() > 1e-4 * float(we.abs().max())).any(1).nonzero().flatten()
    pos = sorted(int(inv[b]) for b in bad.tolist())
    print(f"{tag} run {k}: e err {rel_err(ge, we):.1e} x err {rel_err(gx, wx):.1e} bad rows (csr pos%32): {[p % 32 for p in pos][:24]} tiles {[p // 32 for p in pos][:12]}")
print(tag, "runs bit-equal:", all(torch.equal(runs[0][1], r[1]) and torch.equal(runs[0][0], r[0]) for r in runs))
