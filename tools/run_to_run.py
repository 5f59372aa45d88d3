#!/usr/bin/env python3
"""Developer diagnostic: run-to-run differences of one configuration over several step counts (race hunting).
usage: tools/run_to_run.py latent act residual layer_norm max_steps runs [precision]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from amber_amd import csr as csr_mod, engine, synthetic as syn  # noqa: E402
from amber_amd.architecture.supervised_mpn import SupervisedMPN  # noqa: E402
from amber_amd.config import AttrDict  # noqa: E402

dev = torch.device("cuda:0")
latent, act = int(sys.argv[1]), sys.argv[2]
res = None if sys.argv[3] == "None" else sys.argv[3]
ln = None if sys.argv[4] == "None" else sys.argv[4]
max_steps, nruns = int(sys.argv[5]), int(sys.argv[6])
engine.default_precision = sys.argv[7] if len(sys.argv) > 7 else "f16x2"
fe = 2
g = syn.grid_batch(num_graphs=2, nx=40, ny=30, in_edge_features=fe)
gd = g.to(dev)
csr = csr_mod.build_csr(gd.edge_index, g.x.shape[0])
perm = csr.perm.long().cpu()
inv = torch.empty_like(perm)
inv[perm] = torch.arange(inv.numel())
src, dst = g.edge_index[0], g.edge_index[1]
for steps in range(1, max_steps + 1):
    torch.manual_seed(latent + steps)
    cfg = {"latent_dimension": latent, "create_graph_copy": True, "assert_graph_shapes": False, "edge_dropout": 0.0, "embedding": None,
           "stack": {"layer_norm": ln, "num_steps": steps, "residual_connections": res, "mlp": {"activation_function": act, "num_layers": 2}},
           "decoder": {"activation_function": "leakyrelu", "num_layers": 2,
                       "regularization": {"dropout": 0, "spectral_norm": False, "latent_normalization": None}}}
    model = SupervisedMPN(AttrDict(cfg), syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, fe))).to(dev).eval()
    runs = []
    with torch.no_grad():
        for _ in range(nruns):
            nodes, edges, _ = model.mpn(gd)
            runs.append((nodes["node"].cpu(), edges["edges"]["edge_attr"].cpu()))
    # majority vote per element is overkill: take the most common whole result as the truth
    keys = [hash((r[0].numpy().tobytes(), r[1].numpy().tobytes())) for r in runs]
    truth = runs[max(range(nruns), key=lambda i: keys.count(keys[i]))]
    odd = [i for i in range(nruns) if keys[i] != keys[keys.index(max(keys, key=keys.count))]]
    print(f"L{latent} {act} res={res} ln={ln} steps={steps}: {len(set(keys))} distinct results over {nruns} runs, odd runs {odd}", flush=True)
    for i in odd[:6]:
        gx, ge = runs[i]
        be = (ge != truth[1]).any(1).nonzero().flatten().tolist()
        bx = (gx != truth[0]).any(1).nonzero().flatten().tolist()
        pos = sorted(int(inv[b]) for b in be)
        emax = float((ge - truth[1]).abs().max()) / float(truth[1].abs().max())
        xmax = float((gx - truth[0]).abs().max()) / float(truth[0].abs().max())
        print(f"   run {i}: {len(be)} edge rows differ (max {emax:.1e}), {len(bx)} node rows differ (max {xmax:.1e}); csr pos%32 {[p % 32 for p in pos][:16]} tiles {[p // 32 for p in pos][:16]}"
              f" srcs {sorted(set(int(src[b]) for b in be))[:8]} dsts {sorted(set(int(dst[b]) for b in be))[:8]} nodes {bx[:8]}", flush=True)
