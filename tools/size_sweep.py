#!/usr/bin/env python3
"""Forward time of the one-mesh configuration (L = 64, 20 steps) against the mesh size: the fixed cost of a launch pair.
usage: tools/size_sweep.py [latent] [steps] [grid sides ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amber_amd import csr as csr_mod, synthetic as syn  # noqa: E402
from amber_amd.architecture.supervised_mpn import SupervisedMPN  # noqa: E402
from amber_amd.config import default_architecture  # noqa: E402

latent = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[3:]] or [4, 16, 36, 72, 102, 144, 204, 288]
for nx in sizes:
    g = syn.grid_batch(num_graphs=1, nx=nx, ny=nx, in_edge_features=2).to(dev)
    cfg = default_architecture(latent_dimension=latent, num_steps=steps)
    model = SupervisedMPN(cfg, syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, 2))).to(dev).eval()
    g.amber_csr = csr_mod.build_csr(g.edge_index, g.x.shape[0])
    with torch.no_grad():
        for _ in range(10):
            model(g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 100
        for _ in range(n):
            model(g)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
    E = g.edge_index.shape[1]
    print(f"nx {nx:4d} N {g.x.shape[0]:7d} E {E:8d} tiles(edge) {(E + 31) // 32:6d}: {ms:.4f} ms / forward = {ms / steps * 1e3:.1f} us a step", flush=True)
