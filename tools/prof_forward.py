#!/usr/bin/env python3
"""Small driver for rocprofv3 counter passes: a few SupervisedMPN forwards on the BASELINE config-5 batch.
usage: rocprofv3 --pmc ... -- python3 tools/prof_forward.py [graphs] [forwards]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amber_amd import csr as csr_mod, synthetic as syn  # noqa: E402
from amber_amd.architecture.supervised_mpn import SupervisedMPN  # noqa: E402
from amber_amd.config import default_architecture  # noqa: E402

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
forwards = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = default_architecture(latent_dimension=128, num_steps=10)
model = SupervisedMPN(cfg, syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, 2))).eval().to(dev)
g = syn.grid_batch(num_graphs=graphs).to(dev)
g.amber_csr = csr_mod.build_csr(g.edge_index, int(g.x.shape[0]))
with torch.no_grad():
    for _ in range(forwards):
        out = model(g)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
