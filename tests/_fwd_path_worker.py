"""Helper of tests/test_gpu_precision_modes.py: one split-precision forward of a seeded model and batch, written to an
.npz (run as a script in a process of its own: the library reads AMBER_MPN_NO_FUSED_EMBED / _NO_FUSED_HEAD once)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from amber_amd import engine, synthetic as syn  # noqa: E402
from test_gpu_parity import _supervised  # noqa: E402

out, fe = sys.argv[1], int(sys.argv[2])
engine.default_precision = "f16x2"
dev = torch.device("cuda:0")
model, sd, cfg = _supervised(64, 3, 5, fe, dev, seed=11)
g = syn.grid_batch(num_graphs=2, nx=23, ny=17, in_node_features=5, in_edge_features=fe)
with torch.no_grad():
    pred = model(g.to(dev))
torch.cuda.synchronize(dev)
np.savez(out, pred=pred.cpu().numpy())
print("FWD_OK")
