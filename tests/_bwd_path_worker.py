"""Helper of tests/test_gpu_backward_paths.py: parameter gradients of one training step of AMBER's default configuration,
written to an .npz (run as a script in a process of its own: the library reads its environment switches once)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amber_amd import synthetic as syn  # noqa: E402
from amber_amd.architecture.supervised_mpn import SupervisedMPN  # noqa: E402
from amber_amd.config import default_architecture  # noqa: E402

out, latent = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
torch.manual_seed(7)
cfg = default_architecture(latent_dimension=latent, num_steps=3)
model = SupervisedMPN(cfg, syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, 2))).to(dev).train()
g = syn.grid_batch(num_graphs=3, nx=23, ny=17).to(dev)  # 1,173 nodes: ragged last tiles, several tiles a wave at 32 rows
pred = model(g)
weight = torch.linspace(-1.0, 1.0, pred.numel(), device=dev).reshape(pred.shape)
(pred * weight).sum().backward()
torch.cuda.synchronize(dev)
np.savez(out, pred=pred.detach().cpu().numpy(), **{n.replace(".", "/"): p.grad.cpu().numpy() for n, p in model.named_parameters() if p.grad is not None})
print("BWD_OK")
