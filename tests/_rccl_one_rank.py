"""Helper of tests/test_gpu_rccl_one_rank.py: a one-rank RCCL process group on cuda:0 (run as a script, own process)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amber_amd import sharding, synthetic as syn  # noqa: E402
from amber_amd.architecture.supervised_mpn import SupervisedMPN  # noqa: E402
from amber_amd.config import default_architecture  # noqa: E402

port = int(sys.argv[1])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
# the collectives the multi-rank paths use, on device tensors, ordered behind queued kernels on the current stream
flat = torch.arange(1 << 20, dtype=torch.float32, device=dev) * 2.0  # a flat gradient buffer
want = flat.clone()
dist.all_reduce(flat, op=dist.ReduceOp.SUM)
dist.barrier()
outs = [torch.empty(3, dtype=torch.float64, device=dev)]
dist.all_gather(outs, torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device=dev))
torch.cuda.synchronize(dev)
assert torch.equal(flat, want) and outs[0].tolist() == [1.0, 2.0, 3.0]
# graph-wise sharding through the group: one rank holds every graph, the gathered output is the whole-batch forward
torch.manual_seed(0)
cfg = default_architecture(latent_dimension=64, num_steps=3)
model = SupervisedMPN(cfg, syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, 2))).eval().to(dev)
g = syn.grid_batch(num_graphs=5, nx=9, ny=8).to(dev)
with torch.no_grad():
    whole = model(g)
    got = sharding.sharded_forward(model, g, 0, 1, gather=True)
assert torch.equal(whole, got)
# a training step's gradient sync on the group (world size 1: the all-reduce is skipped, the hook must not fail)
model.train()
model(g).sum().backward()
torch.cuda.synchronize(dev)
dist.destroy_process_group()
print("RCCL_OK")
