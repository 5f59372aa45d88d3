"""Two ranks on ONE GPU (gloo rendezvous; RCCL cannot place two ranks on one device): the multi-rank code paths of the
library on real kernels.  Run by tests/test_gpu_two_ranks.py under torch.distributed.run; prints 'TWO-RANK OK'.

1. graph-wise sharding (amber_amd/sharding.py): the per-rank forwards, gathered in graph order, equal the single-process
   forward of the whole batch (1e-5: tiles are cut differently, so the split-precision tile scales differ).
2. training (engine.MpnFunction + _sync_flat_gradient): the synchronised gradient on every rank is bit-equal to the mean
   of the two ranks' local gradients."""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amber_amd import engine, sharding, synthetic as syn  # noqa: E402
from amber_amd.architecture.supervised_mpn import SupervisedMPN  # noqa: E402
from amber_amd.config import default_architecture  # noqa: E402

faulthandler.dump_traceback_later(90, exit=True)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)

torch.manual_seed(0)
cfg = default_architecture(latent_dimension=64, num_steps=3)
cfg["edge_dropout"] = 0.0  # training mode must be deterministic for the bit-equality below
model = SupervisedMPN(cfg, syn.GraphBatch(x=torch.zeros(1, 4), edge_attr=torch.zeros(1, 2))).to(dev)

# 1. sharded inference
model.eval()
full = syn.grid_batch(num_graphs=5, nx=23, ny=17, seed=7, hetero=False).to(dev)
with torch.no_grad():
    want = model(full)
    got = sharding.sharded_forward(lambda b: model(b), full, rank, world, gather=True)
assert got.shape == want.shape, (got.shape, want.shape)
err = float((got - want).abs().max() / want.abs().max())
assert err <= 1e-5, err

# 2. synchronised gradients
model.train()
g = syn.grid_batch(num_graphs=2, nx=19 + 3 * rank, ny=16, seed=100 + rank).to(dev)
target = torch.randn(g.x.shape[0], 1, generator=torch.Generator().manual_seed(5 + rank)).to(dev)


def grads(sync):
    engine.grad_sync = sync
    model.zero_grad(set_to_none=True)
    torch.nn.functional.mse_loss(model(g), target).backward()
    return torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()


local = grads(None)
both = [torch.empty_like(local) for _ in range(world)]
dist.all_gather(both, local)
mean = (both[0] + both[1]) / world
synced = grads("mean")
assert torch.equal(synced, mean), float((synced - mean).abs().max())
summed = grads("sum")
assert torch.equal(summed, both[0] + both[1])
assert float(local.abs().max()) > 0 and not torch.equal(both[0], both[1])
dist.barrier()
if rank == 0:
    print(f"TWO-RANK OK sharded forward rel err {err:.1e}, {local.numel()} gradient entries bit-equal to the ranks' mean", flush=True)
dist.destroy_process_group()
