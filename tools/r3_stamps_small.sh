#!/bin/bash
export AMBER_MPN_LIB="$PWD/amber_amd/csrc/variants/libamber_stamps.so"
for nx in 4 144; do
echo "== nx $nx"
timeout -k 10 120 python - $nx <<'PY' 2>&1 | grep "stamps" | sort | uniq -c | sort -rn | head -8
import sys, torch
sys.path.insert(0,'.')
from amber_amd import csr as csr_mod, synthetic as syn
from amber_amd.architecture.supervised_mpn import SupervisedMPN
from amber_amd.config import default_architecture
nx=int(sys.argv[1])
dev=torch.device('cuda:0')
g=syn.grid_batch(num_graphs=1,nx=nx,ny=nx,in_edge_features=2).to(dev)
model=SupervisedMPN(default_architecture(latent_dimension=64,num_steps=3), syn.GraphBatch(x=torch.zeros(1,4),edge_attr=torch.zeros(1,2))).to(dev).eval()
g.amber_csr=csr_mod.build_csr(g.edge_index,g.x.shape[0])
with torch.no_grad():
    for _ in range(3): model(g)
torch.cuda.synchronize()
PY
done
