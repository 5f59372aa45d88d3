#!/usr/bin/env python3
"""Developer diagnostic (not collected by pytest; lives under tests/ because it uses the oracle): error pattern of one
block application vs the oracle on a golden fixture, and of the partial sums the edge kernel leaves in the workspace.
usage (GPU box): python3 tests/diag_block.py [fixture]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from amber_amd import csr as csr_mod, engine  # noqa: E402
from oracle import mpn_oracle as orc  # noqa: E402
import test_gpu_parity as T  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "debug_L64_s2"
dev = torch.device("cuda:0")
cfg, params, d = T.load_golden(name)
model = T._model_from_golden(cfg, params, d, dev)
mpn = model.mpn
csr = csr_mod.build_csr(d["edge_index"].to(dev), int(d["x"].shape[0]))
desc = engine.make_desc(**mpn.desc_kwargs())
flat, packed = engine.PackedParams().get(desc, mpn.param_list())
_, _, steps = orc.mpn_forward(params, cfg, d["x"], d["edge_attr"], d["edge_index"], prefix="mpn.", collect_steps=True)
x, e = engine.embed(desc, flat, packed, d["x"].to(dev), d["edge_attr"].to(dev), csr)
perm = csr.perm.long().cpu()
engine.block(desc, flat, packed, 0, x, e, csr)
torch.set_printoptions(linewidth=200, precision=4)
for label, got, want in (("edge", e.cpu(), steps[1][1][perm]), ("node", x.cpu(), steps[1][0])):
    err = (got - want).abs()
    scale = want.abs().max()
    print(f"{label}: shape {tuple(got.shape)} max rel err {float(err.max() / scale):.3e}")
    bad = err > 1e-4 * scale
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print(f"  bad rows {rows.numel()} / {got.shape[0]}: {rows[:64].tolist()}")
    print(f"  bad cols {cols.numel()} / {got.shape[1]}: {cols[:64].tolist()}")
    if rows.numel():
        r = int(rows[0])
        print(f"  row {r} got  {got[r, :16].tolist()}")
        print(f"  row {r} want {want[r, :16].tolist()}")
print("rowptr", csr.rowptr[:40].tolist())

# ---- partial sums left in the workspace by the edge kernel of a fresh block application ----
import ctypes as C  # noqa: E402
from amber_amd import _lib  # noqa: E402
N, E, L = csr.num_nodes, csr.num_edges, desc.latent
x2, e2 = engine.embed(desc, flat, packed, d["x"].to(dev), d["edge_attr"].to(dev), csr)
ws = engine.workspace(desc, N, E, dev)
engine.block(desc, flat, packed, 0, x2, e2, csr, ws=ws)
torch.cuda.synchronize()
a256 = lambda v: (v + 255) & ~255
base = (-ws.data_ptr()) % 256
nb, tb = a256(4 * N * L), a256(4 * ((E + 15) // 16) * L)
f = ws[base:].view(torch.float32) if (ws.numel() - base) % 4 == 0 else ws[base:base + (ws.numel() - base) // 4 * 4].view(torch.float32)
agg = f[3 * nb // 4: 3 * nb // 4 + N * L].view(N, L).cpu()
head = f[4 * nb // 4: 4 * nb // 4 + ((E + 31) // 32) * L].view(-1, L).cpu()
tail = f[(4 * nb + tb) // 4: (4 * nb + tb) // 4 + ((E + 31) // 32) * L].view(-1, L).cpu()
ew = steps[1][1][perm]
rp = csr.rowptr.cpu().tolist()
for node in range(N):
    r0, r1 = rp[node], rp[node + 1]
    if r1 == r0:
        continue
    t0, t1 = r0 // 32, (r1 - 1) // 32
    if t0 == t1:
        err = float((agg[node] - ew[r0:r1].sum(0)).abs().max())
        if err > 1e-4:
            print(f"node {node} rows [{r0},{r1}) agg err {err:.3e}")
    else:
        wt = ew[r0:(t0 + 1) * 32].sum(0)
        print(f"node {node} rows [{r0},{r1}) tiles {t0}..{t1}: tail[{t0}] err {float((tail[t0] - wt).abs().max()):.3e}", end="")
        for t in range(t0 + 1, t1 + 1):
            wh = ew[t * 32:min(r1, (t + 1) * 32)].sum(0)
            print(f"  head[{t}] err {float((head[t] - wh).abs().max()):.3e}", end="")
        print()
