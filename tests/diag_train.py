"""Developer diagnostic (not collected by pytest): backward workspace contents per precision mode."""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from amber_amd import engine, synthetic as syn
import test_gpu_training as T
from oracle import mpn_oracle as orc
from conftest import rel_err
dev = torch.device("cuda:0")
latent, steps, layers = 128, 1, 3
rng = np.random.default_rng(latent + 2)
el, n, mask = syn.hierarchical_graph(30, 24, 2, 2, rng)
g = syn.make_batch([el], [n], 4, 2, seed=5, mask_outputs=[mask])
cfg = T._cfg(latent, steps, layers=layers)
params = orc.init_params(cfg, seed=3)
d = dict(x=g.x, edge_attr=g.edge_attr, edge_index=g.edge_index, batch=g.batch, mask_output=g.mask_output)
N, E, L = g.x.shape[0], g.edge_index.shape[1], latent
n_out = int(g.mask_output.sum())
grad_out = torch.randn(n_out, 1, generator=torch.Generator().manual_seed(9))
want_out, want = T._oracle_grads(cfg, params, g.x, g.edge_attr, g.edge_index, g.mask_output, grad_out)
engine.debug_keep_backward_workspace = []
for prec in ("fp32", "f16x2"):
    engine.default_precision = prec
    model = T._model(cfg, params, d, dev).train()
    out = model(T._batch(d, dev))
    (out * grad_out.to(dev)).sum().backward()
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        e = rel_err(p.grad.cpu(), want[k])
        if e > 1e-5: print("  ", prec, k, f"{e:.2e}")
a, b = [w.cpu() for w in engine.debug_keep_backward_workspace]
off0 = (256 - a.data_ptr() % 256) % 256  # both were 256-aligned on the device; the library aligns the base itself
a, b = a.view(torch.float32), b.view(torch.float32)
al = lambda x: (x + 255) // 256 * 256
nb, eb = al(N * L * 4) // 4, al(E * L * 4) // 4
names = ["dX", "M", "DM", "Ps", "Pd", "dPs", "dPd"] + [f"{k}{l}" for l in range(layers) for k in ("AN", "DHN")]
off = 0
for nm in names:
    x, y = a[off:off + N * L].view(N, L), b[off:off + N * L].view(N, L)
    dd = (x - y).abs().amax(1)
    print(nm, "maxdiff", float(dd.max()), "scale", float(x.abs().max()), "bad rows", (dd > 1e-4 * float(x.abs().max() + 1e-30)).nonzero().flatten()[:10].tolist())
    off += nb
for nm in ["dE"] + [f"{k}{l}" for l in range(layers) for k in ("AE", "DHE")]:
    x, y = a[off:off + E * L].view(E, L), b[off:off + E * L].view(E, L)
    dd = (x - y).abs().amax(1)
    bad = (dd > 1e-4 * float(x.abs().max() + 1e-30)).nonzero().flatten()
    print(nm, "maxdiff", float(dd.max()), "scale", float(x.abs().max()), "bad rows", len(bad), bad[:10].tolist())
    off += eb
