"""Randomised parity sweep: random network configurations x random graph batches, HIP path (C ABI) against the CPU
oracle.  Not collected by pytest (the test_*.py files hold the fixed cases; it lives in tests/ because it calls the
oracle).  Run on the GPU box after kernel changes:
    python -u tests/fuzz_parity.py --cases 60 --seed 1 [--precision f16x2|fp32] [--grads] [--keep-going]
One line per case; exits non-zero on a case above the 1e-5 bar (1e-4 for gradients).  The bar widens to 4x the case's
own float32 rounding noise (oracle in float32 against the oracle in float64) where that is larger: random networks
can be ill-conditioned.  Checks the readout, the node latents and the edge latents, plus bit-equal reruns."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from amber_amd import engine, synthetic as syn  # noqa: E402
from amber_amd.architecture.supervised_mpn import SupervisedMPN  # noqa: E402
from amber_amd.config import AttrDict  # noqa: E402
from oracle import mpn_oracle as orc  # noqa: E402

TOL = 1e-5


def rel_err(a, b):
    scale = float(b.abs().max())
    return float((a - b).abs().max()) / (scale if scale > 0 else 1.0)


def random_cfg(rng):
    latent = int(rng.choice([32, 64, 128, 256], p=[0.25, 0.3, 0.35, 0.1]))
    act = str(rng.choice(["leakyrelu", "relu", "tanh"], p=[0.5, 0.3, 0.2]))  # LatentMLP's zoo (latent_mlp.py:76-83)
    mode = [None, "inner", "outer"]
    emb = None
    if rng.random() < 0.3:
        emb = {"mlp": {"activation_function": str(rng.choice(["leakyrelu", "relu", "tanh"])), "num_layers": int(rng.integers(1, 3))}}
    cfg = {"latent_dimension": latent, "create_graph_copy": True, "assert_graph_shapes": False, "edge_dropout": 0.1,
           "embedding": emb,
           "stack": {"layer_norm": mode[int(rng.choice(3, p=[0.15, 0.6, 0.25]))], "num_steps": int(rng.integers(1, 5)),
                     "num_step_repeats": int(rng.choice([1, 1, 2])),
                     "residual_connections": mode[int(rng.choice(3, p=[0.15, 0.6, 0.25]))],
                     "mlp": {"activation_function": act, "num_layers": int(rng.choice([1, 2, 2, 2, 3]))}},
           "decoder": {"activation_function": str(rng.choice(["leakyrelu", "relu", "tanh", "elu", "silu", "gelu", "mish"])), "num_layers": int(rng.integers(1, 4)),
                       "regularization": {"dropout": 0, "spectral_norm": bool(rng.random() < 0.25),
                                          "latent_normalization": [None, None, "layer", "batch"][int(rng.integers(0, 4))]}}}
    return cfg


def random_batch(rng, fn, fe):
    kind = int(rng.integers(0, 4))
    if kind == 0:  # mesh-like grids (the production shape)
        g = syn.grid_batch(num_graphs=int(rng.integers(1, 6)), nx=int(rng.integers(3, 60)), ny=int(rng.integers(3, 60)),
                           in_node_features=fn, in_edge_features=fe, seed=int(rng.integers(1 << 30)))
    elif kind == 1:  # hierarchical graphs with hub rows and an output mask
        els, counts, masks = [], [], []
        for _ in range(int(rng.integers(1, 4))):
            el, n, mask = syn.hierarchical_graph(int(rng.integers(4, 40)), int(rng.integers(4, 40)), int(rng.integers(1, 4)),
                                                 int(rng.integers(1, 4)), rng)
            els.append(el), counts.append(n), masks.append(mask)
        g = syn.make_batch(els, counts, fn, fe, seed=int(rng.integers(1 << 30)), mask_outputs=masks)
    elif kind == 2:  # random multigraph with isolated nodes and a random mask
        n = int(rng.integers(1, 3000))
        e = int(rng.integers(0, 8 * n + 1))
        mask = rng.random(n) < 0.6
        if not mask.any():
            mask[0] = True
        g = syn.make_batch([syn.random_graph(n, e, rng, isolated=int(rng.integers(0, n // 4 + 1)))], [n], fn, fe,
                           seed=int(rng.integers(1 << 30)), mask_outputs=[mask])
    else:  # few nodes, very long rows (segments spanning many tiles)
        n = int(rng.integers(1, 40))
        e = int(rng.integers(1, 6000))
        g = syn.make_batch([syn.random_graph(n, e, rng)], [n], fn, fe, seed=int(rng.integers(1 << 30)))
    return g, kind


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--precision", default="both", choices=["both", "f16x2", "fp32"])
    ap.add_argument("--grads", action="store_true", help="training path: parameter gradients against autograd through the "
                    "oracle (smooth activations only: a relu kink within rounding of zero flips between implementations)")
    ap.add_argument("--keep-going", action="store_true", help="report every failing case instead of stopping at the first")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(args.seed)
    precisions = ["f16x2", "fp32"] if args.precision == "both" else [args.precision]
    worst, failed = 0.0, 0
    t0 = time.time()
    for case in range(args.cases):
        cfg = random_cfg(rng)
        if args.grads:  # what the training path supports (everything but decoder BatchNorm), smooth activations only
            cfg["stack"]["mlp"]["activation_function"] = "tanh"
            if cfg["decoder"]["activation_function"] in ("relu", "leakyrelu"):
                cfg["decoder"]["activation_function"] = "tanh"
            if cfg["decoder"]["regularization"]["latent_normalization"] == "batch":
                cfg["decoder"]["regularization"]["latent_normalization"] = "layer"
            cfg["edge_dropout"] = 0.0
            if cfg["embedding"] is not None:
                cfg["embedding"]["mlp"]["activation_function"] = "tanh"
        fn, fe = int(rng.choice([0, 1, 3, 4, 7, 12])), int(rng.choice([0, 1, 2, 5, 9]))
        g, kind = random_batch(rng, fn, fe)
        torch.manual_seed(int(rng.integers(1 << 30)))
        example = syn.GraphBatch(x=torch.zeros(1, fn), edge_attr=torch.zeros(1, fe))
        model = SupervisedMPN(AttrDict(cfg), example)
        with torch.no_grad():
            for name, p in model.named_parameters():
                if "layer_norm" in name or ("decoder" in name and p.dim() == 1):
                    p.add_(0.1 * torch.randn(p.shape))
            for name, b in model.named_buffers():
                if "running_mean" in name:
                    b.add_(0.2 * torch.randn(b.shape))
                if "running_var" in name:
                    b.mul_(0.5 + torch.rand(b.shape))
            # spectral norm starts from random u, v (sigma = u.Wv can be anything): settle them as training would
            for m in model.decoder_mlp.mlp_layers:
                for hook in getattr(m, "_forward_pre_hooks", {}).values():
                    if type(hook).__name__ == "SpectralNorm":
                        for _ in range(10):
                            hook.compute_weight(m, do_power_iteration=True)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        ocfg = dict(cfg, in_node_features=fn, in_edge_features=fe)
        want = orc.supervised_mpn_forward(sd, ocfg, g.x, g.edge_attr, g.edge_index, g.mask_output)
        wx, we = orc.mpn_forward(sd, ocfg, g.x, g.edge_attr, g.edge_index, prefix="mpn.")
        # rounding noise of the case itself: the float32 oracle against the same oracle in float64.  Random networks
        # can be ill-conditioned (a readout sum that cancels, LayerNorm of collapsed latents without residuals), and
        # then any float32 evaluation order differs by more than 1e-5; the bar widens to 4x that noise.
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        want64 = orc.supervised_mpn_forward(sd64, ocfg, g.x.double(), g.edge_attr.double(), g.edge_index, g.mask_output)
        wx64, we64 = orc.mpn_forward(sd64, ocfg, g.x.double(), g.edge_attr.double(), g.edge_index, prefix="mpn.")
        noise = (rel_err(want.double(), want64), rel_err(wx.double(), wx64), rel_err(we.double(), we64) if we.numel() else 0.0)
        bars = [max(TOL, 4 * v) for v in noise]
        if args.grads:
            grad_out = torch.randn(want.shape)

            def oracle_grads(params, dt):
                p = {k: (v.to(dt) if v.is_floating_point() else v).clone().requires_grad_(
                    v.is_floating_point() and not k.endswith(("weight_u", "weight_v"))) for k, v in params.items()}
                o = orc.supervised_mpn_forward(p, ocfg, g.x.to(dt), g.edge_attr.to(dt), g.edge_index, g.mask_output)
                (o * grad_out.to(dt)).sum().backward()
                # None: the output does not depend on the parameter (outer LayerNorm of the last block's edges) = zero
                return {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items() if v.requires_grad}

            g32, g64 = oracle_grads(sd, torch.float32), oracle_grads(sd, torch.float64)
            gnoise = max(rel_err(g32[k].double(), g64[k]) for k in g32)
            # training mode, except with spectral norm: there a training-mode forward first advances the power iteration
            # (as the reference's does), which the oracle's snapshot of the state_dict does not follow
            model = model.to(dev).train(not cfg["decoder"]["regularization"]["spectral_norm"])
            gb = g.to(dev)
            for prec in precisions:
                engine.default_precision = prec
                model._packed = engine.PackedParams()
                model.zero_grad(set_to_none=True)
                out = model(gb)
                (out * grad_out.to(dev)).sum().backward()
                errs = {k: rel_err(p.grad.cpu(), g32[k]) for k, p in model.named_parameters()}
                # the readout bias gradient is the plain sum of grad_out: one number, which can cancel to nothing - measure
                # its error against the size of the terms
                errs["readout.bias"] = float((model.readout.bias.grad.cpu() - g32["readout.bias"]).abs().max()) / float(grad_out.abs().sum())
                kmax = max(errs, key=errs.get)
                ferr = rel_err(out.detach().cpu(), want)
                # gradients are sums over all rows in another order than the oracle's: up to ~7x the oracle's own float32 noise seen
                ok = errs[kmax] <= max(1e-4, 10 * gnoise) and ferr <= bars[0]
                worst = max(worst, errs[kmax])
                st = cfg["stack"]
                print(f"case {case:3d} {prec:5s} L={cfg['latent_dimension']:3d} steps={st['num_steps']}x{st['num_step_repeats']} "
                      f"nl={st['mlp']['num_layers']} res={st['residual_connections']} ln={st['layer_norm']} "
                      f"emb={'mlp' if cfg['embedding'] else 'lin'} dec={cfg['decoder']['num_layers']}/{cfg['decoder']['activation_function']}"
                      f"{'/sn' if cfg['decoder']['regularization']['spectral_norm'] else ''} "
                      f"F=({fn},{fe}) kind={kind} N={g.x.shape[0]} E={g.edge_attr.shape[0]} fwd={ferr:.1e} grad={errs[kmax]:.1e} "
                      f"({kmax}) noise={gnoise:.1e} {'ok' if ok else 'FAIL'}", flush=True)
                if not ok:
                    failed += 1
                    if not args.keep_going:
                        print(f"FAILED at case {case} (seed {args.seed})", flush=True)
                        sys.exit(1)
            continue
        model = model.to(dev).eval()
        gb = g.to(dev)
        for prec in precisions:
            engine.default_precision = prec
            model._packed = engine.PackedParams()
            with torch.no_grad():
                out = model(gb)
                out2 = model(gb)
                nodes, edges, _ = model.mpn(gb)
            errs = (rel_err(out.cpu(), want), rel_err(nodes["node"].cpu(), wx),
                    rel_err(edges["edges"]["edge_attr"].cpu(), we) if we.numel() else 0.0)
            ok = all(v <= bar for v, bar in zip(errs, bars)) and torch.equal(out, out2) and out.shape == want.shape
            worst = max(worst, *errs)
            st = cfg["stack"]
            print(f"case {case:3d} {prec:5s} L={cfg['latent_dimension']:3d} steps={st['num_steps']}x{st['num_step_repeats']} "
                  f"nl={st['mlp']['num_layers']} act={st['mlp']['activation_function']:9s} res={st['residual_connections']} "
                  f"ln={st['layer_norm']} emb={'mlp' if cfg['embedding'] else 'lin'} dec={cfg['decoder']['num_layers']}/"
                  f"{cfg['decoder']['regularization']['latent_normalization']}{'/sn' if cfg['decoder']['regularization']['spectral_norm'] else ''} F=({fn},{fe}) kind={kind} N={g.x.shape[0]} "
                  f"E={g.edge_attr.shape[0]} out={tuple(out.shape)} err=({errs[0]:.1e},{errs[1]:.1e},{errs[2]:.1e}) noise={max(noise):.1e} "
                  f"{'ok' if ok else 'FAIL'}", flush=True)
            if not ok:
                failed += 1
                if not args.keep_going:
                    print(f"FAILED at case {case} (seed {args.seed})", flush=True)
                    sys.exit(1)
    if failed:
        print(f"{failed} of {args.cases} cases FAILED (seed {args.seed})", flush=True)
        sys.exit(1)
    print(f"{args.cases} cases ok, worst rel err {worst:.2e}, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
