"""Generate golden input/output vectors by running the REFERENCE's own MPN classes.

Runs only in the build container, where /root/reference is mounted.  The reference's third-party
dependencies torch_geometric, torch_scatter and omegaconf are not installed, so the attribute-bag
stand-ins under tests/golden/_shims are put on sys.path first.  torch_scatter's
scatter_mean is restated in the stand-in, so that one link stays "parity unpinned".

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [case ...]
Writes tests/golden/<case>.npz holding: the case config (JSON), the reference state_dict, the input graph
and the reference outputs (node latents, edge latents, SupervisedMPN sizing output).  For the configurations the
backward kernels take (TRAINABLE), additionally the reference's autograd gradients of sum(out * grad_out) with respect
to every parameter ("grad/<state_dict name>"), computed in eval mode (edge dropout = identity, so they are
deterministic) with gradient recording on.
The GPU box never sees the reference; tests compare against these files.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

from omegaconf import DictConfig  # noqa: E402  (stand-in)
from torch_geometric.data import Batch  # noqa: E402  (stand-in)

import types  # noqa: E402

# src/algorithm/__init__.py imports lightning (absent, and not on the MPN path): register bare namespace
# packages for `src.algorithm` and `src.algorithm.architecture` so that only the two files on the path
# (supervised_mpn.py, mlp.py) are executed, unmodified, from /root/reference.
for _name, _path in (("src.algorithm", "/root/reference/src/algorithm"),
                     ("src.algorithm.architecture", "/root/reference/src/algorithm/architecture")):
    import src  # noqa: E402,F401
    _m = types.ModuleType(_name)
    _m.__path__ = [_path]
    sys.modules[_name] = _m

from src.algorithm.architecture.supervised_mpn import SupervisedMPN  # noqa: E402  (REFERENCE)

from amber_amd import synthetic as syn  # noqa: E402


def mlp_cfg(act="leakyrelu", layers=2):
    return {"activation_function": act, "num_layers": layers, "add_output_layer": False,
            "regularization": {"dropout": 0, "spectral_norm": False, "latent_normalization": None}}


def arch(latent, steps, residual="inner", layer_norm="inner", act="leakyrelu", layers=2, repeats=1, embedding=None,
         dec_act="leakyrelu", dec_layers=2, dec_norm=None, dec_spectral=False):
    return {
        "name": "mpn", "latent_dimension": latent, "create_graph_copy": True, "assert_graph_shapes": True,
        "edge_dropout": 0.1, "embedding": embedding,
        "stack": {"layer_norm": layer_norm, "num_steps": steps, "num_step_repeats": repeats,
                  "residual_connections": residual, "mlp": mlp_cfg(act, layers)},
        "decoder": {"activation_function": dec_act, "num_layers": dec_layers,
                    "regularization": {"dropout": 0, "spectral_norm": dec_spectral, "latent_normalization": dec_norm}},
    }


def graphs(kind, fn, fe, seed):
    rng = np.random.default_rng(seed)
    if kind == "grids":
        dims = [(6, 5), (4, 7), (9, 3)]
        els = [syn.directed_mesh_edges(syn.grid_triangulation_edges(a, b), a * b) for a, b in dims]
        return syn.make_batch(els, [a * b for a, b in dims], fn, fe, seed=seed)
    if kind == "hier":
        parts = [syn.hierarchical_graph(7, 6, 2, 2, rng), syn.hierarchical_graph(5, 5, 3, 2, rng)]
        return syn.make_batch([p[0] for p in parts], [p[1] for p in parts], fn, fe, seed=seed,
                              mask_outputs=[p[2] for p in parts])
    if kind == "random":
        el = syn.random_graph(70, 500, rng, isolated=6)
        mask = rng.random(70) < 0.7
        return syn.make_batch([el], [70], fn, fe, seed=seed, mask_outputs=[mask])
    if kind == "brick":
        el = syn.directed_mesh_edges(syn.grid3d_edges(4, 3, 3), 36)
        return syn.make_batch([el, el], [36, 36], fn, fe, seed=seed)
    raise ValueError(kind)


CASES = {
    # name: (arch kwargs, graph kind, F_n, F_e, seed)
    "debug_L64_s2": (dict(latent=64, steps=2), "grids", 5, 2, 11),  # config/_runs/debug.yaml shape
    "synth_L128_s2": (dict(latent=128, steps=2), "grids", 4, 2, 12),  # BASELINE config 5 shape, 2 steps
    "hier_L64_s3": (dict(latent=64, steps=3), "hier", 4, 2, 13),  # hub nodes + mask_output
    "outer_relu_L32": (dict(latent=32, steps=2, residual="outer", layer_norm="outer", act="relu"), "random", 3, 2, 14),
    "none_tanh_L32": (dict(latent=32, steps=2, residual=None, layer_norm=None, act="tanh", dec_act="tanh"), "random", 3, 1, 15),
    "mix_io_L32": (dict(latent=32, steps=2, residual="inner", layer_norm="outer"), "brick", 4, 2, 16),
    "mix_oi_L32": (dict(latent=32, steps=2, residual="outer", layer_norm="inner"), "brick", 4, 2, 17),
    "const_edge_L32": (dict(latent=32, steps=2), "random", 6, 0, 18),  # EmptyLinear edge embedding (F_e = 0)
    "const_node_L32": (dict(latent=32, steps=1), "grids", 0, 3, 19),  # EmptyLinear node embedding (F_n = 0)
    "repeat_L32_l3": (dict(latent=32, steps=2, repeats=2, layers=3), "grids", 4, 2, 20),  # shared weights, 3-layer MLPs
    "onelayer_L64": (dict(latent=64, steps=2, layers=1, dec_layers=1, dec_act="gelu"), "hier", 4, 2, 21),
    "embmlp_L32": (dict(latent=32, steps=1, embedding={"mlp": mlp_cfg("leakyrelu", 2)}), "grids", 4, 2, 22),
    "decnorm_L32": (dict(latent=32, steps=1, dec_norm="layer", dec_act="silu"), "random", 4, 2, 23),
    "specnorm_L32": (dict(latent=32, steps=1, dec_spectral=True), "hier", 4, 2, 24),  # decoder spectral_norm (mlp.py:62-63)
    "L256_s2": (dict(latent=256, steps=2), "hier", 4, 2, 25),  # the upper end of north_star's latent range (64-256)
    # training breadth: decoder LayerNorm with an activation the backward takes; 3-layer MLP embeddings; outer modes with
    # step repeats on a hub graph
    "decnorm_lrelu_L32": (dict(latent=32, steps=1, dec_norm="layer", dec_layers=3), "random", 4, 2, 26),
    "embmlp3_tanh_L64": (dict(latent=64, steps=1, act="tanh", dec_act="tanh", embedding={"mlp": mlp_cfg("tanh", 3)}), "hier", 4, 2, 27),
    "outer_hier_L64": (dict(latent=64, steps=2, repeats=2, residual="outer", layer_norm="outer"), "hier", 4, 2, 28),
    # round 3: the DEPTHS AMBER runs, pinned by the reference itself (config/algorithm/architecture/mpn.yaml:14 num_steps 20 at
    # latent 64; BASELINE config 5: latent 128, 10 steps) - an error that only shows after many block applications (the
    # reverted packed-multiply build of round 2 was 1e-3 off at 10-20 steps while every 2-step fixture passed) is caught by
    # the reference's own numbers, not only by the oracle
    "paper_L64_s20": (dict(latent=64, steps=20), "hier", 4, 2, 29),
    "synth_L128_s10": (dict(latent=128, steps=10), "grids", 4, 2, 30),
    # latent 128 outside the feature-per-lane kernels' default configuration (outer modes, tanh)
    "outer_L128_s2": (dict(latent=128, steps=2, residual="outer", layer_norm="outer"), "hier", 4, 2, 31),
    "tanh_L128_s2": (dict(latent=128, steps=2, act="tanh", dec_act="tanh"), "grids", 4, 2, 32),
}


# cases inside what amber_mpn_backward is built for (include/amber_mpn.h)
TRAINABLE = ("debug_L64_s2", "synth_L128_s2", "hier_L64_s3", "none_tanh_L32", "const_edge_L32", "const_node_L32", "repeat_L32_l3",
             "specnorm_L32", "outer_relu_L32", "mix_io_L32", "mix_oi_L32", "embmlp_L32", "decnorm_lrelu_L32", "embmlp3_tanh_L64",
             "outer_hier_L64", "L256_s2", "decnorm_L32", "onelayer_L64", "paper_L64_s20", "synth_L128_s10", "outer_L128_s2",
             "tanh_L128_s2")


def main():
    only = sys.argv[1:]  # optional: regenerate just the named cases
    for name, (akw, kind, fn, fe, seed) in CASES.items():
        if only and name not in only:
            continue
        cfg = arch(**akw)
        g = graphs(kind, fn, fe, seed)
        torch.manual_seed(seed)
        model = SupervisedMPN(DictConfig(cfg), example_graph=g)
        # LayerNorm affine defaults (1, 0) would hide gamma/beta mistakes: perturb deterministically.
        with torch.no_grad():
            for pname, p in model.named_parameters():
                if "layer_norm" in pname:
                    p.add_(0.1 * torch.randn(p.shape))
        if akw.get("dec_spectral"):
            # move the power-iteration vectors off their random start: training-mode forwards of the REFERENCE decoder
            model.train()
            with torch.no_grad():
                for _ in range(3):
                    model.decoder_mlp(torch.randn(7, cfg["latent_dimension"]))
        model.eval()
        batch = Batch(x=g.x.clone(), edge_attr=g.edge_attr.clone(), edge_index=g.edge_index.clone(),
                      batch=g.batch.clone(), mask_output=g.mask_output.clone())
        with torch.no_grad():
            out = model(batch)  # REFERENCE SupervisedMPN.forward
            nodes, edges, bvec = model.mpn(batch)  # REFERENCE MessagePassingBase.forward
        # caller's batch must be untouched (create_graph_copy=True)
        assert torch.equal(batch.x, g.x) and torch.equal(batch.edge_attr, g.edge_attr)
        blob = {f"param/{k}": v.detach().numpy() for k, v in model.state_dict().items()}
        cfg_full = dict(cfg, in_node_features=fn, in_edge_features=fe)
        blob.update({
            "config": np.frombuffer(json.dumps(cfg_full).encode(), dtype=np.uint8),
            "x": g.x.numpy(), "edge_attr": g.edge_attr.numpy(), "edge_index": g.edge_index.numpy(),
            "batch": g.batch.numpy(), "ptr": g.ptr.numpy(), "mask_output": g.mask_output.numpy(),
            "out_node": nodes["node"].numpy(), "out_edge": edges["edges"]["edge_attr"].numpy(),
            "out_sizing": out.numpy(),
        })
        if name in TRAINABLE:
            gen = torch.Generator().manual_seed(1000 + seed)
            grad_out = torch.randn(out.shape, generator=gen)
            model.zero_grad()
            out_g = model(batch)  # REFERENCE forward with autograd
            (out_g * grad_out).sum().backward()  # REFERENCE backward
            blob["grad_out"] = grad_out.numpy()
            for pname, p in model.named_parameters():
                # None: the output does not depend on the parameter (outer LayerNorm of the LAST block's edges) = zero
                blob[f"grad/{pname}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy()
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **blob)
        print(f"{name}: N={g.x.shape[0]} E={g.edge_index.shape[1]} params={sum(p.numel() for p in model.parameters())} "
              f"-> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
