"""Golden vectors for the pre / post glue, from the REFERENCE's own classes (build container only).

Executes, unmodified, from /root/reference: src/algorithm/normalizer/{torch_running_mean_std,running_normalizer,
graph_running_normalizer}.py and src/algorithm/prediction_transform/{__init__,prediction_transform,softplus_transform,
exponential_transform,no_transform}.py.  `src.algorithm` and `src.algorithm.normalizer` are registered as bare
namespace packages so that their __init__ files (which import lightning / the image pipeline) do not run; omegaconf and
torch_geometric come from tests/golden/_shims.  The numpy clip / damping step of the inference loop needs a
LightningModule instance, so its expected values are produced here by the same two numpy expressions
(mesh_generation_algorithm.py:370, amber.py:238-243) applied to the reference transforms' outputs.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_glue.py   ->  tests/golden/glue_*.npz
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_shims"))
sys.path.insert(0, "/root/reference")

import src  # noqa: E402,F401

for _name, _path in (("src.algorithm", "/root/reference/src/algorithm"),
                     ("src.algorithm.normalizer", "/root/reference/src/algorithm/normalizer")):
    _m = types.ModuleType(_name)
    _m.__path__ = [_path]
    sys.modules[_name] = _m

from omegaconf import DictConfig  # noqa: E402  (stand-in)
from torch_geometric.data import Data  # noqa: E402  (stand-in)

from src.algorithm.normalizer.graph_running_normalizer import GraphRunningNormalizer  # noqa: E402  (REFERENCE)
from src.algorithm.prediction_transform import get_transform  # noqa: E402  (REFERENCE)


def case(name, *, kind, residual, in_loss, seed, n_nodes=257, n_edges=1031, fn=4, fe=2, input_clip=20, updates=3,
         clip=(None, None), damping=(None, 3, 0), last_step=None, extreme=False):
    g = torch.Generator().manual_seed(seed)
    tf = get_transform(DictConfig({"name": kind, "predict_residual": residual, "inverse_transform_in_loss": in_loss}))
    example = Data(x=torch.zeros(2, fn), edge_attr=torch.zeros(2, fe), y=torch.zeros(2))
    norm = GraphRunningNormalizer(example_graph=example, normalize_inputs=True, normalize_predictions=True,
                                  prediction_transform=tf, input_clip=input_clip)
    # running statistics from a few "collected" graphs (labels: positive sizing fields)
    upd = []
    for _ in range(updates):
        nn_, ne_ = int(torch.randint(5, 60, (1,), generator=g)), int(torch.randint(9, 200, (1,), generator=g))
        d = Data(x=torch.randn(nn_, fn, generator=g) * 3 + 1, edge_attr=torch.rand(ne_, fe, generator=g) * 5,
                 y=torch.rand(nn_, generator=g) * 0.3 + 0.01, current_sizing_field=torch.rand(nn_, generator=g) * 0.5 + 0.02)
        norm.update_normalizers(d)
        upd.append(d)
    x = torch.randn(n_nodes, fn, generator=g) * 4 + 1
    ea = torch.rand(n_edges, fe, generator=g) * 6
    if extreme:  # values that hit the clip, zeros, denormals
        x[0, 0], x[1, 1], x[2, 2], x[3, 3] = 1e6, -1e6, 0.0, 1e-40
        ea[0, 0], ea[1, 1] = 1e9, -3e8
    batch = Data(x=x.clone(), edge_attr=ea.clone())
    out = norm.normalize_inputs(batch)
    pred = torch.randn(n_nodes, 1, generator=g) * (30 if extreme else 2)  # what the model returns: [N, 1] float32
    baseline = torch.rand(n_nodes, generator=g) * 0.5 + 0.02
    if extreme:
        baseline[0], baseline[1], baseline[2] = 1e-6, 25.0, 90.0
    flat = pred.flatten()
    den = norm.denormalize_predictions(flat)
    res = {}
    for is_train in (False, True):
        res[is_train] = tf.forward(den.clone(), baseline=baseline if residual else None, is_train=is_train)
    labels = torch.rand(n_nodes, generator=g) * 0.3 + 0.01
    inv = tf.inverse(labels, baseline=baseline if residual else None)
    # inference post-step: np.clip + damping on the eval prediction
    sf = np.clip(res[False].detach().cpu().numpy(), clip[0], clip[1]) if (clip[0] is not None or clip[1] is not None) \
        else res[False].detach().cpu().numpy()
    factor, steps, depth = damping
    scaling = 1.0 if (factor is None or factor == 0) else 1 / (factor ** (steps - depth - 1))
    sf = sf * scaling
    if last_step is not None and last_step != 1.0:
        sf = sf * last_step
    meta = dict(kind=kind, predict_residual=residual, inverse_transform_in_loss=in_loss, input_clip=input_clip, epsilon=norm.epsilon,
                clip_min=clip[0], clip_max=clip[1], scale0=scaling, scale1=1.0 if last_step is None else last_step)
    arrays = dict(
        x=x.numpy(), edge_attr=ea.numpy(), x_norm=out.x.numpy(), edge_attr_norm=out.edge_attr.numpy(),
        node_mean=norm.node_normalizer.mean.numpy(), node_var=norm.node_normalizer.var.numpy(),
        node_count=norm.node_normalizer.count.numpy(),
        edge_mean=norm.edge_normalizer.mean.numpy(), edge_var=norm.edge_normalizer.var.numpy(),
        pred_mean=norm.prediction_normalizer.mean.numpy(), pred_var=norm.prediction_normalizer.var.numpy(),
        pred_count=norm.prediction_normalizer.count.numpy(),
        pred=pred.numpy(), baseline=baseline.numpy(), denorm=den.numpy(), out_eval=res[False].numpy(), out_train=res[True].numpy(),
        labels=labels.numpy(), labels_inv=inv.numpy(), sizing_final=sf)
    for i, d in enumerate(upd):
        arrays[f"upd{i}_x"] = d.x.numpy()
        arrays[f"upd{i}_edge_attr"] = d.edge_attr.numpy()
        arrays[f"upd{i}_y"] = d.y.numpy()
        arrays[f"upd{i}_csf"] = d.current_sizing_field.numpy()
    np.savez_compressed(os.path.join(HERE, f"glue_{name}.npz"), meta=np.array(json.dumps(meta)), **arrays)
    print(name, "ok", {k: v.shape for k, v in arrays.items() if k in ("x_norm", "out_eval", "sizing_final")})


if __name__ == "__main__":
    case("softplus_residual", kind="softplus", residual=True, in_loss=True, seed=1, clip=(0.01, 0.4), damping=(0.618, 3, 0))
    case("softplus_plain", kind="softplus", residual=False, in_loss=False, seed=2, clip=(None, 0.3), damping=(0.618, 3, 2), last_step=0.9)
    case("exp_residual", kind="exp", residual=True, in_loss=True, seed=3, clip=(0.02, None), damping=(0, 3, 1))
    case("none_plain", kind=None, residual=False, in_loss=False, seed=4, input_clip=10)
    case("none_wide", kind="none", residual=True, in_loss=True, seed=7, fn=5, fe=1, n_nodes=64, n_edges=100)
    case("softplus_extreme", kind="softplus", residual=True, in_loss=True, seed=5, extreme=True, clip=(1e-3, 50.0), damping=(0.5, 3, 1))
    case("exp_extreme", kind="exponential", residual=False, in_loss=True, seed=6, extreme=True, n_nodes=33, n_edges=7)
