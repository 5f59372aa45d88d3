"""Pre / post glue of _predict on the GPU, through the C ABI (amber_mpn_normalize_rows, amber_mpn_postprocess(+_backward),
amber_mpn_inverse_transform, amber_mpn_predict), against the reference-generated golden vectors and the CPU oracle.
Needs an MI355X: run with `-m gpu`.

Bars: input normalisation and de-normalisation are float64 IEEE arithmetic with one rounding to float32 -> BIT-EXACT.
The transforms use float32 exp / log / expm1 / log1p, whose device implementations differ from the CPU's in the last
ulp: POST_TOL = 2e-6 relative per element (plus 1e-30 absolute).  The "extreme" fixtures push the transform's argument u to
|u| ~ 100, where one ulp of u (7.6e-6 absolute, from the last-ulp difference of log / expm1 in transform^-1(baseline))
already moves exp(u) by 7.6e-6 relative: EXTREME_TOL = 2e-5 there.  The fused call inherits the forward's 1e-5."""
import numpy as np
import pytest
import torch

from amber_amd import engine, glue, predict as predict_mod, synthetic as syn
from amber_amd.architecture.supervised_mpn import SupervisedMPN
from amber_amd.config import AttrDict
from amber_amd.normalizer import GraphRunningNormalizer
from amber_amd.prediction_transform import get_transform
from conftest import close, glue_cases, load_glue, rel_err
from oracle import glue_oracle as go
from oracle import mpn_oracle as orc

pytestmark = pytest.mark.gpu
POST_TOL = 2e-6
EXTREME_TOL = 2e-5
CASES = glue_cases()


def _close(got: torch.Tensor, want: torch.Tensor, tol=POST_TOL):
    got, want = got.detach().cpu().double(), want.double()
    assert got.shape == want.shape
    nan = torch.isnan(want)
    assert torch.equal(torch.isnan(got), nan)
    inf = torch.isinf(want)
    assert torch.equal(got[inf], want[inf])
    ok = ~(nan | inf)
    err = (got[ok] - want[ok]).abs() - tol * want[ok].abs() - 1e-30
    assert (err <= 0).all(), f"worst excess {float(err.max()):.3e}"


def _normalizer(meta, d, device):
    fn, fe = d["x"].shape[1], d["edge_attr"].shape[1]
    tf = get_transform(AttrDict(name=meta["kind"], predict_residual=meta["predict_residual"],
                                inverse_transform_in_loss=meta["inverse_transform_in_loss"]))
    example = syn.GraphBatch(x=torch.zeros(2, fn), edge_attr=torch.zeros(2, fe), y=torch.zeros(2))
    norm = GraphRunningNormalizer(example, True, True, tf, input_clip=meta["input_clip"], epsilon=meta["epsilon"])
    norm.load_state_dict({"node_normalizer.mean": d["node_mean"], "node_normalizer.var": d["node_var"],
                          "node_normalizer.count": d["node_count"], "edge_normalizer.mean": d["edge_mean"],
                          "edge_normalizer.var": d["edge_var"], "edge_normalizer.count": d["node_count"],
                          "prediction_normalizer.mean": d["pred_mean"], "prediction_normalizer.var": d["pred_var"],
                          "prediction_normalizer.count": d["pred_count"]}, strict=True)
    return norm.to(device), tf


@pytest.mark.parametrize("name", CASES)
def test_normalize_inputs_bit_exact_vs_reference(gpu_device, name):
    meta, d = load_glue(name)
    norm, _ = _normalizer(meta, d, gpu_device)
    batch = syn.GraphBatch(x=d["x"].to(gpu_device), edge_attr=d["edge_attr"].to(gpu_device))
    out = norm.normalize_inputs(batch)
    assert out.x.dtype == torch.float32
    assert torch.equal(out.x.cpu(), d["x_norm"]) and torch.equal(out.edge_attr.cpu(), d["edge_attr_norm"])


def test_normalize_rows_large_and_special_values(gpu_device):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(300_001, 7, generator=g) * 50
    x[5, 0], x[6, 1], x[7, 2], x[8, 3] = float("nan"), float("inf"), float("-inf"), -0.0
    mean = torch.randn(7, generator=g, dtype=torch.float64) * 3
    var = torch.rand(7, generator=g, dtype=torch.float64) * 40
    var[3] = 0.0  # only epsilon under the root
    got = glue.normalize_rows(x.to(gpu_device), mean, var, 1e-6, 20.0).cpu()
    want = go.normalize(x, mean, var, 1e-6, 20.0)
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    assert torch.equal(torch.nan_to_num(got, nan=0.0), torch.nan_to_num(want, nan=0.0))
    assert glue.normalize_rows(torch.empty(0, 7, device=gpu_device), mean, var, 1e-6, 20.0).shape == (0, 7)
    with pytest.raises(ValueError, match="statistics"):
        glue.normalize_rows(x[:4].to(gpu_device), mean[:3], var[:3], 1e-6, 20.0)
    with pytest.raises(TypeError, match="float64"):
        glue.normalize_rows(x[:4].to(gpu_device), mean.float(), var.float(), 1e-6, 20.0)


@pytest.mark.parametrize("name", CASES)
def test_postprocess_stages_vs_reference(gpu_device, name):
    meta, d = load_glue(name)
    norm, tf = _normalizer(meta, d, gpu_device)
    pred = d["pred"].flatten().to(gpu_device)
    base = d["baseline"].to(gpu_device) if meta["predict_residual"] else None
    den = norm.denormalize_predictions(pred)
    assert torch.equal(den.cpu(), d["denorm"])  # float64 multiply-add, rounded once: bit-exact
    tol = EXTREME_TOL if "extreme" in name else POST_TOL
    _close(tf.forward(den, baseline=base, is_train=False), d["out_eval"], tol)
    _close(tf.forward(den, baseline=base, is_train=True), d["out_train"], tol)
    # labels into network space: a difference of two logs / inverse softplus values (cancellation), so the bar is
    # relative to the tensor's scale
    assert rel_err(tf.inverse(d["labels"].to(gpu_device), baseline=base).cpu(), d["labels_inv"]) <= 1e-6
    # everything after the model call in one kernel, clip + damping included
    final = glue.postprocess(pred, transform=meta["kind"], predict_residual=meta["predict_residual"], baseline=base,
                             mean=d["pred_mean"], var=d["pred_var"], epsilon=meta["epsilon"], clip_min=meta["clip_min"],
                             clip_max=meta["clip_max"], scale0=meta["scale0"], scale1=meta["scale1"])
    _close(final, d["sizing_final"], tol)


def _cfg(latent=64, steps=3, act="leakyrelu"):
    mlp = {"activation_function": act, "num_layers": 2, "add_output_layer": False,
           "regularization": {"dropout": 0, "spectral_norm": False, "latent_normalization": None}}
    return {"name": "mpn", "latent_dimension": latent, "create_graph_copy": True, "assert_graph_shapes": False, "edge_dropout": 0.0,
            "embedding": None, "in_node_features": 4, "in_edge_features": 2,
            "stack": {"layer_norm": "inner", "num_steps": steps, "num_step_repeats": 1, "residual_connections": "inner", "mlp": mlp},
            "decoder": {"activation_function": act, "num_layers": 2,
                        "regularization": {"dropout": 0, "spectral_norm": False, "latent_normalization": None}}}


def _setup(gpu_device, name, act="leakyrelu", masked=True, seed=3):
    meta, d = load_glue(name)
    norm, tf = _normalizer(meta, d, gpu_device)
    cfg = _cfg(act=act)
    rng = np.random.default_rng(seed)
    if masked:
        el, n, mask = syn.hierarchical_graph(14, 11, 2, 2, rng)
        g = syn.make_batch([el], [n], 4, 2, seed=seed, mask_outputs=[mask])
    else:
        g = syn.grid_batch(num_graphs=2, nx=13, ny=9)
    g.x = g.x * 3 + 1  # raw, un-normalised features
    g.edge_attr = g.edge_attr.abs() * 4
    n_out = int(g.mask_output.sum())
    g.current_sizing_field = torch.rand(n_out, generator=torch.Generator().manual_seed(seed)) * 0.4 + 0.02
    params = orc.init_params(cfg, seed=seed)
    model = SupervisedMPN(AttrDict(cfg), syn.GraphBatch(x=g.x, edge_attr=g.edge_attr))
    model.load_state_dict(params, strict=True)
    return meta, d, norm, tf, cfg, params, g, model.to(gpu_device)


def _oracle_predict(meta, d, cfg, params, g, is_train=False):
    eps, clip = meta["epsilon"], meta["input_clip"]
    xn = go.normalize(g.x, d["node_mean"], d["node_var"], eps, clip)
    en = go.normalize(g.edge_attr, d["edge_mean"], d["edge_var"], eps, clip)
    out = orc.supervised_mpn_forward(params, cfg, xn, en, g.edge_index, g.mask_output)
    return go.predict(out, pred_stats=(d["pred_mean"], d["pred_var"]), epsilon=eps, kind=meta["kind"],
                      predict_residual=meta["predict_residual"], inverse_transform_in_loss=meta["inverse_transform_in_loss"],
                      baseline=g.current_sizing_field if meta["predict_residual"] else None, is_train=is_train)


@pytest.mark.parametrize("name", ["softplus_residual", "softplus_plain", "exp_residual", "none_plain"])
@pytest.mark.parametrize("masked", [True, False])
def test_fused_predict_matches_oracle_pipeline(gpu_device, name, masked):
    """amber_mpn_predict (normalise -> forward -> post-processing in the readout epilogue) against the oracle chain."""
    meta, d, norm, tf, cfg, params, g, model = _setup(gpu_device, name, masked=masked)
    if d["x"].shape[1] != 4:
        pytest.skip("fixture statistics are for other feature widths")
    gb = g.to(gpu_device)
    x_before, e_before = gb.x.clone(), gb.edge_attr.clone()
    model.eval()
    with torch.no_grad():
        got = predict_mod.predict(model, norm, tf, gb, is_train=False)
    want = _oracle_predict(meta, d, cfg, params, g)
    assert got.shape == want.shape and close(got.cpu(), want)
    assert torch.equal(gb.x, x_before) and torch.equal(gb.edge_attr, e_before)  # the caller's batch is untouched
    # with the inference post-step folded in
    with torch.no_grad():
        got2 = predict_mod.predict(model, norm, tf, gb, clip_min=meta["clip_min"], clip_max=meta["clip_max"],
                                   scale0=meta["scale0"], scale1=meta["scale1"])
    want2 = torch.from_numpy(go.clip_detach_dampen(want, meta["clip_min"], meta["clip_max"], meta["scale0"], meta["scale1"]))
    assert close(got2.cpu(), want2)
    # the fused call equals the staged calls of the same library bit for bit
    with torch.no_grad():
        staged = model(syn.GraphBatch(x=glue.normalize_rows(gb.x, d["node_mean"], d["node_var"], meta["epsilon"], meta["input_clip"]),
                                      edge_attr=glue.normalize_rows(gb.edge_attr, d["edge_mean"], d["edge_var"], meta["epsilon"],
                                                                    meta["input_clip"]),
                                      edge_index=gb.edge_index, batch=gb.batch, mask_output=gb.mask_output)).flatten()
        staged = tf.forward(norm.denormalize_predictions(staged), baseline=gb.current_sizing_field if meta["predict_residual"] else None)
    assert torch.equal(staged, got)


@pytest.mark.parametrize("name", ["softplus_residual", "softplus_plain", "exp_residual"])
def test_training_predict_gradients_match_oracle(gpu_device, name):
    """is_train=True with gradients: normalise kernel -> MpnFunction -> post-processing kernel and its backward."""
    meta, d, norm, tf, cfg, params, g, model = _setup(gpu_device, name, act="tanh", masked=True, seed=5)
    gb = g.to(gpu_device)
    model.train()
    out = predict_mod.predict(model, norm, tf, gb, is_train=True)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    want = _oracle_predict(meta, d, cfg, p, g, is_train=True)
    assert close(out.detach().cpu(), want.detach())
    w = torch.randn(want.shape, generator=torch.Generator().manual_seed(1))
    (want * w).sum().backward()
    (out * w.to(gpu_device)).sum().backward()
    for k, q in model.named_parameters():
        assert rel_err(q.grad.cpu(), p[k].grad) <= 1e-4, k
    with pytest.raises(ValueError, match="detached inference path"):
        predict_mod.predict(model, norm, tf, gb, is_train=True, clip_min=0.1)


def test_postprocess_backward_matches_torch_autograd(gpu_device):
    g = torch.Generator().manual_seed(2)
    pred = (torch.randn(5000, generator=g) * 8)
    pred[:3] = torch.tensor([25.0, -40.0, 0.0])
    base = torch.rand(5000, generator=g) + 0.05
    mean, var = torch.tensor([0.3], dtype=torch.float64), torch.tensor([2.5], dtype=torch.float64)
    w = torch.randn(5000, generator=g)
    for kind in ("softplus", "exp", None):
        for residual in (True, False):
            pc = pred.clone().requires_grad_(True)
            want = go.prediction_forward(kind, residual, False, go.denormalize(pc, mean, var, 1e-6), base, False)
            (want * w).sum().backward()
            pg = pred.clone().to(gpu_device).requires_grad_(True)
            got = glue.postprocess(pg, transform=kind, predict_residual=residual, baseline=base.to(gpu_device), mean=mean, var=var)
            (got * w.to(gpu_device)).sum().backward()
            _close(got, want.detach(), tol=5e-6)
            _close(pg.grad, pc.grad, tol=1e-5)


def test_glue_errors(gpu_device):
    meta, d, norm, tf, cfg, params, g, model = _setup(gpu_device, "softplus_residual")
    gb = g.to(gpu_device)
    del gb.current_sizing_field
    with torch.no_grad(), pytest.raises(AssertionError, match="Baseline must be provided"):
        predict_mod.predict(model.eval(), norm, tf, gb)
    gb.current_sizing_field = torch.ones(3, device=gpu_device)
    with torch.no_grad(), pytest.raises(ValueError, match="baseline"):
        predict_mod.predict(model.eval(), norm, tf, gb)
    with pytest.raises(AssertionError, match="Baseline must be provided"):
        glue.postprocess(torch.zeros(4, device=gpu_device), transform="softplus", predict_residual=True)
