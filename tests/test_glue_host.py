"""Pre / post glue, CPU side: the oracle against the reference-generated golden vectors (bit for bit: both are torch
CPU), the host mirrors' bookkeeping (running statistics, state_dict names, configuration errors), and the product's
refusal to compute on CPU tensors."""
import numpy as np
import pytest
import torch

from amber_amd import glue, predict as predict_mod
from amber_amd.config import AttrDict
from amber_amd.normalizer import DummyRunningNormalizer, GraphRunningNormalizer, TorchRunningMeanStd, get_normalizer
from amber_amd.prediction_transform import ExponentialTransform, NoTransform, SoftplusTransform, get_transform
from amber_amd.synthetic import GraphBatch
from conftest import glue_cases, load_glue
from oracle import glue_oracle as go

CASES = glue_cases()


def test_fixtures_exist():
    assert len(CASES) >= 6


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    meta, d = load_glue(name)
    eps, clip = meta["epsilon"], meta["input_clip"]
    assert torch.equal(go.normalize(d["x"], d["node_mean"], d["node_var"], eps, clip), d["x_norm"])
    assert torch.equal(go.normalize(d["edge_attr"], d["edge_mean"], d["edge_var"], eps, clip), d["edge_attr_norm"])
    den = go.denormalize(d["pred"].flatten(), d["pred_mean"], d["pred_var"], eps)
    assert torch.equal(den, d["denorm"])
    kind, res, in_loss = meta["kind"], meta["predict_residual"], meta["inverse_transform_in_loss"]
    base = d["baseline"] if res else None
    for is_train, key in ((False, "out_eval"), (True, "out_train")):
        got = go.predict(d["pred"], pred_stats=(d["pred_mean"], d["pred_var"]), epsilon=eps, kind=kind, predict_residual=res,
                         inverse_transform_in_loss=in_loss, baseline=base, is_train=is_train)
        assert torch.equal(got, d[key]) or (torch.isnan(got) == torch.isnan(d[key])).all() and torch.equal(
            torch.nan_to_num(got), torch.nan_to_num(d[key]))
    inv = go.prediction_inverse(kind, res, in_loss, d["labels"], base)
    assert torch.equal(torch.nan_to_num(inv), torch.nan_to_num(d["labels_inv"]))
    sf = go.clip_detach_dampen(d["out_eval"], meta["clip_min"], meta["clip_max"], meta["scale0"], meta["scale1"])
    assert np.array_equal(sf, d["sizing_final"].numpy(), equal_nan=True)


@pytest.mark.parametrize("name", CASES)
def test_running_statistics_match_reference(name):
    """TorchRunningMeanStd.update (mirror and oracle) reproduces the statistics the reference accumulated."""
    meta, d = load_glue(name)
    fn, fe = d["x"].shape[1], d["edge_attr"].shape[1]
    tf = get_transform(AttrDict(name=meta["kind"], predict_residual=meta["predict_residual"],
                                inverse_transform_in_loss=meta["inverse_transform_in_loss"]))
    example = GraphBatch(x=torch.zeros(2, fn), edge_attr=torch.zeros(2, fe), y=torch.zeros(2))
    norm = GraphRunningNormalizer(example, True, True, tf, input_clip=meta["input_clip"])
    mean, var, count = torch.zeros(fn, dtype=torch.float64), torch.ones(fn, dtype=torch.float64), torch.tensor(1e-6, dtype=torch.float64)
    i = 0
    while f"upd{i}_x" in d:
        norm.node_normalizer.update(d[f"upd{i}_x"])
        norm.edge_normalizer.update(d[f"upd{i}_edge_attr"])
        mean, var, count = go.running_update(mean, var, count, d[f"upd{i}_x"])
        i += 1
    assert i >= 1
    for got in (norm.node_normalizer.mean, mean):
        assert torch.equal(got, d["node_mean"])
    for got in (norm.node_normalizer.var, var):
        assert torch.equal(got, d["node_var"])
    assert torch.equal(norm.node_normalizer.count, d["node_count"]) and torch.equal(count, d["node_count"])
    assert torch.equal(norm.edge_normalizer.mean, d["edge_mean"]) and torch.equal(norm.edge_normalizer.var, d["edge_var"])


def test_state_dict_names_match_the_reference():
    tf = SoftplusTransform(AttrDict(name="softplus", predict_residual=True, inverse_transform_in_loss=True))
    example = GraphBatch(x=torch.zeros(2, 4), edge_attr=torch.zeros(2, 2), y=torch.zeros(2))
    norm = get_normalizer(AttrDict(normalize_inputs=1, normalize_predictions=1, input_clip=20), example, tf)
    assert isinstance(norm, GraphRunningNormalizer) and norm.input_clip == 20 and norm.epsilon == 1e-6
    assert sorted(norm.state_dict()) == sorted(f"{m}.{b}" for m in ("node_normalizer", "edge_normalizer", "prediction_normalizer")
                                               for b in ("mean", "var", "count"))
    assert all(v.dtype == torch.float64 for v in norm.state_dict().values())
    assert norm.prediction_normalizer.mean.shape == (1,) and norm.node_normalizer.var.shape == (4,)
    assert isinstance(get_normalizer(AttrDict(normalize_inputs=0, normalize_predictions=0), example, tf), DummyRunningNormalizer)
    only_pred = get_normalizer(AttrDict(normalize_inputs=0, normalize_predictions=1, input_clip=20), example, tf)
    assert only_pred.node_normalizer is None and only_pred.prediction_normalizer is not None
    with pytest.raises(ValueError, match="Unsupported input type"):
        get_normalizer(AttrDict(normalize_inputs=1, normalize_predictions=1, input_clip=20), object(), tf)


def test_get_transform_mirrors_the_reference():
    mk = lambda n: AttrDict(name=n, predict_residual=False, inverse_transform_in_loss=True)  # noqa: E731
    assert isinstance(get_transform(mk("softplus")), SoftplusTransform)
    assert isinstance(get_transform(mk("exp")), ExponentialTransform) and isinstance(get_transform(mk("exponential")), ExponentialTransform)
    for n in (None, False, "none", "null"):
        assert isinstance(get_transform(mk(n)), NoTransform)
    with pytest.raises(ValueError, match="Unsupported transform type"):
        get_transform(mk("log"))
    t = get_transform(mk("softplus"))
    assert not t.applies(is_train=True) and t.applies(is_train=False)
    raw = torch.ones(3)
    assert t.forward(raw, is_train=True) is raw  # training with inverse_transform_in_loss: untouched, no kernel call


def test_single_row_update_gives_zero_variance_like_the_reference():
    rs = TorchRunningMeanStd(epsilon=1e-6, shape=(3,))
    rs.update(torch.tensor([[1.0, 2.0, 3.0]]))
    m, v, c = go.running_update(torch.zeros(3, dtype=torch.float64), torch.ones(3, dtype=torch.float64),
                                torch.tensor(1e-6, dtype=torch.float64), torch.tensor([[1.0, 2.0, 3.0]]))
    assert torch.equal(rs.mean, m) and torch.equal(rs.var, v) and torch.equal(rs.count, c)
    assert torch.isfinite(rs.var).all()


def test_damping_scale():
    assert predict_mod.damping_scale(None, 3, 0) == 1.0 and predict_mod.damping_scale(0, 3, 0) == 1.0
    assert predict_mod.damping_scale(0.618, 3, 0) == 1 / (0.618 ** 2) and predict_mod.damping_scale(0.618, 3, 2) == 1.0


def test_cpu_tensors_fail_loudly():
    with pytest.raises(RuntimeError, match="no CPU path"):
        glue.normalize_rows(torch.zeros(4, 2), torch.zeros(2, dtype=torch.float64), torch.ones(2, dtype=torch.float64), 1e-6, 10)
    with pytest.raises(RuntimeError, match="no CPU path"):
        glue.postprocess(torch.zeros(4), transform="softplus")
    with pytest.raises(RuntimeError, match="no CPU path"):
        glue.inverse_transform(torch.ones(4), "softplus")
    with pytest.raises(ValueError, match="Unsupported transform type"):
        glue.make_glue_desc(transform="log")
