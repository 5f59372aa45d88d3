"""CPU oracle for the pre / post numeric glue of AMBER's model call.  TEST INFRASTRUCTURE ONLY.

Plain torch CPU restatement (float64 statistics, float32 data) of what MeshGenerationAlgorithm._predict does either side
of self.model(batch), and of the clip + damping that follows.  Only `tests/` and `__graft_entry__.smoke()` may import
it, as the checker; the product (`amber_amd`) never imports anything from `oracle/`.

Pinning: tests/golden/make_golden_glue.py runs the reference's own classes (GraphRunningNormalizer,
TorchRunningMeanStd, Softplus / Exponential / NoTransform, unmodified, from /root/reference, under the stand-ins of
tests/golden/_shims for the absent omegaconf / torch_geometric) in the build container and writes
tests/golden/glue_*.npz; tests/test_glue_oracle.py holds this oracle to those vectors bit for bit.  The two host-side
numpy steps (_clip_detach, _clip_detach_dampen) need `self` of a LightningModule and are restated from their source
lines; their fixtures come from the same numpy expressions, run on the reference transforms' outputs.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


def normalize(x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, epsilon: float, input_clip: float) -> torch.Tensor:
    """RunningNormalizer._normalize (src/algorithm/normalizer/running_normalizer.py:87-106), feature dimension 1."""
    scaled = (x - mean.view(1, -1)) / torch.sqrt(var.view(1, -1) + epsilon)  # float32 - float64 -> float64
    return torch.clip(scaled, -input_clip, input_clip).float()


def denormalize(pred: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, epsilon: float) -> torch.Tensor:
    """RunningNormalizer._denormalize (running_normalizer.py:108-119)."""
    return (pred * torch.sqrt(var + epsilon) + mean).type(pred.dtype)


def transform_fwd(kind: Optional[str], v: torch.Tensor) -> torch.Tensor:
    """_transform of SoftplusTransform (softplus_transform.py:7-8), ExponentialTransform (exponential_transform.py:21-36),
    NoTransform (no_transform.py:7-8)."""
    if kind == "softplus":
        return torch.nn.functional.softplus(v)
    if kind in ("exp", "exponential"):
        out = torch.exp(v)
        if torch.any(torch.isnan(out)):
            out[torch.isnan(out)] = 100  # in place, as the reference does it
        return out
    return v


def transform_inv(kind: Optional[str], v: torch.Tensor) -> torch.Tensor:
    """_inverse_transform: softplus_transform.py:10-11, exponential_transform.py:38-47, no_transform.py:10-11."""
    if kind == "softplus":
        return v + torch.log(-torch.expm1(-v))
    if kind in ("exp", "exponential"):
        return torch.log(v)
    return v


def prediction_forward(kind, predict_residual: bool, inverse_transform_in_loss: bool, pred: torch.Tensor,
                       baseline: Optional[torch.Tensor], is_train: bool) -> torch.Tensor:
    """PredictionTransform.forward (src/algorithm/prediction_transform/prediction_transform.py:58-88)."""
    if inverse_transform_in_loss and is_train:
        return pred
    if predict_residual:
        assert baseline is not None, "Baseline must be provided for residual prediction."
        pred = pred + transform_inv(kind, baseline)
    return transform_fwd(kind, pred)


def prediction_inverse(kind, predict_residual: bool, inverse_transform_in_loss: bool, labels: torch.Tensor,
                       baseline: Optional[torch.Tensor], is_train: bool = False) -> torch.Tensor:
    """PredictionTransform.inverse (prediction_transform.py:90-128)."""
    if inverse_transform_in_loss or not is_train:
        labels = transform_inv(kind, labels)
        if predict_residual:
            labels = labels - transform_inv(kind, baseline)
    return labels


def clip_detach_dampen(sizing_field: torch.Tensor, clip_min, clip_max, scale0: float = 1.0, scale1: float = 1.0) -> np.ndarray:
    """_clip_detach (src/algorithm/core/mesh_generation_algorithm.py:369-371) and the damping of
    Amber._clip_detach_dampen (src/algorithm/core/amber.py:229-244): numpy float32 throughout (a Python float scalar
    does not widen a float32 array)."""
    sf = sizing_field.detach().cpu().numpy()
    if clip_min is not None or clip_max is not None:
        sf = np.clip(sf, clip_min, clip_max)
    sf = sf * scale0
    if scale1 != 1.0:
        sf = sf * scale1
    return sf


def running_update(mean, var, count, arr):
    """TorchRunningMeanStd.update (src/algorithm/normalizer/torch_running_mean_std.py:27-54); returns the new triple."""
    b_mean = arr.mean(dim=0).to(mean)
    b_var = torch.nan_to_num(arr.var(dim=0), nan=0.0).to(var)
    n = arr.shape[0]
    delta = b_mean - mean
    tot = count + n
    new_mean = mean + delta * n / tot
    m2 = var * count + b_var * n + torch.square(delta) * count * n / tot
    return new_mean, m2 / (count + n), tot


def predict(model_out: torch.Tensor, *, pred_stats, epsilon, kind, predict_residual, inverse_transform_in_loss, baseline,
            is_train=False, flatten=True) -> torch.Tensor:
    """The part of _predict after the model call (mesh_generation_algorithm.py:356-366)."""
    p = model_out.flatten() if flatten else model_out
    if pred_stats is not None:
        p = denormalize(p, pred_stats[0], pred_stats[1], epsilon)
    return prediction_forward(kind, predict_residual, inverse_transform_in_loss, p, baseline, is_train)
