"""MeshGenerationAlgorithm._predict as ONE call (mirror of /root/reference/src/algorithm/core/mesh_generation_algorithm.py:
343-367, plus the clip + damping that follows the prediction in the inference step: _clip_detach :369-371,
Amber._clip_detach_dampen src/algorithm/core/amber.py:213-246).

Inference (no gradients): amber_mpn_predict normalises x / edge_attr into the workspace, runs the fused forward and
applies denormalise -> residual baseline -> transform -> clip -> damping inside the head kernel's readout epilogue.
Training (gradients recorded): the same stages as separate differentiable calls (normalise kernel, MpnFunction,
post-processing kernel with its backward)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib, engine, glue


def damping_scale(damping_factor: Optional[float], inference_steps: int, refinement_depth: int) -> float:
    """1 / damping_factor ** (inference_steps - refinement_depth - 1), or 1 (amber.py:231-238)."""
    if damping_factor is None or damping_factor == 0:
        return 1.0
    return 1 / (damping_factor ** (inference_steps - refinement_depth - 1))


def predict(model, normalizer, prediction_transform, batch, is_train: bool = False, flatten: bool = True, *,
            clip_min: Optional[float] = None, clip_max: Optional[float] = None, scale0: float = 1.0, scale1: float = 1.0):
    """Sizing-field prediction for a (batch of) graph(s) already on the GPU.  `model`: amber_amd SupervisedMPN;
    `normalizer`: amber_amd.normalizer.RunningNormalizer; `prediction_transform`: amber_amd.prediction_transform
    .PredictionTransform.  The caller's batch is never modified.  clip_* / scale*: the _clip_detach(_dampen) step, applied
    only when given (inference)."""
    node_n = getattr(normalizer, "node_normalizer", None)
    edge_n = getattr(normalizer, "edge_normalizer", None)
    pred_n = getattr(normalizer, "prediction_normalizer", None)
    baseline = getattr(batch, "current_sizing_field", None)
    applies = prediction_transform.applies(is_train)
    residual = bool(prediction_transform.predict_residual)
    if applies and residual:
        assert baseline is not None, "Baseline must be provided for residual prediction."
    eps = getattr(normalizer, "epsilon", 1.0e-6)
    clip = getattr(normalizer, "input_clip", 10)
    if engine.wants_grad(list(model.parameters())):
        if clip_min is not None or clip_max is not None or scale0 != 1.0 or scale1 != 1.0:
            raise ValueError("amber_amd: clip / damping belong to the detached inference path")
        view = _NormalizedView(batch, None if node_n is None else glue.normalize_rows(batch.x, node_n.mean, node_n.var, eps, clip),
                               None if edge_n is None else glue.normalize_rows(batch.edge_attr, edge_n.mean, edge_n.var, eps, clip))
        out = model(view)
        out = out.flatten() if flatten else out
        if pred_n is None and not applies:
            return out
        return glue.postprocess(out, transform=prediction_transform.kind, predict_residual=residual, baseline=baseline,
                                apply_transform=applies, mean=None if pred_n is None else pred_n.mean,
                                var=None if pred_n is None else pred_n.var, epsilon=eps)
    gdesc = glue.make_glue_desc(normalize_inputs=node_n is not None or edge_n is not None, denormalize_predictions=pred_n is not None,
                                transform=prediction_transform.kind, predict_residual=residual, apply_transform=applies,
                                clip_min=clip_min, clip_max=clip_max, scale0=scale0, scale1=scale1, input_clip=clip, epsilon=eps)
    out = model.fused_predict(batch, gdesc, node_n, edge_n, pred_n, baseline if (applies and residual) else None)
    return out.flatten() if flatten else out


class _NormalizedView:
    """The caller's batch with x / edge_attr replaced (the reference mutates its device copy of the batch instead)."""

    def __init__(self, source, x, edge_attr):
        self.__dict__["_src"] = source
        if x is not None:
            self.__dict__["x"] = x
        if edge_attr is not None:
            self.__dict__["edge_attr"] = edge_attr

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self.__dict__["_src"], name)
