"""Prediction transforms (mirror of /root/reference/src/algorithm/prediction_transform/: prediction_transform.py:7-158,
softplus_transform.py:6-11, exponential_transform.py:6-47, no_transform.py:6-11, __init__.py:6-27).

forward / inverse run as one HIP kernel each through the C ABI (amber_mpn_postprocess with the matching
AmberGlueDesc); tensors must live on the GPU.  The fused inference call (amber_amd.predict.predict) folds forward()
into the model's readout epilogue instead."""
import abc

import torch

from amber_amd import glue


class PredictionTransform(abc.ABC):
    kind: str = "none"

    def __init__(self, transform_config):
        self.transform_config = transform_config
        self.predict_residual = transform_config.predict_residual
        self.inverse_transform_in_loss = transform_config.inverse_transform_in_loss

    def __call__(self, prediction: torch.Tensor, baseline: torch.Tensor, is_train: bool = False) -> torch.Tensor:
        return self.forward(prediction, baseline, is_train)

    def applies(self, is_train: bool) -> bool:
        """Training with inverse_transform_in_loss leaves the raw output to the loss (prediction_transform.py:75-76)."""
        return not (self.inverse_transform_in_loss and is_train)

    def forward(self, prediction: torch.Tensor, baseline: torch.Tensor = None, is_train: bool = False) -> torch.Tensor:
        if not self.applies(is_train):
            return prediction
        if self.predict_residual:
            assert baseline is not None, "Baseline must be provided for residual prediction."
        return glue.postprocess(prediction, transform=self.kind, predict_residual=bool(self.predict_residual), baseline=baseline)

    def inverse(self, labels: torch.Tensor, baseline: torch.Tensor = None, is_train: bool = False) -> torch.Tensor:
        """Labels into network space: transform^-1(labels) (- transform^-1(baseline) for residual prediction)."""
        if not (self.inverse_transform_in_loss or not is_train):
            return labels
        if self.predict_residual:
            assert baseline is not None, "Baseline must be provided for residual prediction."
            return glue.inverse_transform(labels, self.kind, baseline=baseline)
        return self._inverse_transform(labels)

    def _transform(self, prediction: torch.Tensor) -> torch.Tensor:
        return glue.postprocess(prediction, transform=self.kind)

    def _inverse_transform(self, prediction: torch.Tensor) -> torch.Tensor:
        return glue.inverse_transform(prediction, self.kind)


class SoftplusTransform(PredictionTransform):
    kind = "softplus"


class ExponentialTransform(PredictionTransform):
    kind = "exp"


class NoTransform(PredictionTransform):
    kind = "none"


def get_transform(transform_config) -> PredictionTransform:
    name = transform_config.name
    if name in ["exp", "exponential"]:
        return ExponentialTransform(transform_config)
    if name == "softplus":
        return SoftplusTransform(transform_config)
    if name is None or name in [False, "none", "null"]:
        return NoTransform(transform_config)
    raise ValueError(f"Unsupported transform type: {name=}")
