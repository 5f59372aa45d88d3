"""Input / prediction normalisers (mirror of /root/reference/src/algorithm/normalizer/running_normalizer.py:11-119,
graph_running_normalizer.py:8-108, dummy_running_normalizer.py:6-17).

normalize_inputs / denormalize_predictions run as HIP kernels through the C ABI (amber_mpn_normalize_rows,
amber_mpn_postprocess): tensors must live on the GPU, there is no CPU path.  The fused inference call
(amber_amd.predict.predict) does not even call these methods: it hands the statistics to amber_mpn_predict."""
import abc
from typing import Optional

import torch
from torch import nn

from amber_amd import glue
from amber_amd.normalizer.torch_running_mean_std import TorchRunningMeanStd


class RunningNormalizer(nn.Module, abc.ABC):
    def __init__(self, num_predictions: int, normalize_predictions: bool, prediction_transform=None, input_clip: float = 10,
                 epsilon: float = 1.0e-6):
        super().__init__()
        self.prediction_normalizer = TorchRunningMeanStd(epsilon=epsilon, shape=(num_predictions,)) if normalize_predictions else None
        self.prediction_transform = prediction_transform
        self.epsilon = epsilon
        self.input_clip = input_clip

    @abc.abstractmethod
    def update_normalizers(self, inputs):
        raise NotImplementedError

    @abc.abstractmethod
    def normalize_inputs(self, inputs):
        raise NotImplementedError

    def update_prediction_normalizer(self, labels: torch.Tensor, baseline: torch.Tensor = None) -> None:
        """Statistics of the labels in network space (host-side bookkeeping of the training loop)."""
        if self.prediction_normalizer is None:
            return
        if self.prediction_transform is not None:
            labels = self.prediction_transform.inverse(labels, baseline=baseline)
        self.prediction_normalizer.update(labels)

    def denormalize_predictions(self, predictions: torch.Tensor) -> torch.Tensor:
        if self.prediction_normalizer is None:
            return predictions
        return glue.denormalize(predictions, self.prediction_normalizer.mean, self.prediction_normalizer.var, self.epsilon)

    def _normalize(self, input_tensor: torch.Tensor, normalizer: TorchRunningMeanStd, feature_dimension: int = 1) -> torch.Tensor:
        if feature_dimension != input_tensor.ndim - 1:
            raise NotImplementedError("amber_amd: the normalised dimension must be the last one (rows of features)")
        return glue.normalize_rows(input_tensor, normalizer.mean, normalizer.var, self.epsilon, self.input_clip)


class GraphRunningNormalizer(RunningNormalizer):
    def __init__(self, example_graph, normalize_inputs: bool, normalize_predictions: bool, prediction_transform,
                 input_clip: float = 10, epsilon: float = 1.0e-6):
        y = example_graph.y
        super().__init__(num_predictions=1 if y.ndim == 1 else y.shape[1], normalize_predictions=normalize_predictions,
                         prediction_transform=prediction_transform, input_clip=input_clip, epsilon=epsilon)
        if normalize_inputs:
            self.node_normalizer = TorchRunningMeanStd(epsilon=epsilon, shape=(example_graph.x.shape[1],))
            self.edge_normalizer = TorchRunningMeanStd(epsilon=epsilon, shape=(example_graph.edge_attr.shape[1],))
        else:
            self.node_normalizer = None
            self.edge_normalizer = None

    def update_normalizers(self, inputs) -> None:
        if hasattr(inputs, "x") and self.node_normalizer is not None:
            self.node_normalizer.update(inputs.x)
        if hasattr(inputs, "edge_attr") and self.edge_normalizer is not None:
            self.edge_normalizer.update(inputs.edge_attr)
        self.update_prediction_normalizer(labels=inputs.y, baseline=inputs.current_sizing_field)

    def normalize_inputs(self, inputs):
        """Replaces inputs.x / inputs.edge_attr by their normalised, clipped float32 versions (statistics untouched)."""
        if self.node_normalizer is not None:
            inputs.x = self._normalize(input_tensor=inputs.x, normalizer=self.node_normalizer)
        if self.edge_normalizer is not None:
            inputs.edge_attr = self._normalize(input_tensor=inputs.edge_attr, normalizer=self.edge_normalizer)
        return inputs


class DummyRunningNormalizer(RunningNormalizer):
    def __init__(self):
        super().__init__(num_predictions=0, normalize_predictions=False)

    def update_normalizers(self, inputs):
        pass

    def normalize_inputs(self, inputs):
        return inputs

    def denormalize_predictions(self, predictions: torch.Tensor) -> torch.Tensor:
        return predictions
