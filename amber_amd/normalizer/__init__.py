"""Mirror of /root/reference/src/algorithm/normalizer/__init__.py:14-69 (graph inputs only: the image baseline's
normaliser is out of scope)."""
from amber_amd.normalizer.running_normalizer import DummyRunningNormalizer, GraphRunningNormalizer, RunningNormalizer  # noqa: F401
from amber_amd.normalizer.torch_running_mean_std import TorchRunningMeanStd  # noqa: F401


def get_normalizer(normalizer_config, example_input, prediction_transform) -> RunningNormalizer:
    normalize_inputs = normalizer_config.get("normalize_inputs")
    normalize_predictions = normalizer_config.get("normalize_predictions")
    if not (normalize_inputs or normalize_predictions):
        return DummyRunningNormalizer()
    if not all(hasattr(example_input, k) for k in ("x", "edge_attr", "y")):
        raise ValueError(f"Unsupported input type {type(example_input)}")
    return GraphRunningNormalizer(example_graph=example_input, normalize_inputs=normalize_inputs,
                                  normalize_predictions=normalize_predictions, prediction_transform=prediction_transform,
                                  input_clip=normalizer_config.get("input_clip"))
