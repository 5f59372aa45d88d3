"""Minimal attribute-access config for hosts without omegaconf (the reference passes a DictConfig;
SupervisedMPN only needs `.latent_dimension`, `.decoder` and `.get`)."""


class AttrDict(dict):
    def __init__(self, content=None, **kwargs):
        super().__init__()
        for k, v in {**(content or {}), **kwargs}.items():
            self[k] = AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __getattr__(self, item):
        try:
            return self[item]
        except KeyError as exc:
            raise AttributeError(item) from exc

    def __setattr__(self, key, value):
        self[key] = value


def default_architecture(latent_dimension: int = 64, num_steps: int = 20, **stack_overrides) -> AttrDict:
    """The reference defaults (config/algorithm/architecture/mpn.yaml, default_gnn.yaml, decoder/mlp.yaml)."""
    stack = {"layer_norm": "inner", "num_steps": num_steps, "residual_connections": "inner",
             "mlp": {"activation_function": "leakyrelu", "num_layers": 2, "add_output_layer": False,
                     "regularization": {"dropout": 0, "spectral_norm": False, "latent_normalization": None}}}
    stack.update(stack_overrides)
    return AttrDict({
        "name": "mpn", "latent_dimension": latent_dimension, "create_graph_copy": True, "assert_graph_shapes": False,
        "edge_dropout": 0.1, "stack": stack,
        "decoder": {"activation_function": "leakyrelu", "num_layers": 2,
                    "regularization": {"dropout": 0, "spectral_norm": False, "latent_normalization": None}},
    })
