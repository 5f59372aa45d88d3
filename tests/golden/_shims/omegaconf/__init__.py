"""Stand-in for the absent package `omegaconf`: DictConfig as a dict with attribute access."""


class DictConfig(dict):
    def __init__(self, content=None):
        super().__init__()
        for k, v in (content or {}).items():
            self[k] = DictConfig(v) if isinstance(v, dict) else v

    def __getattr__(self, item):
        try:
            return self[item]
        except KeyError as exc:
            raise AttributeError(item) from exc
