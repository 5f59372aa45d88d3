"""Stand-in for the absent third-party package `torch_geometric` (container-only, golden generation).
Only the attribute-bag behaviour the reference MPN relies on is provided."""
from . import data, utils  # noqa: F401
