from .data import Data
from .batch import Batch

__all__ = ["Data", "Batch"]
