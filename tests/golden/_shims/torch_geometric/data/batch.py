from .data import Data


class Batch(Data):
    pass
