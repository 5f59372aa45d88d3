class BaseData:
    pass


class Data(BaseData):
    """Attribute bag: keyword arguments become attributes; survives copy.deepcopy / hasattr / setattr."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
