class Data:            # name only; see the package docstring
    def __init__(self, *a, **k):
        raise NotImplementedError("placeholder")


class Batch(Data):
    pass
