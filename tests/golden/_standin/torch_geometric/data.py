class Data:
    """Attribute container, which is all the reference uses `torch_geometric.data.Data` for on this path
    (inference.py:458 builds it, models/mpn.py:266 reads .x / .edge_index / .edge_attr, inference.py:489 .num_nodes)."""

    def __init__(self, **kwargs):
        self.__dict__.update(kwargs)

    @property
    def num_nodes(self):
        return self.x.shape[0]


class Batch(Data):
    pass
