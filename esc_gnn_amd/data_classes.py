"""
ESCGNNData -- the object process_pyg_data returns with return_data_object=True (models/escgnn_data_classes.py:13-39 of
the reference): a PyG `Data` whose operator attributes ('P', 'Q') collate block-diagonally (`__cat_dim__` = (0, 1)),
'index' attributes along dim 1, everything else along dim 0.

When torch_geometric is importable this IS a `torch_geometric.data.Data` subclass (so `Batch.from_data_list` and the
reference's loaders take it as they take the reference's class).  torch_geometric is absent from the build image; then a
minimal attribute container with the same surface (`keys()`, item and attribute access, iteration, `to`,
`__cat_dim__`, `operator_keys`) stands in, so code that only reads attributes keeps working.  PyG's collate itself is
not restated here: esc_gnn_amd.operators ingests the collated block-diagonal tensors (parity of the collate: unpinned).
"""
import torch

try:                                                    # pragma: no cover - not importable in the build image
    from torch_geometric.data import Data as _Base
    HAVE_PYG = True
except Exception:
    HAVE_PYG = False

    class _Base:
        def __init__(self, **kwargs):
            for k, v in kwargs.items():
                setattr(self, k, v)

        def keys(self):
            return [k for k in self.__dict__ if not k.startswith("_")]

        def __getitem__(self, key):
            return getattr(self, key)

        def __setitem__(self, key, value):
            setattr(self, key, value)

        def __contains__(self, key):
            return key in self.__dict__

        def __iter__(self):
            for k in self.keys():
                yield k, getattr(self, k)

        def to(self, device, *args, **kwargs):
            for k in self.keys():
                v = getattr(self, k)
                if torch.is_tensor(v):
                    setattr(self, k, v.to(device))
            return self


class ESCGNNData(_Base):
    operator_keys = ('P', 'Q')

    def __init__(self, *args, operator_keys=None, **kwargs):
        super().__init__(*args, **kwargs)
        if operator_keys is not None:
            self.operator_keys = operator_keys

    def __cat_dim__(self, key, value, *args, **kwargs):
        if 'index' in key:
            return 1
        elif key in self.operator_keys:
            return (0, 1)                               # block-diagonal batching of the operators
        return 0

    def to(self, device, *args, **kwargs):
        data = super().to(device, *args, **kwargs)
        for key in list(self.keys()):
            item = getattr(self, key, None)
            if torch.is_tensor(item):
                moved = item.to(device)
                op = getattr(item, "_escat_operator", None)
                if op is not None and moved.device == item.device:
                    moved._escat_operator = op          # keep the canonical form attached by process_pyg_data
                setattr(data, key, moved)
        return data
