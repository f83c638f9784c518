"""SparseTensor / PointTensor containers with the attribute surface the reference uses
(latte/models/spvcnn.py:125, latte/models/torchsparse_utils.py:25-29,55-57,82-90,
latte/data/semantic_kitti/semantic_kitti_dataloader.py:569, latte/tta/tools/latte.py:139)."""
from __future__ import annotations

import torch

__all__ = ["SparseTensor", "PointTensor", "cat"]


def _stride3(s):
    if isinstance(s, (tuple, list)):
        return tuple(int(v) for v in s)
    return (int(s),) * 3


class SparseTensor:
    def __init__(self, feats, coords, stride=1):
        self.feats = feats
        self.coords = coords
        self.stride = _stride3(stride)
        self.cmaps = {}
        self.kmaps = {}

    # torchsparse aliases
    @property
    def F(self):
        return self.feats

    @F.setter
    def F(self, v):
        self.feats = v

    @property
    def C(self):
        return self.coords

    @C.setter
    def C(self, v):
        self.coords = v

    @property
    def s(self):
        return self.stride

    @s.setter
    def s(self, v):
        self.stride = _stride3(v)

    def _moved(self, fn):
        out = SparseTensor(fn(self.feats), fn(self.coords), self.stride)
        out.cmaps, out.kmaps = self.cmaps, self.kmaps
        return out

    def cuda(self, *a, **k):
        self.feats = self.feats.cuda(*a, **k)
        self.coords = self.coords.cuda(*a, **k)
        return self

    def cpu(self):
        self.feats = self.feats.cpu()
        self.coords = self.coords.cpu()
        return self

    def to(self, device, non_blocking=True):
        self.feats = self.feats.to(device, non_blocking=non_blocking)
        self.coords = self.coords.to(device, non_blocking=non_blocking)
        return self

    def detach(self):
        self.feats = self.feats.detach()
        self.coords = self.coords.detach()
        return self

    def __add__(self, other):
        out = SparseTensor(self.feats + other.feats, self.coords, self.stride)
        out.cmaps, out.kmaps = self.cmaps, self.kmaps
        return out


class PointTensor:
    def __init__(self, feats, coords, idx_query=None, weights=None):
        self.F = feats
        self.C = coords
        self.idx_query = idx_query if idx_query is not None else {}
        self.weights = weights if weights is not None else {}
        self.additional_features = {"idx_query": {}, "counts": {}}

    @property
    def feats(self):
        return self.F

    @property
    def coords(self):
        return self.C

    def cuda(self, *a, **k):
        self.F = self.F.cuda(*a, **k)
        self.C = self.C.cuda(*a, **k)
        return self

    def to(self, device, non_blocking=True):
        self.F = self.F.to(device, non_blocking=non_blocking)
        self.C = self.C.to(device, non_blocking=non_blocking)
        return self

    def detach(self):
        self.F = self.F.detach()
        self.C = self.C.detach()
        return self

    def __add__(self, other):
        out = PointTensor(self.F + other.F, self.C, self.idx_query, self.weights)
        out.additional_features = self.additional_features
        return out


def cat(inputs):
    """torchsparse.cat (latte/models/spvcnn.py:144,148,156,160): channel concat of tensors sharing coords."""
    out = SparseTensor(torch.cat([t.feats for t in inputs], dim=1), inputs[0].coords, inputs[0].stride)
    out.cmaps, out.kmaps = inputs[0].cmaps, inputs[0].kmaps
    return out
