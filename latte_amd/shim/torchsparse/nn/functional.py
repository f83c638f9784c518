"""torchsparse.nn.functional surface used by latte/models/torchsparse_utils.py:15-23,40-54,68-81."""
import torch

from latte_amd import functional as LF
from latte_amd import ops

__all__ = ["sphash", "sphashquery", "spcount", "spvoxelize", "spdevoxelize", "calc_ti_weights", "spdownsample",
           "conv3d"]


def sphash(coords, offsets=None):
    coords = coords.int().contiguous()
    if offsets is None:
        return ops.sphash(coords)
    return ops.sphash(coords, offsets.int().contiguous())


def sphashquery(queries, references):
    return ops.sphashquery(queries.long().contiguous(), references.long().contiguous())


def spcount(coords, num):
    return ops.spcount(coords.int().contiguous(), int(num))


def spvoxelize(feats, coords, counts):
    return LF.spvoxelize(feats, coords, counts)


def spdevoxelize(feats, coords, weights):
    return LF.spdevoxelize(feats, coords, weights)


def calc_ti_weights(coords, idx_query, scale=1):
    with torch.no_grad():
        return ops.calc_ti_weights(coords.float().contiguous(), idx_query.long().contiguous(), int(scale))


def spdownsample(coords, stride=2, kernel_size=2, tensor_stride=1):
    s = stride[0] if isinstance(stride, (tuple, list)) else stride
    ts = tensor_stride[0] if isinstance(tensor_stride, (tuple, list)) else tensor_stride
    ks = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
    if s not in (1, ks):
        raise NotImplementedError("spdownsample: only stride in {1, kernel_size} (the case Latte's SPVCNN uses)")
    return ops.downsample_coords(coords.int().contiguous(), int(s) * int(ts))


def conv3d(input, weight, bias=None, kernel_size=3, stride=1, dilation=1, transposed=False):
    from latte_amd.sparse_modules import conv3d as _conv3d
    return _conv3d(input, weight, bias, kernel_size, stride, dilation, transposed)
