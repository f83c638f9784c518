"""spnn.Conv3d / spnn.BatchNorm / spnn.ReLU with torchsparse v1.4.0's module surface
(reference use: latte/models/basic_blocks.py:17-72, latte/models/spvcnn.py:29-33).

Parameter names and layouts are kept so reference checkpoints load (latte/common/utils/checkpoint.py:75):
Conv3d.kernel is [K, Cin, Cout] ([Cin, Cout] for a 1x1x1 kernel); BatchNorm is an nn.BatchNorm1d subclass."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import functional as LF
from . import ops
from .tensor import SparseTensor

__all__ = ["Conv3d", "BatchNorm", "ReLU", "conv3d", "conv3d_bn", "build_kernel_map"]


def _n3(v):
    return LF._ntuple(v, 3)


def build_kernel_map(input: SparseTensor, kernel_size, stride, dilation):
    """-> (KernelMap, output coords).  Cached on the tensor under torchsparse's key
    (tensor_stride, kernel_size, stride, dilation)."""
    kernel_size, stride, dilation = _n3(kernel_size), _n3(stride), _n3(dilation)
    key = (input.stride, kernel_size, stride, dilation)
    hit = input.kmaps.get(key)
    if hit is not None:
        out_stride = tuple(input.stride[k] * stride[k] for k in range(3))
        return hit, input.cmaps[out_stride]
    coords = input.coords
    if coords.dtype != torch.int32:
        coords = coords.int()
    coords = coords.contiguous()
    offsets = LF.get_kernel_offsets(kernel_size, stride=input.stride, dilation=dilation, device=coords.device)
    if any(s > 1 for s in stride):
        if not all(stride[k] in (1, kernel_size[k]) for k in range(3)) or len(set(stride)) != 1:
            raise NotImplementedError("strided conv: only stride == kernel_size (the case SPVCNN uses)")
        out_stride = tuple(input.stride[k] * stride[k] for k in range(3))
        out_coords = input.cmaps.get(out_stride)
        if out_coords is None:
            out_coords = ops.downsample_coords(coords, stride[0] * input.stride[0])
    else:
        out_coords = coords
    # stride 1 and an odd kernel: same coordinate set on both sides, get_kernel_offsets is point-symmetric
    # (offsets[K-1-k] = -offsets[k]) -> half the probes, the other half of the table mirrored
    sub = all(s == 1 for s in stride) and (offsets.shape[0] % 2 == 1)
    results = ops.kmap_build(coords, out_coords, offsets, submanifold=sub)
    kmap = LF.KernelMap(results, coords.shape[0], out_coords.shape[0], submanifold=sub)
    input.kmaps[key] = kmap
    return kmap, out_coords


def conv3d(input: SparseTensor, weight, bias=None, kernel_size=3, stride=1, dilation=1, transposed=False,
           wcache: dict | None = None):
    """Functional sparse conv with torchsparse v1.4.0 semantics (kernel-map caching, transposed conv re-using the
    map of the finer stride, output coords from cmaps)."""
    feats = input.feats
    kernel_size, stride, dilation = _n3(kernel_size), _n3(stride), _n3(dilation)
    if kernel_size == (1, 1, 1) and stride == (1, 1, 1) and dilation == (1, 1, 1):
        out_feats = feats.matmul(weight)  # dense GEMM: stays in torch/rocBLAS
        if bias is not None:
            out_feats = out_feats + bias
        output = SparseTensor(out_feats, input.coords, input.stride)
    elif not transposed:
        kmap, out_coords = build_kernel_map(input, kernel_size, stride, dilation)
        out_feats = LF.sparse_conv(feats, weight, kmap, False, wcache)
        if bias is not None:
            out_feats = out_feats + bias
        output = SparseTensor(out_feats, out_coords, tuple(input.stride[k] * stride[k] for k in range(3)))
    else:
        tensor_stride = tuple(input.stride[k] // stride[k] for k in range(3))
        kmap = input.kmaps[(tensor_stride, kernel_size, stride, dilation)]
        out_feats = LF.sparse_conv(feats, weight, kmap, True, wcache)
        if bias is not None:
            out_feats = out_feats + bias
        output = SparseTensor(out_feats, input.cmaps[tensor_stride], tensor_stride)
    output.cmaps = input.cmaps
    output.cmaps.setdefault(output.stride, output.coords)
    output.kmaps = input.kmaps
    return output


def conv3d_bn(input: SparseTensor, conv: "Conv3d", bn: "BatchNorm", relu: bool, residual=None):
    """conv3d + BatchNorm(batch statistics) [+ residual] [+ ReLU] as one composite op (see functional._ConvBN); same
    kernel-map / output-coordinate logic as conv3d.  Returns None when the output would have a single row (BN refuses)."""
    ks, st, dl = conv.kernel_size, conv.stride, conv.dilation
    if not conv.transposed:
        kmap, out_coords = build_kernel_map(input, ks, st, dl)
        out_stride = tuple(input.stride[k] * st[k] for k in range(3))
    else:
        out_stride = tuple(input.stride[k] // st[k] for k in range(3))
        kmap = input.kmaps[(out_stride, ks, st, dl)]
        out_coords = input.cmaps[out_stride]
    if out_coords.shape[0] < 2:
        return None
    feats = LF.conv_bn(input.feats, conv.kernel, bn.weight, bn.bias, kmap, conv.transposed, residual=residual, relu=relu,
                       eps=bn.eps, wcache=conv._packed_cache)
    output = SparseTensor(feats, out_coords, out_stride)
    output.cmaps = input.cmaps
    output.cmaps.setdefault(output.stride, output.coords)
    output.kmaps = input.kmaps
    return output


class Conv3d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, bias=False,
                 transposed=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = _n3(kernel_size), _n3(stride), _n3(dilation)
        self.transposed = transposed
        self.kernel_volume = int(self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2])
        if self.kernel_volume > 1:
            self.kernel = nn.Parameter(torch.zeros(self.kernel_volume, in_channels, out_channels))
        else:
            self.kernel = nn.Parameter(torch.zeros(in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)
        self._packed_cache = {}  # mode -> ((param version, data_ptr), packed image); see functional._packed
        self.reset_parameters()

    def extra_repr(self):
        s = f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}"
        if self.stride != (1, 1, 1):
            s += f", stride={self.stride}"
        if self.transposed:
            s += ", transposed=True"
        return s

    def invalidate_packed(self) -> None:
        """drop the cached MFMA-fragment images of the kernel: required after writing the kernel through `.data`
        (torch does not version those writes; see functional.PACK_EPOCH)"""
        self._packed_cache.clear()

    def reset_parameters(self):
        std = 1.0 / math.sqrt((self.out_channels if self.transposed else self.in_channels) * self.kernel_volume)
        self.kernel.data.uniform_(-std, std)
        if self.bias is not None:
            self.bias.data.uniform_(-std, std)
        self.invalidate_packed()

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_packed()
        return super()._apply(fn, *args, **kwargs)

    def __deepcopy__(self, memo):
        # a copy (the EMA teacher, ema_util.create_ema_model) starts without packed images: its kernels get written
        # through `.data` right after the copy
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k == "_packed_cache" else copy.deepcopy(v, memo)
        return new

    def forward(self, input: SparseTensor) -> SparseTensor:
        return conv3d(input, self.kernel, self.bias, self.kernel_size, self.stride, self.dilation, self.transposed,
                      wcache=self._packed_cache)


def _wrap(input: SparseTensor, feats) -> SparseTensor:
    out = SparseTensor(feats, input.coords, input.stride)
    out.cmaps, out.kmaps = input.cmaps, input.kmaps
    return out


class BatchNorm(nn.BatchNorm1d):
    """BatchNorm1d over SparseTensor.F.  Batch statistics (training mode or track_running_stats=False, which is what
    configure_model_norm forces: latte/common/utils/ema_util.py:26-29) run in the HIP kernels."""

    def bn_feats(self, feats, residual=None, relu=False, post_add=False):
        """relu?(bn(feats) + residual); post_add: relu(bn(feats)) + residual (the sum after the activation)."""
        use_batch = self.training or not self.track_running_stats or self.running_mean is None
        if use_batch:
            stats = [] if (self.training and self.track_running_stats and self.running_mean is not None) else None
            y = LF.batch_norm(feats, self.weight, self.bias, self.eps, residual=residual, relu=relu, stats_out=stats,
                              post_add=post_add)
            if stats:
                mean, var = stats[0]
                n = feats.shape[0]
                with torch.no_grad():
                    if self.num_batches_tracked is not None:
                        self.num_batches_tracked += 1
                    m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                    self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                    self.running_var.mul_(1 - m).add_(var * (n / max(n - 1, 1)), alpha=m)
            return y
        # inference statistics: same HIP apply kernel with the running estimates (gradients flow through torch ops)
        if torch.is_grad_enabled() and (feats.requires_grad or self.weight.requires_grad):
            y = (feats - self.running_mean) * torch.rsqrt(self.running_var + self.eps) * self.weight + self.bias
            if post_add and relu and residual is not None:
                return torch.relu(y) + residual
            if residual is not None:
                y = y + residual
            return torch.relu(y) if relu else y
        return ops.bn_apply(feats, self.running_mean, self.running_var, self.weight, self.bias, self.eps,
                            residual=residual, relu=relu, post_add=post_add)

    def forward(self, input):
        if isinstance(input, SparseTensor):
            return _wrap(input, self.bn_feats(input.feats))
        return super().forward(input)


class ReLU(nn.ReLU):
    def forward(self, input):
        if isinstance(input, SparseTensor):
            return _wrap(input, super().forward(input.feats))
        return super().forward(input)
