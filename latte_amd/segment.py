"""torch_scatter.scatter replacement (reference: `import torch_scatter.scatter as ts_scatter`,
latte/models/hr_module.py:5,236,237,252,474): segmented sum / mean / max over dim 0 by radix-sorting the index
and reducing contiguous segments in liblatte_hip.so (atomic-free, deterministic)."""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import ops

__all__ = ["scatter"]


class _Scatter(Function):
    @staticmethod
    def forward(ctx, src, index, dim_size, mode):
        idx = index.to(torch.int64).contiguous()
        n_rows = int(dim_size) if dim_size is not None else (int(idx.max().item()) + 1 if idx.numel() else 0)
        selem, inv, uk, seg = ops.sort_segments(idx)
        # segments are the distinct index values present; scatter's output has a row for every value in [0, n_rows)
        red = ops.segment_reduce(src.contiguous(), selem, seg, mode, want_argmax=(mode == "max"))
        arg = None
        if mode == "max":
            red, arg = red
        out = torch.zeros((n_rows,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        out[uk] = red
        ctx.mode = mode
        ctx.src_shape = src.shape
        ctx.save_for_backward(inv, seg, uk, arg)
        return out

    @staticmethod
    def backward(ctx, gout):
        inv, seg, uk, arg = ctx.saved_tensors
        g = gout.contiguous()[uk]
        if ctx.mode == "max":
            gs = torch.zeros(ctx.src_shape, dtype=gout.dtype, device=gout.device)
            flat = gs.reshape(gs.shape[0], -1)
            a = arg.reshape(arg.shape[0], -1).long()
            cols = torch.arange(flat.shape[1], device=gout.device).expand_as(a)
            ok = a >= 0
            flat[a[ok], cols[ok]] = g.reshape(g.shape[0], -1)[ok]
            return gs, None, None, None
        return ops.segment_reduce_bwd(g, inv, seg, ctx.mode == "mean"), None, None, None


def scatter(src: torch.Tensor, index: torch.Tensor, dim: int = 0, out=None, dim_size=None, reduce: str = "sum"):
    if dim != 0 or out is not None:
        raise NotImplementedError("scatter: only dim=0 without `out` (the form Latte uses)")
    if reduce not in ("sum", "add", "mean", "max"):
        raise ValueError(f"scatter: unsupported reduce {reduce!r}")
    if index.dim() != 1:
        raise NotImplementedError("scatter: 1-D index over dim 0")
    return _Scatter.apply(src, index, dim_size, "sum" if reduce == "add" else reduce)
