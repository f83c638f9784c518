"""Input voxelization contract of the dataloaders on the GPU.

sparse_quantize restates torchsparse.utils.quantize.sparse_quantize as the reference calls it
(latte/data/semantic_kitti/semantic_kitti_dataloader.py:562, nuscenes_dataloader.py:359, a2d2_dataloader.py:285,
synthia_dataloader.py:292; first-party copy of the algorithm: latte/models/hr_module.py:58-97):
floor(coords / voxel_size) -> int32, unique voxels in lexicographic (x,y,z) order, first-point index, inverse.
The unique/sort/rank runs in liblatte_hip.so (hash table + ballot compaction + radix sort of the unique keys)."""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .tensor import SparseTensor

__all__ = ["sparse_quantize", "sparse_collate", "sparse_collate_fn", "quantize_scan_gpu"]


def _quantize_device(c: torch.Tensor):
    """c int32 [N,3] on the GPU -> (indices int64 [V], inverse int64 [N])."""
    if c.shape[0] == 0:
        z = torch.zeros(0, dtype=torch.int64, device=c.device)
        return z, z.clone()
    mn = c.amin(dim=0)
    shifted = (c - mn).contiguous()
    ext = int(shifted.max().item())
    if ext >= (1 << 20):
        raise ValueError(f"sparse_quantize: coordinate extent {ext} exceeds 2^20 cells per axis")
    keys = ops.pack_xyz_keys(shifted)
    _, first, inv = ops.unique_sorted(keys, key_bits=60)
    return first, inv


def sparse_quantize(coords, voxel_size=1, *, return_index: bool = False, return_inverse: bool = False,
                    device="cuda"):
    """numpy in -> numpy out (drop-in for the dataloader call sites); torch CUDA tensor in -> torch out."""
    is_np = isinstance(coords, np.ndarray)
    vs = np.array([voxel_size] * 3 if isinstance(voxel_size, (float, int)) else voxel_size)
    if is_np:
        c_host = np.floor(coords / vs).astype(np.int32)  # same host arithmetic as the reference (fp64 divide)
        c = torch.from_numpy(c_host).to(device)
    else:
        vt = torch.as_tensor(vs, dtype=torch.float64, device=coords.device)
        c = torch.floor(coords.to(torch.float64) / vt).to(torch.int32)
        if not c.is_cuda:
            c = c.to(device)
    first, inv = _quantize_device(c.contiguous())
    vox = c[first]
    if is_np:
        outs = [vox.cpu().numpy()]
        if return_index:
            outs.append(first.cpu().numpy())
        if return_inverse:
            outs.append(inv.cpu().numpy())
    else:
        outs = [vox] + ([first] if return_index else []) + ([inv] if return_inverse else [])
    return outs[0] if len(outs) == 1 else outs


def quantize_scan_gpu(points: torch.Tensor, intensity: torch.Tensor, scale: int = 20):
    """Dataloader contract a1 with raw points already in HBM: coords = round(p*scale) - min
    (latte/data/utils/augmentation_3d.py:48-56), sparse_quantize, F = [x,y,z,i][inds], C = coords[inds]
    (semantic_kitti_dataloader.py:542-577).  -> (coords int32 [V,3], feats fp32 [V,4], inds, inverse)."""
    c = torch.round(points * scale)
    c = c - c.amin(dim=0)
    c = c.to(torch.int32).contiguous()
    first, inv = _quantize_device(c)
    feats = torch.cat([points[first], intensity[first].reshape(-1, 1)], dim=1).float()
    return c[first], feats, first, inv


def sparse_collate(inputs):
    """torchsparse.utils.collate.sparse_collate (latte/data/collate.py:403-405): batch index appended as the LAST
    coordinate column."""
    coords, feats = [], []
    stride = inputs[0].stride
    for k, x in enumerate(inputs):
        c = x.coords if torch.is_tensor(x.coords) else torch.from_numpy(np.asarray(x.coords))
        f = x.feats if torch.is_tensor(x.feats) else torch.from_numpy(np.asarray(x.feats))
        b = torch.full((c.shape[0], 1), k, dtype=torch.int32, device=c.device)
        coords.append(torch.cat([c.int(), b], dim=1))
        feats.append(f)
    return SparseTensor(torch.cat(feats, 0), torch.cat(coords, 0), stride)


def sparse_collate_fn(inputs):
    if isinstance(inputs[0], dict):
        out = {}
        for name in inputs[0].keys():
            vals = [d[name] for d in inputs]
            if isinstance(vals[0], SparseTensor):
                out[name] = sparse_collate(vals)
            elif isinstance(vals[0], np.ndarray):
                out[name] = torch.stack([torch.from_numpy(v) for v in vals], 0)
            elif torch.is_tensor(vals[0]):
                out[name] = torch.stack(vals, 0)
            else:
                out[name] = vals
        return out
    return inputs
