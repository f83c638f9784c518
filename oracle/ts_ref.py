"""ORACLE (test infrastructure, never the product path) -- CPU restatement of the
third-party sparse-voxel ops Latte's 3D hot path calls.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

The arithmetic restated here lives in dependencies that are NOT vendored under
/root/reference and are not installed in this image:
  * mit-han-lab/torchsparse pinned at tag v1.4.0 (reference install.sh:9)
  * rusty1s/pytorch_scatter built for torch-1.13.0+cu118 (reference install.sh:1)
so every function below restates the library's published algorithm and cites the
reference call site that consumes it.  PARITY UNPINNED w.r.t. the real libraries: the
reference holds no golden vectors / tests for these ops.  The one
routine that IS pinned is sparse_quantize/ravel_hash: reference
latte/models/hr_module.py:58-97 is a first-party copy of torchsparse's, and
tests/golden/hr_*.npz (captured from that file) exercises it.

numpy for integer/byte work, torch-CPU (autograd-capable) for fp32 work.
"""
from __future__ import annotations

import numpy as np
import torch

FNV_OFFSET = np.uint64(14695981039346656037)
FNV_PRIME = np.uint64(1099511628211)
MASK60 = np.uint64(0xFFFFFFFFFFFFFFF)


# ---------------------------------------------------------------------------
# hashing  (torchsparse.nn.functional.sphash; call sites
# reference latte/models/torchsparse_utils.py:15,40,45,68,73)
# ---------------------------------------------------------------------------
def _fnv1a_rows(c: np.ndarray) -> np.ndarray:
    """64-bit FNV-1a over the 4 int32 of each row, folded to 60 bits."""
    c = np.ascontiguousarray(c, dtype=np.int32)
    h = np.full(c.shape[0], FNV_OFFSET, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(4):
            h ^= c[:, j].astype(np.uint32).astype(np.uint64)
            h *= FNV_PRIME
    h = (h >> np.uint64(60)) ^ (h & MASK60)
    return h.astype(np.int64)


def sphash(coords: np.ndarray, offsets: np.ndarray | None = None) -> np.ndarray:
    """coords [N,4] int32 (x,y,z,b) -> int64 [N]; with offsets [K,3] -> [K,N] hashes of (xyz+off_k, b)."""
    coords = np.ascontiguousarray(coords, dtype=np.int32)
    assert coords.ndim == 2 and coords.shape[1] == 4
    if offsets is None:
        return _fnv1a_rows(coords)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    K, N = offsets.shape[0], coords.shape[0]
    cc = np.repeat(coords[None, :, :], K, axis=0)
    with np.errstate(over="ignore"):
        cc[:, :, :3] += offsets[:, None, :]
    return _fnv1a_rows(cc.reshape(-1, 4)).reshape(K, N)


def sphashquery(queries: np.ndarray, references: np.ndarray) -> np.ndarray:
    """index of each query hash in `references`, -1 on miss; keeps the query shape
    (torchsparse_utils.py:17,46,74).  References are assumed collision-free (as the library does)."""
    q = np.asarray(queries, dtype=np.int64)
    r = np.asarray(references, dtype=np.int64).reshape(-1)
    if r.size == 0:
        return np.full(q.shape, -1, dtype=np.int64)
    order = np.argsort(r, kind="stable")
    rs = r[order]
    pos = np.searchsorted(rs, q.reshape(-1))
    pos_c = np.minimum(pos, rs.size - 1)
    hit = rs[pos_c] == q.reshape(-1)
    out = np.where(hit, order[pos_c], -1).astype(np.int64)
    return out.reshape(q.shape)


def spcount(idx: np.ndarray, num: int) -> np.ndarray:
    """histogram of idx ignoring negatives -> int32 [num]  (torchsparse_utils.py:18,47)."""
    idx = np.asarray(idx).reshape(-1)
    valid = idx[(idx >= 0) & (idx < num)]
    return np.bincount(valid, minlength=num).astype(np.int32)


def get_kernel_offsets(size: int, stride: int = 1, dilation: int = 1) -> np.ndarray:
    """torchsparse.nn.utils.get_kernel_offsets (torchsparse_utils.py:67):
    odd volume -> x fastest, even volume -> z fastest."""
    ax = np.arange(-size // 2 + 1, size // 2 + 1) * stride * dilation
    if (size ** 3) % 2 == 1:
        offs = [[x, y, z] for z in ax for y in ax for x in ax]
    else:
        offs = [[x, y, z] for x in ax for y in ax for z in ax]
    return np.asarray(offs, dtype=np.int32)


# ---------------------------------------------------------------------------
# quantize / collate (torchsparse.utils.quantize / .collate; dataloader call sites
# semantic_kitti_dataloader.py:562, collate.py:404; first-party copy hr_module.py:58-97)
# ---------------------------------------------------------------------------
def ravel_hash(x: np.ndarray, min_offset=None) -> np.ndarray:
    x = np.asarray(x)
    x = x - (np.min(x, axis=0) if min_offset is None else min_offset)
    x = x.astype(np.uint64, copy=False)
    xmax = np.max(x, axis=0).astype(np.uint64) + np.uint64(1)
    h = np.zeros(x.shape[0], dtype=np.uint64)
    for k in range(x.shape[1] - 1):
        h += x[:, k]
        h *= xmax[k + 1]
    h += x[:, -1]
    return h


def sparse_quantize(coords: np.ndarray, voxel_size=1):
    """-> (voxel coords [V,3] int32 lexicographic, indices [V] int64 first point, inverse [N] int64)."""
    vs = np.array([voxel_size] * 3, dtype=np.float64) if np.isscalar(voxel_size) else np.asarray(voxel_size)
    c = np.floor(coords / vs).astype(np.int32)
    _, indices, inverse = np.unique(ravel_hash(c), return_index=True, return_inverse=True)
    return c[indices], indices.astype(np.int64), inverse.reshape(-1).astype(np.int64)


def quantize_scan(points: np.ndarray, intensity: np.ndarray, scale: int = 20):
    """Dataloader contract a1: augmentation_3d.py:48-56 (round(p*scale) - min) then
    sparse_quantize, F=[x,y,z,i][inds], C=coords[inds] (semantic_kitti_dataloader.py:542-577)."""
    coords = np.round(points * scale)
    coords -= coords.min(0)
    coords = coords.astype(np.int64)
    vox, inds, inverse = sparse_quantize(coords)
    feats = np.concatenate([points[inds].astype(np.float32), intensity[inds].reshape(-1, 1).astype(np.float32)], 1)
    return coords[inds].astype(np.int32), feats, inds, inverse


def sparse_collate(coords_list, feats_list):
    """append the batch index as the LAST coord column, concat (collate.py:403-405)."""
    cs = [np.concatenate([c.astype(np.int32), np.full((c.shape[0], 1), k, np.int32)], 1)
          for k, c in enumerate(coords_list)]
    return np.concatenate(cs, 0), np.concatenate(feats_list, 0).astype(np.float32)


def spdownsample(coords: np.ndarray, stride: int, tensor_stride: int) -> np.ndarray:
    """coords//(s*ts)*(s*ts), unique rows sorted by (b,x,y,z) (torchsparse F.spdownsample, used by spnn.Conv3d)."""
    ss = stride * tensor_stride
    c = coords.astype(np.int32).copy()
    c[:, :3] = c[:, :3] // ss * ss
    c = c[:, [3, 0, 1, 2]]
    c = np.unique(c, axis=0)
    return np.ascontiguousarray(c[:, [1, 2, 3, 0]])


def build_kmap(coords: np.ndarray, tensor_stride: int, kernel_size: int, stride: int):
    """Kernel map of spnn.Conv3d (reference basic_blocks.py:17,34,51,58; semantics SURVEY §2.2):
    -> (nbmaps [M,2] int32 (in,out) k-major / out-ascending, nbsizes [K] int32, out_coords, results [K,Vout])."""
    offsets = get_kernel_offsets(kernel_size, tensor_stride)
    refs = sphash(coords)
    out_coords = spdownsample(coords, stride, tensor_stride) if stride > 1 else coords
    q = sphash(out_coords, offsets)
    res = sphashquery(q, refs)
    nbsizes = (res != -1).sum(1).astype(np.int32)
    kk, oo = np.nonzero(res != -1)
    nbmaps = np.stack([res[kk, oo], oo], 1).astype(np.int32)
    return nbmaps, nbsizes, out_coords, res


# ---------------------------------------------------------------------------
# fp32 ops (torch CPU; differentiable)
# ---------------------------------------------------------------------------
def spvoxelize(feats: torch.Tensor, idx: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """out[idx[i]] += feats[i]/counts[idx[i]] (torchsparse_utils.py:20,23,54)."""
    V = counts.shape[0]
    valid = (idx >= 0) & (idx < V)
    ii = idx[valid].long()
    f = feats[valid] / counts[ii].to(feats.dtype).unsqueeze(1)
    out = torch.zeros(V, feats.shape[1], dtype=feats.dtype)
    return out.index_add(0, ii, f)


def calc_ti_weights(coords: torch.Tensor, idx_query: torch.Tensor, scale: int = 1) -> torch.Tensor:
    """trilinear weights [8,N] (torchsparse_utils.py:75)."""
    p = coords
    pf = torch.floor(coords / scale) * scale if scale != 1 else torch.floor(coords)
    pc = pf + scale
    x, y, z = p[:, 0:1], p[:, 1:2], p[:, 2:3]
    xf, yf, zf = pf[:, 0:1].float(), pf[:, 1:2].float(), pf[:, 2:3].float()
    xc, yc, zc = pc[:, 0:1].float(), pc[:, 1:2].float(), pc[:, 2:3].float()
    w = torch.cat([
        (xc - x) * (yc - y) * (zc - z), (xc - x) * (yc - y) * (z - zf),
        (xc - x) * (y - yf) * (zc - z), (xc - x) * (y - yf) * (z - zf),
        (x - xf) * (yc - y) * (zc - z), (x - xf) * (yc - y) * (z - zf),
        (x - xf) * (y - yf) * (zc - z), (x - xf) * (y - yf) * (z - zf)], dim=1)
    w = w.transpose(1, 0).contiguous()
    if scale != 1:
        w = w / scale ** 3
    w[idx_query == -1] = 0
    w = w / (torch.sum(w, dim=0) + 1e-8)
    return w


def spdevoxelize(feats: torch.Tensor, idx: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """out[i] = sum_k w[i,k] * feats[idx[i,k]] skipping -1 (torchsparse_utils.py:81,93)."""
    N = idx.shape[0]
    out = torch.zeros(N, feats.shape[1], dtype=feats.dtype)
    for k in range(idx.shape[1]):
        ik = idx[:, k]
        valid = ik >= 0
        g = feats[ik.clamp(min=0).long()] * (w[:, k] * valid.to(w.dtype)).unsqueeze(1)
        out = out + g
    return out


def conv3d(feats: torch.Tensor, weight: torch.Tensor, nbmaps: np.ndarray, nbsizes: np.ndarray,
           n_out: int, transposed: bool = False) -> torch.Tensor:
    """gather -> per-offset GEMM -> scatter-add (spnn.Conv3d forward)."""
    out = torch.zeros(n_out, weight.shape[-1], dtype=feats.dtype)
    off = 0
    nb = torch.from_numpy(np.ascontiguousarray(nbmaps)).long()
    for k in range(weight.shape[0]):
        m = int(nbsizes[k])
        if m == 0:
            continue
        pairs = nb[off:off + m]
        off += m
        i_in, i_out = (pairs[:, 1], pairs[:, 0]) if transposed else (pairs[:, 0], pairs[:, 1])
        out = out.index_add(0, i_out, feats[i_in] @ weight[k])
    return out


def scatter_mean(src: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """torch_scatter.scatter(src, index, dim=0, reduce='mean') (hr_module.py:236,237,252):
    segment sum / count, count clamped >= 1, rows = index.max()+1."""
    n = int(index.max()) + 1
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).index_add(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add(0, index, torch.ones_like(index, dtype=src.dtype)).clamp(min=1)
    return out / cnt.view((-1,) + (1,) * (src.dim() - 1))
