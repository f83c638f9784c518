"""Point <-> voxel bridges of SPVCNN with the reference's names and semantics
(latte/models/torchsparse_utils.py:11-31 initial_voxelize, :36-59 point_to_voxel, :64-100 voxel_to_point).
Index maps (idx_query, counts) are produced by the HIP hash table and must be bit-exact with the reference."""
from __future__ import annotations

import torch

from . import functional as LF
from . import ops
from .tensor import PointTensor, SparseTensor

__all__ = ["initial_voxelize", "point_to_voxel", "voxel_to_point", "p2v_index", "v2p_index", "v2p_cells"]


def initial_voxelize(z: PointTensor, init_res: float, after_res: float, cache: dict | None = None) -> SparseTensor:
    """`cache` (optional, owned by the input batch): everything here except the pooled features depends on the
    coordinates only, so a second forward over the same batch (teacher, then student: latte/tta/tools/latte.py:159,176)
    reuses the voxel set, the point->voxel map and -- through the shared cmaps/kmaps dicts -- every kernel map/plan."""
    if cache is not None and "x0" in cache:
        new_float_coord, idx_query, counts, inserted_coords, cmaps, kmaps, csr = cache["x0"]
    else:
        new_float_coord = torch.cat([(z.C[:, :3] * init_res) / after_res, z.C[:, -1].view(-1, 1)], 1)
        floor_c = torch.floor(new_float_coord)
        pc_hash = ops.sphash(floor_c.int().contiguous())
        # torch.unique(pc_hash): voxels of x0 are ordered by ascending 60-bit hash
        sparse_hash, _, idx_query = ops.unique_sorted(pc_hash, key_bits=60, want_first=False)
        idx_query = idx_query.int()
        counts = ops.spcount(idx_query, sparse_hash.numel())
        csr = ops.csr_build(idx_query, sparse_hash.numel())   # points of every voxel, ascending: atomic-free pooling
        inserted_coords = LF.spvoxelize(floor_c, idx_query, counts, csr)
        inserted_coords = torch.round(inserted_coords).int()
        cmaps, kmaps = {}, {}
        if cache is not None:
            cache["x0"] = (new_float_coord, idx_query, counts, inserted_coords, cmaps, kmaps, csr)
    inserted_feat = LF.spvoxelize(z.F, idx_query, counts, csr)
    new_tensor = SparseTensor(inserted_feat, inserted_coords, 1)
    new_tensor.cmaps, new_tensor.kmaps = cmaps, kmaps
    new_tensor.cmaps.setdefault(new_tensor.stride, new_tensor.coords)
    z.additional_features["idx_query"][1] = idx_query
    z.additional_features["counts"][1] = counts
    # torchsparse strides are ints, ours are 3-tuples: register the map under both so that point_to_voxel at stride 1
    # finds it (same map: the stride-1 cells ARE the voxels of x0) instead of rebuilding it by a hash query
    z.additional_features["idx_query"].setdefault(new_tensor.stride, idx_query)
    z.additional_features["counts"].setdefault(new_tensor.stride, counts)
    z.additional_features.setdefault("p2v_csr", {})[new_tensor.stride] = csr
    z.C = new_float_coord
    return new_tensor


def _cell_coords(z: PointTensor, s: int) -> torch.Tensor:
    return torch.cat([torch.floor(z.C[:, :3] / s).int() * s, z.C[:, -1].int().view(-1, 1)], 1).contiguous()


def p2v_index(x: SparseTensor, z: PointTensor):
    """point -> voxel map of stride x.s (cached on z.additional_features)."""
    af = z.additional_features
    if af is None or af.get("idx_query") is None or af["idx_query"].get(x.s) is None:
        pc_hash = ops.sphash(_cell_coords(z, x.s[0]))
        sparse_hash = ops.sphash(x.C.int().contiguous())
        idx_query = ops.sphashquery(pc_hash, sparse_hash).int()
        counts = ops.spcount(idx_query, x.C.shape[0])
        af["idx_query"][x.s] = idx_query
        af["counts"][x.s] = counts
        af.setdefault("p2v_csr", {})[x.s] = ops.csr_build(idx_query, x.C.shape[0])
    return af["idx_query"][x.s], af["counts"][x.s]


def point_to_voxel(x: SparseTensor, z: PointTensor) -> SparseTensor:
    idx_query, counts = p2v_index(x, z)
    csr = z.additional_features.get("p2v_csr", {}).get(x.s)
    new_tensor = SparseTensor(LF.spvoxelize(z.F, idx_query, counts, csr), x.C, x.s)
    new_tensor.cmaps, new_tensor.kmaps = x.cmaps, x.kmaps
    return new_tensor


def v2p_index(x: SparseTensor, z: PointTensor, nearest: bool = False):
    """8-corner voxel ids and trilinear weights of every point at stride x.s -> ([N,8] int64, [N,8] fp32)."""
    off = LF.get_kernel_offsets(2, x.s, 1, device=z.C.device)
    old_hash = ops.sphash(_cell_coords(z, x.s[0]), off)
    pc_hash = ops.sphash(x.C.int().contiguous())
    idx_query = ops.sphashquery(old_hash, pc_hash)  # [8, N]
    weights = ops.calc_ti_weights(z.C.float().contiguous(), idx_query, x.s[0]).transpose(0, 1).contiguous()
    idx_query = idx_query.transpose(0, 1).contiguous()
    if nearest:
        weights[:, 1:] = 0.0
        idx_query[:, 1:] = -1
    return idx_query.int(), weights


def _v2p_cells(af: dict, idx_query_of: dict, s, n_vox: int):
    """(cells, point->voxel CSR, point->voxel index) of stride s, or None when the level has no point->voxel map."""
    cs = af.setdefault("v2p_cells", {})
    if cs.get(s) is None:
        p2v = af.get("p2v_csr", {}).get(s)
        if p2v is None or af["idx_query"].get(s) is None:
            return None
        cs[s] = ops.devoxelize_cells(LF._i32(idx_query_of[s]), p2v, n_vox)
    return cs[s], af["p2v_csr"][s], LF._i32(af["idx_query"][s])


def v2p_cells(x: SparseTensor, z: PointTensor):
    """What the backward of spdevoxelize at stride x.s gathers over (ops.devoxelize_cell_bwd): every point of a cell shares
    its 8 corner voxels, so the point->voxel CSR of the level plus a [V, 8] table replaces a CSR of the 8N-entry map.
    Built on first use and cached with the map; x's voxel set must be the one the point->voxel map was built on."""
    p2v_index(x, z)
    return _v2p_cells(z.additional_features, z.idx_query, x.s, x.C.shape[0])


def voxel_to_point(x: SparseTensor, z: PointTensor, nearest: bool = False) -> PointTensor:
    csr = None
    if torch.is_grad_enabled() and x.F.requires_grad:
        # backward only: looked up (or built) when the gradient arrives
        p2v_index(x, z)
        af, idxq, s, n_vox = z.additional_features, z.idx_query, x.s, x.C.shape[0]
        csr = lambda: _v2p_cells(af, idxq, s, n_vox)
    if z.idx_query is None or z.weights is None or z.idx_query.get(x.s) is None or z.weights.get(x.s) is None:
        # first use of this stride: the map is cached on the point tensor (and on every tensor derived from it)
        z.idx_query[x.s], z.weights[x.s] = v2p_index(x, z, nearest)
    new_feat = LF.spdevoxelize(x.F, z.idx_query[x.s], z.weights[x.s], csr)
    new_tensor = PointTensor(new_feat, z.C, idx_query=z.idx_query, weights=z.weights)
    new_tensor.additional_features = z.additional_features
    return new_tensor
