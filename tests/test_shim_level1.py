"""Level-1 boundary: the reference's latte/models/*.py import `torchsparse` / `torch_scatter`; with
latte_amd/shim on sys.path those names resolve to the drop-in packages.  These tests are OUR restatement of the reference's
call sequences through exactly those import names (latte/models/torchsparse_utils.py:11-100,
latte/data/semantic_kitti/semantic_kitti_dataloader.py:562-577, latte/data/collate.py:403-405), bit-equal to the
latte_amd functions the product path uses -- plus the module tree of the reference's own Net3DSeg/SPVCNN instantiated on
the shim (tests/golden/spvcnn_state_keys.json, made by tests/golden/make_state_keys.py) against latte_amd.spvcnn."""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def shim():
    import latte_amd
    latte_amd.install_shims()
    import torchsparse
    assert "latte_amd" in torchsparse.__file__
    return torchsparse


@pytest.mark.parametrize("tag,cr,ncls", [("cr1.0_C10", 1.0, 10), ("cr0.5_C9", 0.5, 9)])
def test_state_dict_matches_the_reference_module_tree_on_the_shim(tag, cr, ncls):
    from latte_amd.spvcnn import Net3DSeg
    want = json.load(open(os.path.join(GOLDEN, "spvcnn_state_keys.json")))[tag]
    net = Net3DSeg(ncls, True, "SPVCNN", {"cr": cr})
    got = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert set(got) == set(want["keys"]), (set(got) ^ set(want["keys"]))
    assert all(got[k] == want["keys"][k] for k in got)
    # same leaf classes under the same names where the reference has a module (ema_util.py:23-33 matches class names)
    mods = {n: type(m).__name__ for n, m in net.named_modules() if n}
    for n, cls in want["modules"].items():
        if cls in ("Conv3d", "BatchNorm", "ReLU", "Linear", "Dropout", "BatchNorm1d"):
            assert mods.get(n) == cls or (cls == "BatchNorm1d" and mods.get(n) == "BatchNorm"), (n, cls, mods.get(n))


def test_shim_exports_every_name_the_reference_imports(shim):
    import torchsparse
    import torchsparse.nn as spnn
    import torchsparse.nn.functional as F
    from torchsparse import PointTensor, SparseTensor  # noqa: F401  (torchsparse_utils.py:3, spvcnn.py)
    from torchsparse.nn.utils import get_kernel_offsets  # noqa: F401  (torchsparse_utils.py:4)
    from torchsparse.utils.collate import sparse_collate, sparse_collate_fn  # noqa: F401  (collate.py:8)
    from torchsparse.utils.quantize import sparse_quantize  # noqa: F401  (semantic_kitti_dataloader.py)
    import torch_scatter.scatter as ts_scatter  # noqa: F401  (hr_module.py:5)
    for name in ("sphash", "sphashquery", "spcount", "spvoxelize", "spdevoxelize", "calc_ti_weights"):
        assert callable(getattr(F, name))
    for name in ("Conv3d", "BatchNorm", "ReLU"):
        assert isinstance(getattr(spnn, name), type)
    assert callable(torchsparse.cat)
    off3 = get_kernel_offsets(3, 1, 1)
    off2 = get_kernel_offsets(2, 2, 1)
    assert off3.shape == (27, 3) and off3[0].tolist() == [-1, -1, -1] and off3[1].tolist() == [0, -1, -1]   # x fastest
    assert off2.shape == (8, 3) and off2[1].tolist() == [0, 0, 2]                                          # z fastest


@pytest.mark.gpu
def test_dataloader_quantize_and_collate_through_the_shim(shim):
    """semantic_kitti_dataloader.py:562-577 + collate.py:403-405 replayed with the shim's names."""
    from torchsparse import SparseTensor
    from torchsparse.utils.collate import sparse_collate
    from torchsparse.utils.quantize import sparse_quantize
    from latte_amd import synth
    from latte_amd.quantize import quantize_scan_gpu
    scans = []
    for b in range(3):
        points, inten = synth.make_scan(8192, b, beams=32)
        pc_ = np.round(points * 20)
        pc_ = (pc_ - pc_.min(0)).astype(np.int64)                                   # augmentation_3d.py:48-56
        _, inds, inverse = sparse_quantize(pc_, return_index=True, return_inverse=True)
        lidar = SparseTensor(feats=torch.cat((torch.from_numpy(points[inds]).float(),
                                              torch.from_numpy(inten[inds]).view(-1, 1).float()), dim=1),
                             coords=torch.from_numpy(pc_[inds]).int())
        # == np.unique on the ravelled key (the algorithm of hr_module.py:58-97 / torchsparse's sparse_quantize)
        key = (pc_[:, 0] * (pc_[:, 1].max() + 1) + pc_[:, 1]) * (pc_[:, 2].max() + 1) + pc_[:, 2]
        _, i_np, inv_np = np.unique(key, return_index=True, return_inverse=True)
        assert np.array_equal(inds, i_np) and np.array_equal(inverse, inv_np.reshape(-1))
        c, f, i2, inv2 = quantize_scan_gpu(torch.from_numpy(points).cuda(), torch.from_numpy(inten).cuda())
        assert np.array_equal(c.cpu().numpy(), pc_[inds]) and np.array_equal(i2.cpu().numpy(), inds)
        assert torch.equal(f.cpu(), lidar.F)
        scans.append(lidar)
    batch = sparse_collate(scans)
    assert batch.C.dtype == torch.int32 and batch.C.shape[1] == 4
    sizes = [s.C.shape[0] for s in scans]
    want_b = torch.repeat_interleave(torch.arange(3, dtype=torch.int32), torch.tensor(sizes))
    assert torch.equal(batch.C[:, 3].cpu(), want_b)                                # batch index is the LAST column
    assert torch.equal(batch.C[:, :3].cpu(), torch.cat([s.C for s in scans], 0).cpu())
    assert torch.equal(batch.F.cpu(), torch.cat([s.F for s in scans], 0).cpu())


@pytest.mark.gpu
def test_point_voxel_bridges_through_the_shim_names(shim, scans_small):
    """torchsparse_utils.py:11-100 restated line by line over torchsparse.nn.functional / get_kernel_offsets /
    SparseTensor / PointTensor from the shim; bit-equal to latte_amd.point_voxel (what latte_amd.spvcnn runs)."""
    import torchsparse.nn.functional as F
    from torchsparse import PointTensor, SparseTensor
    from torchsparse.nn.utils import get_kernel_offsets
    from latte_amd import point_voxel as pv
    from latte_amd import sparse_modules as spnn
    coords = torch.from_numpy(scans_small["coords"]).cuda()
    feats = torch.from_numpy(scans_small["feats"]).cuda()

    # --- initial_voxelize (:11-31)
    z = PointTensor(feats, coords.float())
    new_float_coord = torch.cat([(z.C[:, :3] * 0.05) / 0.05, z.C[:, -1].view(-1, 1)], 1)
    pc_hash = F.sphash(torch.floor(new_float_coord).int())
    sparse_hash = torch.unique(pc_hash)
    idx_query = F.sphashquery(pc_hash, sparse_hash)
    counts = F.spcount(idx_query.int(), len(sparse_hash))
    inserted_coords = torch.round(F.spvoxelize(torch.floor(new_float_coord), idx_query, counts)).int()
    inserted_feat = F.spvoxelize(z.F, idx_query, counts)
    x0 = SparseTensor(inserted_feat, inserted_coords, 1)
    x0.cmaps.setdefault(x0.stride, x0.coords)
    z.additional_features["idx_query"][1] = idx_query
    z.additional_features["counts"][1] = counts
    z.C = new_float_coord
    z2 = PointTensor(feats, coords.float())
    want0 = pv.initial_voxelize(z2, 0.05, 0.05)
    assert torch.equal(x0.C, want0.C) and torch.equal(x0.F, want0.F)
    assert torch.equal(idx_query.int(), z2.additional_features["idx_query"][1].int())
    assert torch.equal(counts, z2.additional_features["counts"][1])

    # a stride-2 level to bridge to (the coordinates spnn.Conv3d(ks=2, stride=2) produces)
    _, oc = spnn.build_kernel_map(want0, 2, 2, 1)
    x1 = SparseTensor(torch.randn(oc.shape[0], 32, device="cuda"), oc, 2)
    x1w = SparseTensor(x1.F, oc, 2)

    # --- voxel_to_point (:64-91)
    off = get_kernel_offsets(2, x1.s, 1, device=z.F.device)
    cell = torch.cat([torch.floor(z.C[:, :3] / x1.s[0]).int() * x1.s[0], z.C[:, -1].int().view(-1, 1)], 1)
    old_hash = F.sphash(cell, off)
    idx8 = F.sphashquery(old_hash, F.sphash(x1.C.to(z.F.device)))
    weights = F.calc_ti_weights(z.C, idx8, scale=x1.s[0]).transpose(0, 1).contiguous()
    idx8 = idx8.transpose(0, 1).contiguous()
    new_feat = F.spdevoxelize(x1.F, idx8, weights)
    zw = pv.voxel_to_point(x1w, z2)
    assert torch.equal(idx8.int(), z2.idx_query[x1w.s].int()) and torch.equal(weights, z2.weights[x1w.s])
    assert torch.equal(new_feat, zw.F)
    assert float(weights.sum(1).min()) > 0.999 and int((idx8 >= 0).sum(1).min()) >= 1

    # --- point_to_voxel (:36-59)
    zf = PointTensor(torch.randn(coords.shape[0], 32, device="cuda"), z.C)
    pc_hash = F.sphash(cell)
    idx_query = F.sphashquery(pc_hash, F.sphash(x1.C))
    counts = F.spcount(idx_query.int(), x1.C.shape[0])
    got = F.spvoxelize(zf.F, idx_query, counts)
    zf2 = PointTensor(zf.F, z2.C)
    want = pv.point_to_voxel(x1w, zf2)
    assert torch.equal(idx_query.int(), zf2.additional_features["idx_query"][x1w.s].int())
    assert torch.allclose(got, want.F, rtol=1e-6, atol=1e-7)      # atomic kernel (no CSR given) vs CSR kernel: same sums, other order
    assert int(counts.sum()) == coords.shape[0] and int(counts.min()) >= 1


@pytest.mark.gpu
def test_reference_style_block_runs_on_shim_modules(shim, scans_small):
    """basic_blocks.py's composition (spnn.Conv3d -> spnn.BatchNorm -> spnn.ReLU in nn.Sequential over SparseTensors, and
    torchsparse.cat) through the shim's module classes == latte_amd.spvcnn's fused block."""
    import torchsparse
    import torchsparse.nn as spnn
    from torchsparse import SparseTensor
    from latte_amd.spvcnn import BasicConvolutionBlock
    coords = torch.from_numpy(scans_small["coords"]).cuda()
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(coords.shape[0], 32, generator=g).cuda()
    torch.manual_seed(4)
    seq = torch.nn.Sequential(spnn.Conv3d(32, 48, kernel_size=3, dilation=1, stride=1), spnn.BatchNorm(48), spnn.ReLU(True)).cuda()
    blk = BasicConvolutionBlock(32, 48, ks=3).cuda()
    blk.net.load_state_dict(seq.state_dict())
    y_seq = seq(SparseTensor(feats, coords, 1))
    y_blk = blk(SparseTensor(feats, coords, 1))
    assert torch.equal(y_seq.C, y_blk.C) and y_seq.s == y_blk.s
    assert torch.allclose(y_seq.F, y_blk.F, rtol=1e-5, atol=1e-6)
    both = torchsparse.cat([y_seq, y_blk])
    assert both.F.shape == (coords.shape[0], 96) and torch.equal(both.C, y_seq.C)
