"""Size-independent properties at BASELINE.json's full sizes (configs[1] nuScenes-like batch, 187 k voxels; one
131 072-point scan of configs[2]), where the CPU oracle is too slow to be the checker."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    from latte_amd import synth
    from latte_amd.workload import build_batch
    w = synth.workload("nuscenes_34k_w5")
    return w, build_batch(w, device="cuda")


def test_quantize_round_trip_and_order(big):
    from latte_amd import synth
    from latte_amd.quantize import quantize_scan_gpu
    for n, beams, full in ((131072, 64, False), (34816, 32, True)):
        p, i = synth.make_scan(n, 3, beams=beams, full_360=full)
        pts = torch.from_numpy(p).cuda()
        c, f, inds, inv = quantize_scan_gpu(pts, torch.from_numpy(i).cuda())
        allc = torch.round(pts * 20)
        allc = (allc - allc.amin(0)).int()
        assert torch.equal(c[inv], allc)                       # unique -> inverse reproduces every point's voxel
        assert torch.equal(allc[inds], c)                      # indices point at a member of the voxel ...
        first = torch.full((c.shape[0],), n, dtype=torch.int64, device="cuda").scatter_reduce(
            0, inv, torch.arange(n, device="cuda"), "amin")
        assert torch.equal(first, inds)                        # ... the FIRST one (np.unique return_index)
        key = (c[:, 0].long() << 40) | (c[:, 1].long() << 20) | c[:, 2].long()
        assert bool((key[1:] > key[:-1]).all())                # lexicographic (x,y,z), strictly increasing = unique


def test_kernel_map_symmetry_and_plan_conv_bitwise(big):
    from latte_amd import ops
    from latte_amd import sparse_modules as spnn
    from latte_amd.point_voxel import initial_voxelize
    from latte_amd.tensor import PointTensor
    w, batch = big
    lidar = batch["lidar"]
    z = PointTensor(lidar.F, lidar.C.float())
    x0 = initial_voxelize(z, 0.05, 0.05)
    assert x0.C.shape[0] == lidar.C.shape[0]                  # input voxels are already unique
    idx = z.additional_features["idx_query"][1]
    assert torch.equal(x0.C[idx.long()], lidar.C)             # point -> voxel map is exact
    kmap, _ = spnn.build_kernel_map(x0, 3, 1, 1)
    res = kmap.results
    K, V = res.shape
    assert torch.equal(res[13], torch.arange(V, dtype=torch.int32, device="cuda"))   # centre offset = identity
    kk, oo = torch.nonzero(res >= 0, as_tuple=True)
    ii = res[kk, oo].long()
    assert torch.equal(res[K - 1 - kk, ii], oo.int())          # submanifold symmetry: (k,o)->i  <=>  (K-1-k,i)->o
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(V, 96, generator=g).cuda()
    pw = ops.pack_weights((torch.randn(27, 96, 96, generator=g) / 50).cuda(), 0)
    a = ops.spconv(feats, pw, res, 96)
    b = ops.spconv_plan(feats, pw, kmap.plan(), 27, V, 96, variant=202)   # single-chain variant: bit for bit
    assert torch.equal(a, b)
    b = ops.spconv_plan(feats, pw, kmap.plan(), 27, V, 96)                # default (split-k): re-associated sum
    assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())
    assert torch.equal(b, ops.spconv_plan(feats, pw, kmap.plan(), 27, V, 96))   # and reproducible
    # linearity (fp32 tolerance): conv(x + y) = conv(x) + conv(y)
    f2 = torch.randn(V, 96, generator=g).cuda()
    lhs = ops.spconv_plan(feats + f2, pw, kmap.plan(), 27, V, 96)
    rhs = b + ops.spconv_plan(f2, pw, kmap.plan(), 27, V, 96)
    assert float((lhs - rhs).norm() / rhs.norm()) < 1e-6
    # strided map: every fine voxel has exactly one parent, parents sorted by (b,x,y,z) and unique
    km2, oc = spnn.build_kernel_map(x0, 2, 2, 1)
    assert int((km2.results >= 0).sum()) == V
    key = (oc[:, 3].long() << 48) | ((oc[:, 0].long() + 32768) << 32) | ((oc[:, 1].long() + 32768) << 16) | (oc[:, 2].long() + 32768)
    assert bool((key[1:] > key[:-1]).all())


def test_batchnorm_and_devoxelize_invariants(big):
    from latte_amd import ops
    w, batch = big
    V = batch["lidar"].C.shape[0]
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(V, 96, generator=g) * 3 + 5).cuda()
    mean, var = ops.bn_stats(x)
    y = ops.bn_apply(x, mean, var, torch.ones(96, device="cuda"), torch.zeros(96, device="cuda"), 1e-5)
    assert float(y.mean(0).abs().max()) < 1e-4 and float((y.var(0, unbiased=False) - 1).abs().max()) < 1e-3


def test_devoxelize_backward_over_cells_is_the_adjoint_at_full_size(big):
    """strides 1, 4, 16 of the 187 k-voxel batch, the maps SPVCNN.prepare builds: lt_devoxelize_cell_bwd against the atomic
    scatter kernel, and <devox(x), g> == <x, devox^T(g)> (the property that does not need an oracle)"""
    from latte_amd import ops
    from latte_amd.spvcnn import SPVCNN
    from latte_amd.tensor import SparseTensor
    w, batch = big
    lidar = batch["lidar"]
    view = SparseTensor(lidar.F, lidar.C, lidar.s)
    net = SPVCNN(cr=0.5).cuda()
    net.prepare(view)
    cache, z = net._batch_cache(view)
    g = torch.Generator().manual_seed(2)
    N = lidar.C.shape[0]
    for s in (1, 4, 16):
        s3 = (s, s, s)
        V = cache["levels"][s].C.shape[0]
        idx8, w8 = z.idx_query[s3], z.weights[s3]
        cells = z.additional_features["v2p_cells"][s3]
        p2v_csr, p2v_idx = z.additional_features["p2v_csr"][s3], z.additional_features["idx_query"][s3]
        assert int((p2v_idx < 0).sum()) == 0                  # every point's cell is a voxel of the level
        gout = torch.randn(N, 32, generator=g).cuda()
        x = torch.randn(V, 32, generator=g).cuda()
        got = ops.devoxelize_cell_bwd(gout, cells, p2v_csr, p2v_idx.int(), idx8.int(), w8)
        ref = ops.devoxelize_bwd(gout, idx8.int(), w8, V)
        assert float((got - ref).norm() / ref.norm()) < 1e-6, s
        lhs = float((ops.devoxelize_fwd(x, idx8.int(), w8).double() * gout.double()).sum())
        rhs = float((x.double() * got.double()).sum())
        assert abs(lhs - rhs) < 1e-6 * max(abs(lhs), 1.0), (s, lhs, rhs)
        assert torch.equal(got, ops.devoxelize_cell_bwd(gout, cells, p2v_csr, p2v_idx.int(), idx8.int(), w8))   # no atomics ran


def test_full_size_step_is_finite_and_st_voxels_conserve_points(big):
    from latte_amd.hr_module import HR_Block
    from latte_amd.workload import hr_cfg_of
    w, batch = big
    hr = HR_Block(**hr_cfg_of(w))
    n = sum(p.shape[0] for p in batch["ori_points"])
    g = torch.Generator().manual_seed(5)
    la = (3 * torch.randn(n, w["C"], generator=g)).cuda()
    lh = (3 * torch.randn(n, w["C"], generator=g)).cuda()
    extra = {k: batch[k] for k in ("pose", "scene_name", "scan_ID", "img_indices", "ori_points")}
    out = hr(dict(seg_logit=la, **extra), "3D", dict(seg_logit=lh, **extra), "3D")
    B = len(batch["pose"])
    assert len(out["ac_logits"]) == B
    for a, t, av, tv in zip(out["ac_logits"], out["tr_logits"], out["ac_voxels"], out["tr_voxels"]):
        assert a.shape == t.shape and torch.equal(av, tv)       # matched voxel sets are identical and aligned (Q3)
        assert torch.allclose(a.sum(1), torch.ones_like(a[:, 0]), atol=1e-5)   # voxel means of softmax rows sum to 1
        assert torch.allclose(t.sum(1), torch.ones_like(t[:, 0]), atol=1e-5)
        key = (av[:, 0].long() << 32) | (av[:, 1].long() << 16) | av[:, 2].long()
        assert bool((key[1:] > key[:-1]).all())                 # lexicographic voxel order (np.unique on ravel keys)
    ety = out["hs_pc_ety"]
    assert ety.shape[0] == n and bool(torch.isfinite(ety).all()) and float(ety.min()) >= 0.0 and float(ety.max()) <= 1.0 + 1e-5
    # every hist point inside max_range belongs to a window of its own anchor frame -> has an ST entropy
    pts = torch.cat(batch["ori_points"], 0)
    inside = pts.norm(dim=1) <= w["R"]
    assert bool((ety[inside] > 0).all())
