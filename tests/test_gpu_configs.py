"""BASELINE.json configs[0] (against the CPU oracle at its own size) and configs[2], [3], [4] on the GPU at their full sizes, through size-independent properties (the CPU oracle is
too slow to be the checker there; small-size oracle parity of the same code is in test_gpu_ops / _spvcnn / _stvoxel / _step):

  configs[2]  A2D2 -> SemanticKITTI: 6 x 131 072-point scans, 5 cm input voxels, ST voxels 0.1 m, R 40 m, 10 classes
              (reference knobs latte/config/xmuda.py:293-298, configs/a2d2_semantic_kitti/latte.yaml)
  configs[3]  64 scans per step = 8 sequences x 8 frames, sequences sharded over the GPUs
  configs[4]  3^3 sparse conv over ~1 M voxels, Cin = Cout in {16, 32, 64, 96, 128, 256}
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _x0_of(lidar):
    from latte_amd.point_voxel import initial_voxelize
    from latte_amd.tensor import PointTensor
    z = PointTensor(lidar.F, lidar.C.float())
    return initial_voxelize(z, 0.05, 0.05), z


# ---------------------------------------------------------------------------------------------------- configs[0]
def test_config0_one_32k_scan_step_matches_oracle():
    """BASELINE configs[0] at its own shape: ONE synthetic 32 768-point scan (B = 1), Synthia -> SemanticKITTI knobs
    (configs/synthia_semantic_kitti/latte.yaml: 9 classes, W 3, R 40, vs 0.2, conf_perc 0.9), SPVCNN cr = 1.0 -- the
    configuration BASELINE.json names as the CPU-reference case, so the CPU oracle is the checker at full size: every
    index structure of the backbone bit-exact, logits and the four losses of the whole TTA iteration to 1e-3.  B = 1 is its
    own code path: a single anchor whose window is itself (Wmax = 1), one anchor bit in the ST keys.  (The ST-voxel half
    of this shape is also pinned to the reference's own code: tests/golden/hr_b1_32k.npz in test_gpu_stvoxel.py.)"""
    from latte_amd import sparse_modules as spnn
    from latte_amd import synth
    from latte_amd.point_voxel import initial_voxelize
    from latte_amd.spvcnn import Net3DSeg
    from latte_amd.tensor import PointTensor, SparseTensor
    from latte_amd.tta_step import LatteStep
    from latte_amd.workload import build_batch, hr_cfg_of
    from oracle import step_ref
    from oracle.spvcnn_ref import SPVCNNRef
    w = synth.workload("synthia_kitti_32k_b1")
    assert w["B"] == 1 and w["n"] == 32768 and w["C"] == 9 and w["conf"] == 0.9
    torch.manual_seed(1)
    model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": 1.0})
    g = torch.Generator().manual_seed(2)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.weight.data = 0.5 + torch.rand(mod.weight.shape, generator=g)
            mod.bias.data = 0.2 * torch.randn(mod.bias.shape, generator=g)
    step = LatteStep(model.cuda(), w["C"], hr_cfg_of(w))
    batch = build_batch(w, device="cuda")
    lidar = batch["lidar"]
    assert len(batch["ori_points"]) == 1 and batch["ori_points"][0].shape[0] == 32768 and int(lidar.C[:, 3].max()) == 0
    coords = lidar.C.cpu().numpy()

    # ---- the backbone's index structures against the oracle, every level
    sd_t = {k: v.detach().cpu() for k, v in step.ema_model.state_dict().items()}
    ref = SPVCNNRef(sd_t)
    with torch.no_grad():
        want = ref(coords, lidar.F.cpu())
        got = step.ema_model(batch)
    z = PointTensor(lidar.F, lidar.C.float())
    t = initial_voxelize(z, 0.05, 0.05)
    assert np.array_equal(t.C.cpu().numpy(), ref.traced["x0_coords"])
    assert np.array_equal(z.additional_features["idx_query"][1].cpu().numpy(), ref.traced["idx_query_1"])
    for lvl, name in zip((1, 2, 4, 8, 16), ("x1", "x2", "x3", "x4", None)):
        km3, _ = spnn.build_kernel_map(t, 3, 1, 1)
        nb, ns, _, _ = ref.traced["kmaps"][(lvl, 3, 1)]
        assert np.array_equal(km3[0].cpu().numpy(), nb) and np.array_equal(km3[1].cpu().numpy(), ns), lvl
        if name is None:
            break
        km2, oc = spnn.build_kernel_map(t, 2, 2, 1)
        nb, ns, _, want_oc = ref.traced["kmaps"][(lvl, 2, 2)]
        assert np.array_equal(oc.cpu().numpy(), want_oc) and np.array_equal(oc.cpu().numpy(), ref.traced[name + "_coords"])
        assert np.array_equal(km2[0].cpu().numpy(), nb) and np.array_equal(km2[1].cpu().numpy(), ns), lvl
        nt = SparseTensor(torch.zeros(oc.shape[0], 4, device="cuda"), oc, 2 * lvl)
        nt.cmaps, nt.kmaps = t.cmaps, t.kmaps
        nt.cmaps[nt.stride] = oc
        t = nt
    for k in ("feats", "seg_logit", "seg_logit2"):
        a, b = got[k].cpu().double(), want[k].double()
        assert float((a - b).norm() / b.norm()) < 1e-4, k
        assert torch.allclose(got[k].cpu(), want[k], rtol=1e-3, atol=1e-3 * float(want[k].abs().max())), k

    # ---- the whole iteration against the oracle's
    sd_s = {k: v.detach().cpu() for k, v in step.model.state_dict().items()}
    losses = {k: float(v) for k, v in step.forward_backward(batch).items()}
    want_l, _, g2d, aux = step_ref.tta_step_ref(
        sd_s, sd_t, coords, lidar.F.cpu(), [i.cpu().numpy() for i in batch["inverse_map"]],
        [int(i.shape[0]) for i in batch["indices"]], [p.cpu() for p in batch["ori_points"]],
        [m.cpu() for m in batch["pc2img_idx"]], [p.numpy() for p in batch["pose"]], batch["scene_name"], batch["scan_ID"],
        batch["logit_2d_student"].detach().cpu(), batch["logit_2d_teacher"].cpu(), hr_cfg_of(w))
    for k in ("hr_23", "hr_32"):
        assert abs(losses[k] - want_l[k]) < 1e-3 * abs(want_l[k]), (k, losses[k], want_l[k])
    for k in ("pl_2d", "pl_3d"):   # discrete pseudo labels: a flipped label moves the loss by ~1/N
        assert abs(losses[k] - want_l[k]) < 2e-3 * abs(want_l[k]), (k, losses[k], want_l[k])
    g = batch["logit_2d_student"].grad.cpu()
    assert float((g - g2d).norm() / g2d.norm()) < 2e-3
    # the single anchor matched voxels on both HR calls, and the ST entropy reached every in-range point
    assert all(len(aux[h]["ac_logits"]) == 1 and aux[h]["ac_logits"][0].shape[0] > 1000 for h in ("hr23", "hr32"))
    # ---- and it steps: prefetch one iteration ahead, parameters move, everything finite
    batch["logit_2d_student"].grad = None
    before = [p.detach().clone() for p in step.params]
    l1 = step.step(batch, next_batch=batch)
    l2 = step.step(batch)
    step.close()
    assert all(np.isfinite(float(v)) for l in (l1, l2) for v in l.values())
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, step.params))


# ---------------------------------------------------------------------------------------------------- configs[2]
@pytest.fixture(scope="module")
def a2d2():
    from latte_amd import synth
    from latte_amd.workload import build_batch
    w = synth.workload("a2d2_kitti_128k")
    return w, build_batch(w, device="cuda")


def test_a2d2_kernel_maps_and_plan_conv(a2d2):
    from latte_amd import ops
    from latte_amd import sparse_modules as spnn
    w, batch = a2d2
    lidar = batch["lidar"]
    assert lidar.C.shape[0] > 250_000                                   # 6 x 131 072 points -> > 250 k 5 cm voxels
    x0, z = _x0_of(lidar)
    assert x0.C.shape[0] == lidar.C.shape[0]
    assert torch.equal(x0.C[z.additional_features["idx_query"][1].long()], lidar.C)
    kmap, _ = spnn.build_kernel_map(x0, 3, 1, 1)
    res = kmap.results
    K, V = res.shape
    assert torch.equal(res[13], torch.arange(V, dtype=torch.int32, device="cuda"))
    kk, oo = torch.nonzero(res >= 0, as_tuple=True)
    assert torch.equal(res[K - 1 - kk, res[kk, oo].long()], oo.int())   # (k,o)->i  <=>  (K-1-k,i)->o
    # every neighbour really is the voxel at coords + offset
    from latte_amd.functional import get_kernel_offsets
    off = get_kernel_offsets(3, 1, 1, device="cuda")
    sel = torch.randperm(kk.numel(), device="cuda")[:200_000]
    want = x0.C[oo[sel]].clone()
    want[:, :3] += off[kk[sel]]
    assert torch.equal(x0.C[res[kk[sel], oo[sel]].long()], want)
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(V, 96, generator=g).cuda()
    pw = ops.pack_weights((torch.randn(27, 96, 96, generator=g) / 50).cuda(), 0)
    a = ops.spconv(feats, pw, res, 96)                                   # table kernel, one chain per output
    assert torch.equal(a, ops.spconv_plan(feats, pw, kmap.plan(), 27, V, 96, variant=202))   # single chain: bit for bit
    b = ops.spconv_plan(feats, pw, kmap.plan(), 27, V, 96)               # default kernel: re-associated, reproducible
    assert float((a - b).abs().max()) <= 4e-6 * float(a.abs().max())   # fp32 re-association of W sub-chains: a few ulp of the largest partial sum
    assert torch.equal(b, ops.spconv_plan(feats, pw, kmap.plan(), 27, V, 96))
    # the coordinate pyramid: one parent per voxel, parents unique and sorted by (b, x, y, z), down to stride 16
    t = x0
    from latte_amd.tensor import SparseTensor
    for s in (1, 2, 4, 8):
        km2, oc = spnn.build_kernel_map(t, 2, 2, 1)
        assert int((km2.results >= 0).sum()) == t.C.shape[0]
        key = (oc[:, 3].long() << 48) | ((oc[:, 0].long() + 32768) << 32) | ((oc[:, 1].long() + 32768) << 16) | (oc[:, 2].long() + 32768)
        assert bool((key[1:] > key[:-1]).all())
        assert bool((oc[:, :3] % (2 * s) == 0).all())
        nt = SparseTensor(t.feats[:0], oc, 2 * s)
        nt.cmaps, nt.kmaps = t.cmaps, t.kmaps
        nt.cmaps[nt.stride] = oc
        t = nt


def test_a2d2_hr_block_at_10cm_voxels_conserves_points(a2d2):
    from latte_amd.hr_module import HR_Block
    from latte_amd.workload import hr_cfg_of
    w, batch = a2d2
    assert w["vs"] == 0.1 and w["R"] == 40 and w["C"] == 10 and w["W"] == 3
    hr = HR_Block(**hr_cfg_of(w))
    n = sum(p.shape[0] for p in batch["ori_points"])
    g = torch.Generator().manual_seed(5)
    la = (3 * torch.randn(n, w["C"], generator=g)).cuda()
    lh = (3 * torch.randn(n, w["C"], generator=g)).cuda()
    extra = {k: batch[k] for k in ("pose", "scene_name", "scan_ID", "img_indices", "ori_points")}
    out = hr(dict(seg_logit=la, **extra), "3D", dict(seg_logit=lh, **extra), "3D")
    lim = 2 * (int(np.ceil(w["R"] / w["vs"])) + 1)
    for a, t, av, tv, ap, tp in zip(out["ac_logits"], out["tr_logits"], out["ac_voxels"], out["tr_voxels"],
                                    out["ac_p2v_indices"], out["tr_p2v_indices"]):
        assert a.shape == t.shape and torch.equal(av, tv) and a.shape[0] > 1000
        assert torch.allclose(a.sum(1), torch.ones_like(a[:, 0]), atol=1e-5)
        assert torch.allclose(t.sum(1), torch.ones_like(t[:, 0]), atol=1e-5)
        key = (av[:, 0].long() << 32) | (av[:, 1].long() << 16) | av[:, 2].long()
        assert bool((key[1:] > key[:-1]).all())
        assert int(av.min()) >= 0 and int(av.max()) <= lim                # Q1: coordinates shifted by ceil(R/vs)+1
        assert int(ap.min()) >= 0 and int(ap.max()) < n and int(tp.max()) < n
    # the anchor's representative point of a matched voxel lies in that voxel (anchor frame = identity transform)
    pts = torch.cat(batch["ori_points"], 0)
    av0, ap0 = out["ac_voxels"][0], out["ac_p2v_indices"][0]
    cell = torch.floor(pts[ap0].double() / w["vs"]).int() + (int(np.ceil(w["R"] / w["vs"])) + 1)
    assert torch.equal(cell, av0)
    ety = out["hs_pc_ety"]
    assert ety.shape[0] == n and bool(torch.isfinite(ety).all()) and float(ety.min()) >= 0.0
    inside = pts.norm(dim=1) <= w["R"]
    assert bool((ety[inside] > 0).all())
    assert out["tr_conf_mask"].dtype == torch.bool


def test_a2d2_full_tta_step_is_finite_and_moves_only_norm_parameters(a2d2):
    from latte_amd.spvcnn import Net3DSeg
    from latte_amd.tta_step import LatteStep
    from latte_amd.workload import hr_cfg_of
    w, batch = a2d2
    torch.manual_seed(1)
    model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": 1.0}).cuda()
    step = LatteStep(model, w["C"], hr_cfg_of(w))
    before = {k: v.detach().clone() for k, v in step.model.state_dict().items()}
    l1 = {k: float(v) for k, v in step.step(batch).items()}
    l2 = {k: float(v) for k, v in step.step(batch, next_batch=batch).items()}
    l3 = {k: float(v) for k, v in step.step(batch).items()}
    step.close()
    for l in (l1, l2, l3):
        assert all(np.isfinite(v) and v > 0 for v in l.values()), l
    changed = {k for k, v in step.model.state_dict().items() if not torch.equal(v, before[k])}
    norm_keys = {f"{mn}.{pn}" for mn, mod in step.model.named_modules() if "norm" in type(mod).__name__.lower()
                 for pn, _ in mod.named_parameters()}
    assert changed and changed <= norm_keys
    assert all(bool(torch.isfinite(p.grad).all()) for p in step.params if p.grad is not None)
    # the same batch seen twice by one model: identical logits (no state is carried between forwards)
    with torch.no_grad():
        a = step.ema_model(batch)["seg_logit"]
        batch["lidar"]._lt_cache = None
        b = step.ema_model(batch)["seg_logit"]
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------- configs[3]
def test_multiseq_64_step_keeps_sequences_apart():
    """64 scans = 8 sequences x 8 frames in ONE step on one GPU: ST voxels never mix sequences (the property that makes the
    sequence shard of bench.py --gpus N exact for the ST-voxel stage), and the whole step is finite."""
    from latte_amd import synth
    from latte_amd.hr_module import HR_Block
    from latte_amd.spvcnn import Net3DSeg
    from latte_amd.tta_step import LatteStep
    from latte_amd.workload import build_batch, hr_cfg_of
    w = synth.workload("multiseq_64")
    batch = build_batch(w, device="cuda")
    B, L = w["B"], w["seq_len"]
    assert B == 64 and L == 8 and len(set(batch["scene_name"])) == 8
    hr = HR_Block(**hr_cfg_of(w))
    extra = {k: batch[k] for k in ("pose", "scene_name", "scan_ID", "img_indices", "ori_points")}
    geo = hr.geometry(extra, "3D", extra, "3D")
    sizes = [int(p.shape[0]) for p in batch["ori_points"]]
    start = np.concatenate([[0], np.cumsum(sizes)])
    scan_of = torch.bucketize(torch.arange(int(start[-1]), device="cuda"),
                              torch.as_tensor(start[1:], device="cuda"), right=True)
    # every hist element (a point of a window frame, keyed under an anchor) comes from the anchor's own sequence and window
    h_src = geo["h_src"].long()
    seg_anchor = torch.repeat_interleave(torch.arange(B, device="cuda"), torch.as_tensor(
        [sum(1 for j in range(B) if j // L == i // L and abs(j % L - i % L) <= w["W"] // 2) * sizes[0] for i in range(B)],
        device="cuda"))
    assert seg_anchor.numel() == h_src.numel()
    frame = scan_of[h_src]
    assert bool((frame // L == seg_anchor // L).all())
    assert bool(((frame % L) - (seg_anchor % L)).abs().max() <= w["W"] // 2)
    # matched voxels: row_anchor partitions by scene, per-anchor counts add up
    ra = geo["row_anchor"]
    assert bool((ra[1:] >= ra[:-1]).all()) and int(ra.max()) == B - 1
    assert torch.equal(torch.bincount(ra, minlength=B).cpu(), torch.tensor(geo["per_anchor"]))
    assert bool((scan_of[geo["ac_p2v"]] == ra).all())                      # anchor voxel's point lies in the anchor scan
    assert bool((scan_of[geo["tr_p2v"]] // L == ra // L).all())             # hist voxel's point lies in the same sequence
    # one whole step on the 64-scan batch
    torch.manual_seed(1)
    model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": 1.0}).cuda()
    step = LatteStep(model, w["C"], hr_cfg_of(w))
    losses = {k: float(v) for k, v in step.step(batch).items()}
    step.close()
    assert all(np.isfinite(v) and v > 0 for v in losses.values()), losses


def test_sequence_shard_is_the_unsharded_st_voxel_stage():
    """HR_Block over sequences [0,4) and [4,8) separately == the matching slices of HR_Block over all 8 (no halo, no
    data-path collective needed): bit-exact voxels / indices, same probabilities."""
    from latte_amd import synth
    from latte_amd.hr_module import HR_Block
    from latte_amd.tta_step import shard_sequences
    from latte_amd.workload import hr_cfg_of
    w = dict(synth.workload("multiseq_64"), n=4096)
    L, S = w["seq_len"], 8
    g = torch.Generator().manual_seed(11)
    pts = [torch.from_numpy(synth.make_scan(w["n"], b, beams=32, full_360=True)[0]).cuda() for b in range(S * L)]
    poses = [torch.from_numpy(synth.make_pose(b % L)) for b in range(S * L)]
    scenes = [f"scene-{1000 + b // L:04d}" for b in range(S * L)]
    ids = [b % L for b in range(S * L)]
    la = (3 * torch.randn(S * L * w["n"], w["C"], generator=g)).cuda()
    lh = (3 * torch.randn(S * L * w["n"], w["C"], generator=g)).cuda()
    hr = HR_Block(**hr_cfg_of(w))

    def run(b0, b1):
        sl = slice(b0, b1)
        ex = dict(pose=poses[sl], scene_name=scenes[sl], scan_ID=ids[sl], ori_points=pts[sl],
                  img_indices=[torch.empty(w["n"], 2)] * (b1 - b0))
        p0, p1 = b0 * w["n"], b1 * w["n"]
        return hr(dict(seg_logit=la[p0:p1], **ex), "3D", dict(seg_logit=lh[p0:p1], **ex), "3D")

    full = run(0, S * L)
    for rank in range(2):
        s0, ns = shard_sequences(S, 2, rank)
        b0, b1 = s0 * L, (s0 + ns) * L
        part = run(b0, b1)
        for i in range(b1 - b0):
            assert torch.equal(part["ac_voxels"][i], full["ac_voxels"][b0 + i])
            assert torch.equal(part["ac_p2v_indices"][i] + b0 * w["n"], full["ac_p2v_indices"][b0 + i])
            assert torch.equal(part["tr_p2v_indices"][i] + b0 * w["n"], full["tr_p2v_indices"][b0 + i])
            assert torch.equal(part["ac_logits"][i], full["ac_logits"][b0 + i])
            assert torch.equal(part["tr_logits"][i], full["tr_logits"][b0 + i])
        assert torch.equal(part["hs_pc_ety"], full["hs_pc_ety"][b0 * w["n"]:b1 * w["n"]])


# ---------------------------------------------------------------------------------------------------- configs[4]
@pytest.fixture(scope="module")
def conv_1m():
    from latte_amd import sparse_modules as spnn
    from latte_amd.workload import build_conv_1m
    x0 = build_conv_1m(device="cuda")
    kmap, _ = spnn.build_kernel_map(x0, 3, 1, 1)
    return x0, kmap


@pytest.mark.parametrize("C", [16, 32, 64, 96, 128, 256])
def test_conv_1m_default_kernel_against_table_kernel(conv_1m, C):
    from latte_amd import ops
    x0, kmap = conv_1m
    res = kmap.results
    K, V = res.shape
    assert K == 27 and V >= 1_000_000
    g = torch.Generator().manual_seed(C)
    feats = torch.randn(V, C, generator=g).cuda()
    w = (torch.randn(27, C, C, generator=g) / (27 * C) ** 0.5).cuda()
    pw = ops.pack_weights(w, 0)
    a = ops.spconv(feats, pw, res, C)                                     # plain table kernel (one fmaf chain per output)
    b = ops.spconv_plan(feats, pw, kmap.plan(), 27, V, C)                 # the default (hot-path) kernel
    assert float((a - b).abs().max()) <= 4e-6 * float(a.abs().max())   # fp32 re-association of W sub-chains: a few ulp of the largest partial sum
    assert torch.equal(b, ops.spconv_plan(feats, pw, kmap.plan(), 27, V, C))           # bit-reproducible
    # a sample of output rows against fp64 gather-GEMM arithmetic on the host side of the test
    rows = torch.randint(0, V, (256,), generator=g).cuda()
    nb = res[:, rows].long()                                              # [27, 256]
    fin = torch.where((nb >= 0)[..., None], feats[nb.clamp(min=0)], torch.zeros((), device="cuda")).double()
    want = torch.einsum("krc,kcd->rd", fin, w.double())
    assert float((b[rows].double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
