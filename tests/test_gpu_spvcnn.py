"""End-to-end parity of the 3-D backbone (SPVCNN + heads) on the HIP path against the CPU oracle
(oracle/spvcnn_ref.py) with identical weights.  Index structures bit-exact, features/logits 1e-3 relative."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-3


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def make_model(cr, num_classes=9, seed=1):
    from latte_amd.spvcnn import Net3DSeg
    torch.manual_seed(seed)
    m = Net3DSeg(num_classes, True, "SPVCNN", {"cr": cr})
    # non-trivial BN affine parameters so gradient/scale errors are visible
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.weight.data = 0.5 + torch.rand(mod.weight.shape, generator=g)
            mod.bias.data = 0.2 * torch.randn(mod.bias.shape, generator=g)
    m.train()
    m.net_3d.dropout.p = 0.0
    return m


@pytest.mark.parametrize("cr", [0.5, 1.0])
def test_spvcnn_forward_matches_oracle(scans_small, cr):
    from latte_amd.tensor import SparseTensor
    from oracle.spvcnn_ref import SPVCNNRef
    m = make_model(cr)
    ref = SPVCNNRef(m.state_dict())
    coords, feats = scans_small["coords"], torch.from_numpy(scans_small["feats"])
    with torch.no_grad():
        want = ref(coords, feats)
    mg = m.cuda()
    lidar = SparseTensor(feats.cuda(), torch.from_numpy(coords).cuda(), 1)
    with torch.no_grad():
        got = mg({"lidar": lidar})
    # bit-exact index structures: voxel coords per level, kernel maps, voxel<->point maps
    cm = lidar_cmaps = got_cmaps = None
    x0 = ref.traced["x0_coords"]
    # cmaps live on the tensors created inside forward; rebuild them through the public functions
    from latte_amd.point_voxel import initial_voxelize
    from latte_amd.tensor import PointTensor
    from latte_amd import sparse_modules as spnn
    z = PointTensor(lidar.F, lidar.C.float())
    t0 = initial_voxelize(z, 0.05, 0.05)
    assert np.array_equal(t0.C.cpu().numpy(), x0)
    assert np.array_equal(z.additional_features["idx_query"][1].cpu().numpy(), ref.traced["idx_query_1"])
    t = t0
    for lvl, name in zip((1, 2, 4, 8, 16), ("x1", "x2", "x3", "x4", None)):
        km3, _ = spnn.build_kernel_map(t, 3, 1, 1)
        nb, ns, sizes, _oc = ref.traced["kmaps"][(lvl, 3, 1)]
        assert np.array_equal(km3[0].cpu().numpy(), nb) and np.array_equal(km3[1].cpu().numpy(), ns)
        if name is None:      # stride 16: the coarsest level has a 3^3 map only
            break
        km2, oc = spnn.build_kernel_map(t, 2, 2, 1)
        nb, ns, sizes, want_oc = ref.traced["kmaps"][(lvl, 2, 2)]
        assert np.array_equal(oc.cpu().numpy(), want_oc) and np.array_equal(oc.cpu().numpy(), ref.traced[name + "_coords"])
        assert np.array_equal(km2[0].cpu().numpy(), nb) and np.array_equal(km2[1].cpu().numpy(), ns)
        from latte_amd.tensor import SparseTensor as ST
        nt = ST(torch.zeros(oc.shape[0], 4, device="cuda"), oc, 2 * lvl)
        nt.cmaps, nt.kmaps = t.cmaps, t.kmaps
        nt.cmaps[nt.stride] = oc
        t = nt
    for k in ("feats", "seg_logit", "seg_logit2"):
        e = rel_err(got[k].cpu(), want[k])
        assert e < 1e-4, (k, e)
        assert torch.allclose(got[k].cpu(), want[k], rtol=RTOL, atol=1e-3 * float(want[k].abs().max()))


def test_spvcnn_bn_gradients_match_oracle(scans_small):
    from latte_amd.tensor import SparseTensor
    from oracle.spvcnn_ref import SPVCNNRef
    m = make_model(0.5)
    sd = m.state_dict()
    ref = SPVCNNRef(sd)
    names = [f"{mn}.{pn}"[len("net_3d."):] for mn, mod in m.named_modules() if isinstance(mod, torch.nn.BatchNorm1d)
             for pn, _ in mod.named_parameters()]
    ref_params = ref.parameters_requiring_grad(names)
    coords, feats = scans_small["coords"], torch.from_numpy(scans_small["feats"])
    g = torch.Generator().manual_seed(0)
    want = ref(coords, feats)
    tgt = torch.randint(0, 9, (coords.shape[0],), generator=g)
    loss_ref = torch.nn.functional.cross_entropy(want["seg_logit"], tgt) + want["seg_logit2"].square().mean()
    loss_ref.backward()
    # Latte trains norm layers only (latte/common/utils/ema_util.py:6-41)
    mg = m.cuda()
    mg.requires_grad_(False)
    for mod in mg.modules():
        if "norm" in type(mod).__name__.lower():
            mod.requires_grad_(True)
    lidar = SparseTensor(feats.cuda(), torch.from_numpy(coords).cuda(), 1)
    got = mg({"lidar": lidar})
    loss = torch.nn.functional.cross_entropy(got["seg_logit"], tgt.cuda()) + got["seg_logit2"].square().mean()
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * abs(float(loss_ref))
    errs = []
    for n, pr in zip(names, ref_params):
        pg = dict(mg.named_parameters())["net_3d." + n].grad
        assert pg is not None, n
        e = float((pg.cpu().double() - pr.grad.double()).norm() / (pr.grad.double().norm() + 1e-12))
        errs.append((e, n, float(pr.grad.norm())))
    errs.sort(reverse=True)
    print("BN-grad rel errs (worst 8):", errs[:8], "median", errs[len(errs) // 2][0])
    # Measured (scratch/dbg_flip.py): the forward differs from the oracle by ~1e-6, which flips the sign of 2-4 of
    # the ~1e7 ReLU inputs (|x| < 1e-5).  ONE flipped element changes the gradient flowing through that layer by
    # ~1/sqrt(rows*channels) = 2e-3 of its norm and every layer below inherits it, so whole-network gradient parity
    # is bounded by flips, not by kernel arithmetic; each backward kernel alone matches to 1e-7 (test_gpu_ops.py).
    assert errs[len(errs) // 2][0] < 1e-2
    assert errs[0][0] < 3e-2, errs[0]


def test_checkpoint_keys_follow_reference_names():
    """reference checkpoints store torchsparse module names (net_3d.stem.0.kernel, ...):
    latte/common/utils/checkpoint.py:39-87, latte/models/spvcnn.py:29-111."""
    m = make_model(1.0)
    sd = m.state_dict()
    for k, shape in [("net_3d.stem.0.kernel", (27, 4, 32)), ("net_3d.stage1.0.net.0.kernel", (8, 32, 32)),
                     ("net_3d.stage2.1.downsample.0.kernel", (32, 64)), ("net_3d.up1.0.net.0.kernel", (8, 256, 256)),
                     ("net_3d.up1.1.0.net.0.kernel", (27, 384, 256)), ("net_3d.up4.1.1.net.3.kernel", (27, 96, 96)),
                     ("net_3d.point_transforms.0.0.weight", (256, 32)), ("net_3d.stem.1.running_mean", (32,)),
                     ("linear.weight", (9, 96)), ("linear2.bias", (9,))]:
        assert k in sd and tuple(sd[k].shape) == shape, k


def test_spvcnn_bn_gradients_match_oracle_to_1e3_under_one_relu_pattern(scans_small):
    """Whole-network gradient parity at the north_star bar (1e-3, median AND worst).

    The forward passes agree to ~1e-6, which is enough to flip the sign of a handful of ReLU inputs that are zero to fp32
    rounding; one flipped element changes every gradient below it by ~1/sqrt(rows*channels) of its norm, which is what
    limits test_spvcnn_bn_gradients_match_oracle.  Here both facts are tested instead of told:
      (1) the ReLU activity patterns of the two forwards differ only at such elements (|input| <= 1e-4 of the tensor's
          scale, a vanishing fraction of all activations);
      (2) with the oracle's backward taking the GPU forward's branch at every ReLU (oracle.relu_masks), every BatchNorm
          gradient matches within 1e-3 relative -- what is left is kernel arithmetic."""
    import latte_amd.spvcnn as spv
    from latte_amd.tensor import SparseTensor
    from oracle.spvcnn_ref import SPVCNNRef
    m = make_model(0.5)
    sd = m.state_dict()
    names = [f"{mn}.{pn}"[len("net_3d."):] for mn, mod in m.named_modules() if isinstance(mod, torch.nn.BatchNorm1d)
             for pn, _ in mod.named_parameters()]
    coords, feats = scans_small["coords"], torch.from_numpy(scans_small["feats"])
    g = torch.Generator().manual_seed(0)
    tgt = torch.randint(0, 9, (coords.shape[0],), generator=g)

    def loss_of(out, t):
        return torch.nn.functional.cross_entropy(out["seg_logit"], t) + out["seg_logit2"].square().mean()

    # GPU run, recording the output of every BN(+residual)+ReLU
    mg = m.cuda()
    mg.requires_grad_(False)
    for mod in mg.modules():
        if "norm" in type(mod).__name__.lower():
            mod.requires_grad_(True)
    lidar = SparseTensor(feats.cuda(), torch.from_numpy(coords).cuda(), 1)
    spv.ACT_TRACE = {}
    try:
        got = mg({"lidar": lidar})
    finally:
        trace, spv.ACT_TRACE = spv.ACT_TRACE, None
    loss = loss_of(got, tgt.cuda())
    loss.backward()
    name_of = {id(mod): mn[len("net_3d."):] for mn, mod in mg.named_modules() if mn.startswith("net_3d.")}
    gpu_act = {name_of[k]: v.cpu() for k, v in trace.items()}

    # (1) free-running oracle: where do the activity patterns differ?
    ref = SPVCNNRef(sd)
    with torch.no_grad():
        ref(coords, feats)
    relu_names = sorted(ref.pre.keys())
    assert set(relu_names) <= set(gpu_act.keys()), set(relu_names) - set(gpu_act.keys())
    total = flips = 0
    for n in relu_names:
        pre = ref.pre[n]
        differ = (pre > 0) != (gpu_act[n] > 0)
        total += pre.numel()
        flips += int(differ.sum())
        if differ.any():
            assert float(pre[differ].abs().max()) <= 1e-4 * float(pre.abs().max()), n
    assert flips <= 1e-5 * total, (flips, total)
    print(f"ReLU inputs: {total}, sign flips between the oracle and the GPU forward: {flips}")

    # (2) oracle backward under the GPU forward's activity pattern
    ref = SPVCNNRef(sd)
    ref.relu_masks = {n: gpu_act[n] > 0 for n in relu_names}
    ref_params = ref.parameters_requiring_grad(names)
    loss_ref = loss_of(ref(coords, feats), tgt)
    loss_ref.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * abs(float(loss_ref))
    params = dict(mg.named_parameters())
    errs = []
    for n, pr in zip(names, ref_params):
        pg = params["net_3d." + n].grad.cpu().double()
        errs.append((float((pg - pr.grad.double()).norm() / (pr.grad.double().norm() + 1e-12)), n))
    errs.sort(reverse=True)
    print("BN-grad rel errs under one ReLU pattern (worst 4):", errs[:4], "median", errs[len(errs) // 2][0])
    assert errs[len(errs) // 2][0] < RTOL
    assert errs[0][0] < RTOL, errs[0]


def test_bf16x3_conv_mode_stays_inside_the_parity_bar(scans_small):
    """Opt-in bf16x3 split-precision convs (lt_conv_set_precision(1)): the backbone's logits against the CPU oracle stay
    inside the 1e-3 bar (measured ~1e-5), per-conv error vs the fp32 kernel ~5e-6, backward-input included; switching back
    restores the bit-exact fp32 path."""
    from latte_amd import ops
    from latte_amd import functional as LF
    from latte_amd.tensor import SparseTensor
    from oracle.spvcnn_ref import SPVCNNRef
    m = make_model(1.0)
    ref = SPVCNNRef(m.state_dict())
    coords, feats = scans_small["coords"], torch.from_numpy(scans_small["feats"])
    with torch.no_grad():
        want = ref(coords, feats)
    mg = m.cuda()
    lidar = SparseTensor(feats.cuda(), torch.from_numpy(coords).cuda(), 1)
    with torch.no_grad():
        base = mg({"lidar": lidar})["seg_logit"].clone()
    try:
        ops.set_conv_precision("bf16x3")
        assert ops.conv_uses_bf16(96, 96) and not ops.conv_uses_bf16(32, 32) and not ops.conv_uses_bf16(48, 48)
        with torch.no_grad():
            got = mg({"lidar": lidar})
        for k in ("feats", "seg_logit", "seg_logit2"):
            e = rel_err(got[k].cpu(), want[k])
            assert e < 2e-4, (k, e)
        assert not torch.equal(got["seg_logit"], base)            # a different arithmetic really ran
        # one conv, forward and backward-input, against the fp32 kernel
        from latte_amd import sparse_modules as spnn
        g = torch.Generator().manual_seed(8)
        conv = spnn.Conv3d(96, 128, 3).cuda()
        conv.kernel.requires_grad_(False)
        f = torch.randn(coords.shape[0], 96, generator=g).cuda().requires_grad_(True)
        y16 = conv(SparseTensor(f, lidar.C, 1)).F
        gy = torch.randn(y16.shape, generator=g).cuda()
        y16.backward(gy)
        g16 = f.grad.clone()
        ops.set_conv_precision("f32")
        f.grad = None
        y32 = conv(SparseTensor(f, lidar.C, 1)).F
        y32.backward(gy)
        assert 0 < rel_err(y16.detach(), y32.detach()) < 2e-5 and 0 < rel_err(g16, f.grad) < 2e-5
    finally:
        ops.set_conv_precision("f32")
    with torch.no_grad():
        again = mg({"lidar": lidar})["seg_logit"]
    assert torch.equal(again, base)
