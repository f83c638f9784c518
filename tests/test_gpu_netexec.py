"""The backbone pass as ONE library call (lt_net_forward / lt_net_backward through latte_amd/netprog.py) against the
module-by-module path over the same kernels: same features, same BatchNorm gradients.  The sparse convs are the same
launches (bit-equal); the 1x1x1 shortcut convs and the point-branch Linears run through the conv kernel over an identity
plan instead of hipBLASLt, so the comparison is to fp32 rounding (1e-5 relative on the features), not bitwise.
Parity of both paths against the oracle: test_gpu_spvcnn.py (module path), test_gpu_step.py / test_gpu_configs.py (the
TTA step, which takes the one-call path)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def tta_model(cr=1.0, seed=1):
    from latte_amd.spvcnn import Net3DSeg
    from latte_amd.tta_step import configure_model_norm
    torch.manual_seed(seed)
    m = Net3DSeg(9, True, "SPVCNN", {"cr": cr})
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.weight.data = 0.5 + torch.rand(mod.weight.shape, generator=g)
            mod.bias.data = 0.2 * torch.randn(mod.bias.shape, generator=g)
    m, _ = configure_model_norm(m)
    return m.cuda()


def lidar_of(scans):
    from latte_amd.tensor import SparseTensor
    return SparseTensor(torch.from_numpy(scans["feats"]).cuda(), torch.from_numpy(scans["coords"]).cuda(), 1)


def test_identity_plan_runs_a_dense_product():
    from latte_amd import netprog, ops
    torch.manual_seed(0)
    for rows in (1, 63, 64, 1000, 4097):
        x = torch.randn(rows, 32, device="cuda")
        w = torch.randn(1, 32, 48, device="cuda")
        plan = netprog.identity_plan(rows, x.device)
        out = ops.spconv_plan(x, ops.pack_weights(w, 0), plan, 1, rows, 48)
        want = x.double() @ w[0].double()
        assert rel_err(out, want) < 1e-6, rows


@pytest.mark.parametrize("C", [32, 96, 256])
def test_apply_with_the_devoxelized_addend_formed_on_the_fly_is_bit_equal(C):
    """lt_bn_apply_devox == lt_devoxelize_fwd then lt_bn_apply(relu = 3); missing corners, odd row counts"""
    from latte_amd import ops
    torch.manual_seed(C)
    for N, V in ((1, 5), (777, 100), (20001, 3000)):
        x = torch.randn(N, C, device="cuda")
        vox = torch.randn(V, C, device="cuda")
        idx = torch.randint(-1, V, (N, 8), device="cuda", dtype=torch.int32)
        w = torch.rand(N, 8, device="cuda")
        gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
        mean, var = ops.bn_stats(x)
        want = ops.bn_apply(x, mean, var, gamma, beta, 1e-5, residual=ops.devoxelize_fwd(vox, idx, w), relu=True, post_add=True)
        got = ops.bn_apply_devox(x, mean, var, gamma, beta, 1e-5, vox, idx, w)
        assert torch.equal(got, want), (N, V)


@pytest.mark.parametrize("cr", [0.5, 1.0])
def test_one_call_forward_matches_the_module_path(scans_small, cr):
    m = tta_model(cr)
    net = m.net_3d
    prog = net._program()
    lidar = lidar_of(scans_small)
    assert prog.eligible(lidar.feats)
    with torch.no_grad():
        got = net(lidar)                       # keep = 0: buffers reused, BatchNorm applied in place
        want = net._forward_modules(lidar)
    assert got.shape == want.shape
    assert rel_err(got, want) < 2e-5, rel_err(got, want)
    # a second call over the same batch (teacher then student in the step) gives the same bits
    with torch.no_grad():
        again = net(lidar)
    assert torch.equal(got, again)


class _LinBN(torch.autograd.Function):
    """add_to + relu(bn(f @ W^T)) over the conv kernel on an identity plan: the launches lt_net_forward / _backward issue for
    a point branch, as an autograd node (reference for the one-call pass with the dense layers' numerics held equal)"""

    @staticmethod
    def forward(ctx, f, add_to, gamma, beta, weight, eps):
        from latte_amd import netprog, ops
        N, cout = f.shape[0], weight.shape[0]
        w = weight.detach().t().contiguous().unsqueeze(0)
        plan = netprog.identity_plan(N, f.device)
        conv_out, y, mean, var = ops.conv_bn_fwd(f.contiguous(), ops.pack_weights(w, 0), plan, 1, N, cout, gamma.detach(),
                                                 beta.detach(), eps, add_to.contiguous(), 3)
        ctx.save_for_backward(conv_out, mean, var, gamma, beta, w)
        ctx.plan, ctx.eps, ctx.cin = plan, eps, f.shape[1]
        return y

    @staticmethod
    def backward(ctx, dy):
        from latte_amd import ops
        conv_out, mean, var, gamma, beta, w = ctx.saved_tensors
        din, dgamma, dbeta, _ = ops.conv_bn_bwd(dy.contiguous(), conv_out, None, mean, var, gamma.detach(), ctx.eps, True, False,
                                                ops.pack_weights(w, 1), ctx.plan, 1, conv_out.shape[0], ctx.cin,
                                                beta=beta.detach())
        return din, dy, dgamma, dbeta, None, None


def _dense_layers_through_the_conv_kernel(monkeypatch):
    """the module path with its 1x1x1 convs and point-branch Linears issued as the SAME launches the one-call pass uses
    (conv kernel over an identity plan, statistics in its epilogue) instead of hipBLASLt + a statistics pass"""
    from latte_amd import functional as LF
    from latte_amd import netprog, spvcnn
    from latte_amd.tensor import SparseTensor
    orig = spvcnn._conv_bn_impl

    def conv_bn_impl(seq, x, i, relu, residual=None):
        conv, bn = seq[i], seq[i + 1]
        if conv.kernel_volume > 1:
            return orig(seq, x, i, relu, residual)
        V = x.feats.shape[0]
        km = LF.KernelMap(torch.arange(V, dtype=torch.int32, device=x.feats.device).view(1, V), V, V, submanifold=True)
        km._plan = netprog.identity_plan(V, x.feats.device)
        feats = LF.conv_bn(x.feats, conv.kernel, bn.weight, bn.bias, km, False, residual=residual, relu=relu, eps=bn.eps,
                           wcache={})
        y = SparseTensor(feats, x.coords, x.stride)
        y.cmaps, y.kmaps = x.cmaps, x.kmaps
        return y

    def pt(self, i, f, add_to):
        seq = self.point_transforms[i]
        return _LinBN.apply(f, add_to, seq[1].weight, seq[1].bias, seq[0].weight, seq[1].eps)

    monkeypatch.setattr(spvcnn, "_conv_bn_impl", conv_bn_impl)
    monkeypatch.setattr(spvcnn.SPVCNN, "_pt", pt)


def test_one_call_pass_equals_the_same_launches_issued_module_by_module(scans_small, monkeypatch):
    """forward bit for bit, BatchNorm gradients to the rounding of the gradient sums (1e-5): every op kind of the list
    (voxelize, devoxelize, stem, 3^3 / strided / transposed / dense conv+BN, residual, post-add, concat) and the
    gradient accumulation over buffers with several readers"""
    _dense_layers_through_the_conv_kernel(monkeypatch)
    m = tta_model(1.0)
    net = m.net_3d
    lidar = lidar_of(scans_small)
    torch.manual_seed(3)
    out = net(lidar)
    wgt = torch.randn_like(out)
    ref = net._forward_modules(lidar)
    assert torch.equal(out.detach(), ref.detach())
    (out * wgt).sum().backward()
    got = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    for p in net.parameters():
        p.grad = None
    (ref * wgt).sum().backward()
    want = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    assert set(got) == set(want) and len(got) == 2 * 52
    for n in want:
        assert rel_err(got[n], want[n]) < 1e-5, (n, rel_err(got[n], want[n]))


def test_one_call_backward_matches_the_module_path(scans_small):
    """against the shipped module path (dense layers in hipBLASLt): features differ by fp32 rounding, which flips a few
    ReLUs near zero -- the free-running bound of test_gpu_spvcnn.py (1e-2 median, 3e-2 worst)"""
    m = tta_model(1.0)
    net = m.net_3d
    lidar = lidar_of(scans_small)
    torch.manual_seed(3)
    out = net(lidar)                           # keep = 1 + autograd node
    wgt = torch.randn_like(out)
    with torch.no_grad():
        nog = net(lidar)
    assert torch.equal(out.detach(), nog), "keep = 1 and keep = 0 passes must agree bit for bit"
    (out * wgt).sum().backward()
    got = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    for p in net.parameters():
        p.grad = None
    ref = net._forward_modules(lidar)
    assert rel_err(out.detach(), ref.detach()) < 2e-5
    (ref * wgt).sum().backward()
    want = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    assert set(got) == set(want) and len(got) == 2 * 52
    errs = sorted(rel_err(got[n], want[n]) for n in want)
    assert errs[-1] < 3e-2 and errs[len(errs) // 2] < 1e-2, (errs[-1], errs[len(errs) // 2])


def test_batch_prepared_for_inference_gets_the_backward_structures_on_demand(scans_small):
    """a no-grad pass does not build the devoxelize cell tables; a differentiated pass over the SAME batch adds them and gives
    the gradients of a batch that had them from the start"""
    m = tta_model(0.5)
    net = m.net_3d
    lidar = lidar_of(scans_small)
    with torch.no_grad():
        a = net(lidar)
    cache, z = net._batch_cache(lidar)
    assert not cache["net_maps"].struct.v2p_cells[0] and "v2p_cells" not in z.additional_features
    torch.manual_seed(5)
    wgt = torch.randn_like(a)
    out = net(lidar)
    assert cache["net_maps"].struct.v2p_cells[0]
    assert torch.equal(out.detach(), a)
    (out * wgt).sum().backward()
    got = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    for p in net.parameters():
        p.grad = None
    fresh = lidar_of(scans_small)
    (net(fresh) * wgt).sum().backward()
    assert len(got) == 2 * 52
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert torch.equal(got[n], p.grad), n


def test_program_is_not_copied_with_the_model(scans_small):
    m = tta_model(0.5)
    lidar = lidar_of(scans_small)
    with torch.no_grad():
        a = m.net_3d(lidar)
    ema = copy.deepcopy(m)                     # ema_util.create_ema_model
    assert ema.net_3d.__dict__.get("_lt_prog") is None
    with torch.no_grad():
        b = ema.net_3d(lidar_of(scans_small))
    assert torch.equal(a, b)
    # the copy's program points at the COPY's parameters
    with torch.no_grad():
        for p in ema.net_3d.parameters():
            if p.dim() == 1:
                p.mul_(1.5)
        c = ema.net_3d(lidar_of(scans_small))
        d = m.net_3d(lidar)
    assert not torch.equal(b, c) and torch.equal(a, d)


def test_other_configurations_take_the_module_path(scans_small):
    from latte_amd.spvcnn import Net3DSeg
    torch.manual_seed(0)
    m = Net3DSeg(9, True, "SPVCNN", {"cr": 0.5}).cuda()      # plain train mode: running statistics are tracked
    lidar = lidar_of(scans_small)
    assert not m.net_3d._program().eligible(lidar.feats)
    m2 = tta_model(0.5)
    for p in m2.net_3d.stem[0].parameters():
        p.requires_grad_(True)                                # TRAIN.UPDATE_PARAMS == "all": kernels train
    assert not m2.net_3d._program().eligible(lidar.feats)


def test_net_entry_points_refuse_bad_programs(scans_small):
    import ctypes
    from latte_amd import _lib, netprog
    m = tta_model(0.5)
    net = m.net_3d
    lidar = lidar_of(scans_small)
    with torch.no_grad():
        net(lidar)
    prog = net._program()
    maps = lidar._lt_cache["net_maps"]
    lib = _lib.load()
    # an arena that is too small
    prog._ext[prog.in_buf] = lidar.feats.data_ptr()
    small = torch.empty(1024, device="cuda")
    rc = lib.lt_net_forward(ctypes.byref(prog.prog), ctypes.byref(maps.struct), prog._ext, small.data_ptr(), 1024, 0, None, None)
    assert rc != 0 and b"arena" in lib.lt_last_error()
    # an op that reads a buffer nobody wrote
    ops_arr = (netprog.NetOp * prog.n_ops)()
    ctypes.memmove(ops_arr, prog._ops, ctypes.sizeof(ops_arr))
    ops_arr[3].in0 = prog.out_buf
    bad = netprog.NetProg(ops_arr, prog.n_ops, prog._bufs, prog.n_bufs, prog._layers, prog.n_layers, prog.out_buf)
    assert lib.lt_net_arena_floats(ctypes.byref(bad), ctypes.byref(maps.struct), prog._ext, 0) < 0
