"""GPU parity of the individual C-ABI ops against the CPU oracle (oracle/ts_ref.py).
Integer / index results must be bit-exact; fp32 results within 1e-3 relative (north_star tolerance)."""
import numpy as np
import pytest
import torch

from oracle import ts_ref

pytestmark = pytest.mark.gpu
RTOL = 1e-3  # north_star: features/logits within 1e-3 relative, fp32


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def rel_err(a, b):
    a = a.double()
    b = b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_sphash_matches_oracle(scans_small):
    from latte_amd import ops
    c = scans_small["coords"]
    h = ops.sphash(dev(c)).cpu().numpy()
    assert np.array_equal(h, ts_ref.sphash(c))
    off = ts_ref.get_kernel_offsets(3, 1)
    hk = ops.sphash(dev(c), dev(off)).cpu().numpy()
    assert np.array_equal(hk, ts_ref.sphash(c, off))


def test_sphashquery_and_count(scans_small):
    from latte_amd import ops
    c = scans_small["coords"]
    refs = ts_ref.sphash(c)
    off = ts_ref.get_kernel_offsets(2, 1)
    q = ts_ref.sphash(c, off)
    got = ops.sphashquery(dev(q), dev(refs)).cpu().numpy()
    assert np.array_equal(got, ts_ref.sphashquery(q, refs))
    idx = got[0].astype(np.int32)
    cnt = ops.spcount(dev(idx), c.shape[0]).cpu().numpy()
    assert np.array_equal(cnt, ts_ref.spcount(idx, c.shape[0]))


@pytest.mark.parametrize("n,hi", [(1, 5), (1000, 50), (50000, 3000), (200000, 1 << 40)])
def test_unique_sorted_matches_numpy(n, hi):
    from latte_amd import ops
    rng = np.random.default_rng(n)
    keys = rng.integers(0, hi, size=n, dtype=np.int64)
    u, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    gu, gf, gi = ops.unique_sorted(dev(keys))
    assert np.array_equal(gu.cpu().numpy(), u)
    assert np.array_equal(gf.cpu().numpy(), first)
    assert np.array_equal(gi.cpu().numpy(), inv.reshape(-1))


def test_downsample_matches_oracle(scans_small):
    from latte_amd import ops
    c = scans_small["coords"]
    for ts in (1, 2, 4):
        base = ts_ref.spdownsample(c, ts, 1) if ts > 1 else c
        want = ts_ref.spdownsample(base, 2, ts)
        got = ops.downsample_coords(dev(base), 2 * ts).cpu().numpy()
        assert np.array_equal(got, want)


@pytest.mark.parametrize("ks,stride", [(3, 1), (2, 2)])
def test_kernel_map_bit_exact(scans_small, ks, stride):
    from latte_amd import ops
    c = scans_small["coords"]
    nbmaps, nbsizes, out_coords, res = ts_ref.build_kmap(c, 1, ks, stride)
    off = ts_ref.get_kernel_offsets(ks, 1)
    got = ops.kmap_build(dev(c), dev(out_coords), dev(off))
    assert np.array_equal(got.cpu().numpy(), res.astype(np.int32))
    gm, gs = ops.kmap_pairs(got)
    assert np.array_equal(gs.cpu().numpy(), nbsizes)
    assert np.array_equal(gm.cpu().numpy(), nbmaps)
    # reverse table: rev[k, i] = o  <=>  res[k, o] = i
    rev = ops.kmap_reverse(got, c.shape[0]).cpu().numpy()
    kk, oo = np.nonzero(res != -1)
    assert np.array_equal(rev[kk, res[kk, oo]], oo)
    assert (rev != -1).sum() == kk.size


@pytest.mark.parametrize("ks,ts", [(3, 1), (3, 4), ((1, 3, 3), 1), (5, 1), (1, 1)])
def test_submanifold_kernel_map_from_half_the_probes(scans_small, ks, ts):
    """lt_kmap_build_sub (K/2 + 1 probes per row, the other offsets mirrored) == lt_kmap_build == the oracle's table"""
    from latte_amd import functional as LF
    from latte_amd import ops
    # above 32 768 rows the mirrored kernel runs, below it the plain query: 50 000 unique cells of a 60^3 x 2 block (~3
    # neighbours per row) and the small scans
    rng = np.random.default_rng(5)
    cells = rng.choice(60 * 60 * 60 * 2, size=50000, replace=False)
    big = np.stack([cells % 60, cells // 60 % 60, cells // 3600 % 60, cells // 216000], 1).astype(np.int32)
    big[:, :3] *= ts
    for c in (big, scans_small["coords"] if ts == 1 else ts_ref.spdownsample(ts_ref.spdownsample(scans_small["coords"], 2, 1), 2, 2)):
        _check_sub_map(c, ks, ts)


def _check_sub_map(c, ks, ts):
    from latte_amd import functional as LF
    from latte_amd import ops
    cd = dev(c)
    off = LF.get_kernel_offsets(ks, ts, 1, device=cd.device)
    assert torch.equal(off, -off.flip(0))
    full = ops.kmap_build(cd, cd, off)
    half = ops.kmap_build(cd, cd, off, submanifold=True)
    assert torch.equal(full, half)
    if ks == 3:
        _, _, _, res = ts_ref.build_kmap(c, ts, 3, 1)
        assert np.array_equal(half.cpu().numpy(), res.astype(np.int32))


@pytest.mark.parametrize("cin,cout", [(4, 32), (32, 32), (32, 64), (96, 96), (128, 96), (192, 128), (256, 256), (48, 48)])
def test_spconv_forward_matches_oracle(scans_small, cin, cout):
    from latte_amd import ops
    c = scans_small["coords"]
    nbmaps, nbsizes, out_coords, res = ts_ref.build_kmap(c, 1, 3, 1)
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    feats = torch.randn(c.shape[0], cin, generator=g)
    w = torch.randn(27, cin, cout, generator=g) / np.sqrt(27 * cin)
    want = ts_ref.conv3d(feats, w, nbmaps, nbsizes, c.shape[0])
    got = ops.spconv(feats.cuda(), ops.pack_weights(w.cuda(), 0), dev(res, torch.int32), cout).cpu()
    assert rel_err(got, want) < 1e-5
    assert torch.allclose(got, want, rtol=RTOL, atol=1e-4)


def test_spconv_strided_transposed_and_backward_tables(scans_small):
    from latte_amd import ops
    c = scans_small["coords"]
    nbmaps, nbsizes, out_coords, res = ts_ref.build_kmap(c, 1, 2, 2)
    g = torch.Generator().manual_seed(7)
    cin, cout = 32, 64
    feats = torch.randn(c.shape[0], cin, generator=g)
    w = torch.randn(8, cin, cout, generator=g) / np.sqrt(8 * cin)
    want = ts_ref.conv3d(feats, w, nbmaps, nbsizes, out_coords.shape[0])
    resd = dev(res, torch.int32)
    got = ops.spconv(feats.cuda(), ops.pack_weights(w.cuda(), 0), resd, cout).cpu()
    assert rel_err(got, want) < 1e-5
    # transposed conv (coarse -> fine) through the reverse table
    wt = torch.randn(8, cout, cin, generator=g) / np.sqrt(8 * cout)
    coarse = torch.randn(out_coords.shape[0], cout, generator=g)
    want_t = ts_ref.conv3d(coarse, wt, nbmaps, nbsizes, c.shape[0], transposed=True)
    rev = ops.kmap_reverse(resd, c.shape[0])
    got_t = ops.spconv(coarse.cuda(), ops.pack_weights(wt.cuda(), 0), rev, cin).cpu()
    assert rel_err(got_t, want_t) < 1e-5
    # backward-input of the strided conv == transposed conv with Wk^T (mode 1)
    gout = torch.randn(out_coords.shape[0], cout, generator=g)
    f = feats.clone().requires_grad_(True)
    ts_ref.conv3d(f, w, nbmaps, nbsizes, out_coords.shape[0]).backward(gout)
    gin = ops.spconv(gout.cuda(), ops.pack_weights(w.cuda(), 1), rev, cin).cpu()
    assert rel_err(gin, f.grad) < 1e-5


def test_spconv_submanifold_backward_mode2(scans_small):
    from latte_amd import ops
    c = scans_small["coords"]
    nbmaps, nbsizes, _, res = ts_ref.build_kmap(c, 1, 3, 1)
    g = torch.Generator().manual_seed(11)
    cin, cout = 32, 96
    feats = torch.randn(c.shape[0], cin, generator=g, requires_grad=True)
    w = torch.randn(27, cin, cout, generator=g) / np.sqrt(27 * cin)
    gout = torch.randn(c.shape[0], cout, generator=g)
    ts_ref.conv3d(feats, w, nbmaps, nbsizes, c.shape[0]).backward(gout)
    gin = ops.spconv(gout.cuda(), ops.pack_weights(w.cuda(), 2), dev(res, torch.int32), cin).cpu()
    assert rel_err(gin, feats.grad) < 1e-5


def test_spconv_order_permutation_is_result_invariant(scans_small):
    from latte_amd import ops
    c = scans_small["coords"]
    _, _, _, res = ts_ref.build_kmap(c, 1, 3, 1)
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(c.shape[0], 32, generator=g).cuda()
    pw = ops.pack_weights((torch.randn(27, 32, 32, generator=g) / 30).cuda(), 0)
    resd = dev(res, torch.int32)
    a = ops.spconv(feats, pw, resd, 32)
    perm = torch.randperm(c.shape[0], generator=g).to(torch.int32).cuda()
    b = ops.spconv(feats, pw, resd, 32, order=perm)
    assert torch.equal(a, b)


@pytest.mark.parametrize("ks,stride", [(3, 1), (2, 2)])
def test_kmap_order_is_stable_mask_sort(scans_small, ks, stride):
    """lt_kmap_order = stable sort of the rows by a key of their neighbour mask: bits permuted so that the most common
    offset is bit 0 (ties: lower offset first), then the rank of that word in the reflected Gray code."""
    from latte_amd import ops
    c = scans_small["coords"]
    _, _, out_coords, res = ts_ref.build_kmap(c, 1, ks, stride)
    resd = dev(res, torch.int32)
    for table in (resd, ops.kmap_reverse(resd, c.shape[0])):
        t = table.cpu().numpy()
        K = t.shape[0]
        act = t >= 0
        counts = act.sum(1)
        pos = np.empty(K, dtype=np.int64)
        pos[np.lexsort((np.arange(K), -counts))] = np.arange(K)   # descending count, ties by offset
        key = (act.astype(np.int64) << pos[:, None]).sum(0)
        for sh in (1, 2, 4, 8, 16):
            key ^= key >> sh
        want = np.argsort(key, kind="stable")
        got = ops.kmap_order(table).cpu().numpy()
        assert np.array_equal(got, want)
        # what the order is for: useful fraction of the issued 16-row strip x offset MFMA work, against the plain mask order
        def useful(order):
            a = act[:, order]
            pad = (-a.shape[1]) % 16
            a = np.concatenate([a, np.zeros((K, pad), dtype=bool)], 1).reshape(K, -1, 16)
            return act.sum() / (a.any(2).sum() * 16)
        mask = (act.astype(np.int64) << np.arange(K)[:, None]).sum(0)
        assert useful(got) >= useful(np.argsort(mask, kind="stable")) - 0.01
    # the order never changes a result bit
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(c.shape[0], 32, generator=g).cuda()
    pw = ops.pack_weights((torch.randn(res.shape[0], 32, 32, generator=g) / 30).cuda(), 0)
    assert torch.equal(ops.spconv(feats, pw, resd, 32), ops.spconv(feats, pw, resd, 32, order=ops.kmap_order(resd)))


def _sk_variants(cout):
    """split-k variant codes 1000 + 100*SR + 10*NTW + W valid for this width"""
    return [1000 + 100 * sr + 10 * ntw + w for sr in (1, 2, 4) for ntw in (1, 2, 3, 4, 6, 8) for w in (2, 4, 8)
            if (cout // 16) % ntw == 0]


def _sj_variants(cin, cout):
    """split-j variant codes 2000 + 100*SR + 10*NTW + W valid for these widths (W divides Cin/16)"""
    return [2000 + 100 * sr + 10 * ntw + w for sr in (1, 2) for ntw in (2, 3, 4) for w in (2, 3, 4, 6, 8)
            if (cout // 16) % ntw == 0 and (cin // 16) % w == 0]


def _check_sj(ops, a, feats, pw, plan, K, vout, cin, cout):
    """split-j kernels: within fp32 rounding of the single chain `a`, bit for bit from run to run, and bit for bit equal
    to the flat split-k kernel of the same shape (same slab -> wave assignment and summation order when W | Cin/16)"""
    for v in _sj_variants(cin, cout):
        b = ops.spconv_plan(feats, pw, plan, K, vout, cout, variant=v)
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()), v
        assert torch.equal(b, ops.spconv_plan(feats, pw, plan, K, vout, cout, variant=v)), v
        if v % 10 in (2, 4, 8):
            assert torch.equal(b, ops.spconv_plan(feats, pw, plan, K, vout, cout, variant=v - 1000)), v


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 32), (32, 64), (96, 96), (128, 96), (192, 128), (384, 256), (48, 48),
                                      (64, 80)])
@pytest.mark.parametrize("ks,stride", [(3, 1), (2, 2)])
def test_spconv_plan_is_bitwise_equal_to_table_kernel(scans_small, cin, cout, ks, stride):
    """the planned kernels against the plain table kernel, on the forward table and on the reverse table (transposed conv
    / strided backward): single-chain variants (LDS-DMA staged, register-fed) bit for bit; split-k variants (and the
    default, which is one) to fp32 rounding of the re-associated sum, and bit for bit from run to run"""
    from latte_amd import ops
    c = scans_small["coords"]
    _, _, out_coords, res = ts_ref.build_kmap(c, 1, ks, stride)
    K = res.shape[0]
    g = torch.Generator().manual_seed(cin + 7 * cout)
    w = (torch.randn(K, cin, cout, generator=g) / np.sqrt(K * cin)).cuda()
    resd = dev(res, torch.int32)
    feats = torch.randn(c.shape[0], cin, generator=g).cuda()
    pw = ops.pack_weights(w, 0)
    a = ops.spconv(feats, pw, resd, cout)
    plan = ops.kmap_plan(resd)
    # 5000 = the LDS-staged-weights kernel (spconv_ls.hip; default for the 2^3 maps): one chain per output, too
    variants = [1, 5000] + [100 * sr + ntw for sr in (1, 2, 4, 8) for ntw in (1, 2, 4) if (cout // 16) % ntw == 0]
    for v in variants:
        b = ops.spconv_plan(feats, pw, plan, K, resd.shape[1], cout, variant=v)
        assert torch.equal(a, b), v
    for v in [0] + _sk_variants(cout):
        b = ops.spconv_plan(feats, pw, plan, K, resd.shape[1], cout, variant=v)
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()), v
        assert torch.equal(b, ops.spconv_plan(feats, pw, plan, K, resd.shape[1], cout, variant=v)), v
    _check_sj(ops, a, feats, pw, plan, K, resd.shape[1], cin, cout)
    rev = ops.kmap_reverse(resd, c.shape[0])
    gout = torch.randn(out_coords.shape[0], cout, generator=g).cuda()
    pwt = ops.pack_weights(w, 1)
    a = ops.spconv(gout, pwt, rev, cin)
    rplan = ops.kmap_plan(rev)
    for v in [1, 5000] + [100 * sr + ntw for sr in (1, 2, 4, 8) for ntw in (1, 2, 4) if (cin // 16) % ntw == 0]:
        b = ops.spconv_plan(gout, pwt, rplan, K, c.shape[0], cin, variant=v)
        assert torch.equal(a, b), v
    for v in [0] + _sk_variants(cin):
        b = ops.spconv_plan(gout, pwt, rplan, K, c.shape[0], cin, variant=v)
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()), v
    _check_sj(ops, a, gout, pwt, rplan, K, c.shape[0], cout, cin)


def test_spconv_plan_empty_rows_and_ragged_tail():
    """V not a multiple of 64, rows without any neighbour, offsets without any pair"""
    from latte_amd import ops
    g = torch.Generator().manual_seed(21)
    V, K, C = 131, 27, 32
    nbr = torch.full((K, V), -1, dtype=torch.int32)
    nbr[13] = torch.arange(V, dtype=torch.int32)          # centre offset: identity
    nbr[13, 5] = -1                                          # a row with no neighbour at all
    nbr[3, 70:] = torch.randint(0, V, (V - 70,), generator=g).int()
    feats = torch.randn(V, C, generator=g).cuda()
    pw = ops.pack_weights((torch.randn(K, C, C, generator=g) / 10).cuda(), 0)
    nd = nbr.cuda()
    a = ops.spconv(feats, pw, nd, C)
    plan = ops.kmap_plan(nd)
    for v in (1, 101, 102, 201, 202, 401, 402, 801, 802, 5000):
        b = ops.spconv_plan(feats, pw, plan, K, V, C, variant=v)
        assert torch.equal(a, b), v
        assert float(b[5].abs().max()) == 0.0
    for v in [0] + _sk_variants(C) + _sj_variants(C, C):
        b = ops.spconv_plan(feats, pw, plan, K, V, C, variant=v)
        assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()), v
        assert float(b[5].abs().max()) == 0.0


@pytest.mark.parametrize("V,C", [(2, 32), (37, 32), (5000, 96), (20000, 256)])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True)])
def test_batchnorm_fwd_bwd_matches_torch(V, C, relu, res):
    from latte_amd import ops
    g = torch.Generator().manual_seed(V + C)
    x = (torch.randn(V, C, generator=g) * 2 + 3).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g).requires_grad_(True)
    r = torch.randn(V, C, generator=g).requires_grad_(True) if res else None
    eps = 1e-5
    y = torch.nn.functional.batch_norm(x, None, None, gamma, beta, True, 0.0, eps)
    if res:
        y = y + r
    if relu:
        y = torch.relu(y)
    dy = torch.randn(V, C, generator=g)
    y.backward(dy)
    xc = x.detach().cuda()
    mean, var = ops.bn_stats(xc)
    assert torch.allclose(mean.cpu(), x.detach().mean(0), rtol=1e-5, atol=1e-5)
    assert torch.allclose(var.cpu(), x.detach().var(0, unbiased=False), rtol=1e-4, atol=1e-6)
    yg = ops.bn_apply(xc, mean, var, gamma.detach().cuda(), beta.detach().cuda(), eps,
                      residual=None if r is None else r.detach().cuda(), relu=relu)
    assert torch.allclose(yg.cpu(), y.detach(), rtol=RTOL, atol=1e-4)
    if V > 1:
        # ReLU mask from the oracle's y: a y within 1 ulp of 0 may differ in sign between CPU and GPU
        dx, dgamma, dbeta, dres = ops.bn_bwd(dy.cuda(), xc, y.detach().cuda(), mean, var, gamma.detach().cuda(), eps, relu,
                                             need_dx=True, need_dres=res)
        assert rel_err(dx.cpu(), x.grad) < (RTOL if V < 100 else 1e-4)  # V=2: dx ~ 0 by cancellation
        assert rel_err(dgamma.cpu(), gamma.grad) < 1e-4
        assert rel_err(dbeta.cpu(), beta.grad) < 1e-4
        if res:
            assert rel_err(dres.cpu(), r.grad) < 1e-6
        if relu and not res:
            # no residual: the backward can rebuild the ReLU mask from x instead of reading y (relu mode 2 of
            # lt_bn_bwd_reduce / lt_bn_bwd_apply) -- bit for bit the result of the mask taken from the GPU's own y
            a = ops.bn_bwd(dy.cuda(), xc, yg, mean, var, gamma.detach().cuda(), eps, True, need_dx=True)
            b = ops.bn_bwd(dy.cuda(), xc, None, mean, var, gamma.detach().cuda(), eps, True, need_dx=True,
                           beta=beta.detach().cuda())
            for u, v in zip(a[:3], b[:3]):
                assert torch.equal(u, v)


@pytest.mark.parametrize("cin,cout", [(32, 32), (96, 48), (64, 256), (128, 80)])
@pytest.mark.parametrize("ks,stride,transposed", [(3, 1, False), (2, 2, False), (2, 2, True)])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True)])
def test_conv_bn_composite_matches_unfused_ops(scans_small, cin, cout, ks, stride, transposed, relu, res):
    """lt_conv_bn_fwd / lt_conv_bn_bwd (BN statistics from the conv epilogue) against conv -> bn_stats -> bn_apply and
    their backward: the conv output bitwise, statistics / BN output / gradients to fp32 rounding.  Also with bench.py's
    conv launch timer armed."""
    from latte_amd import ops
    from latte_amd import functional as LF
    c = scans_small["coords"]
    _, _, out_coords, results = ts_ref.build_kmap(c, 1, ks, stride)
    nbr = dev(results, torch.int32)
    kmap = LF.KernelMap(nbr, c.shape[0], out_coords.shape[0], submanifold=(stride == 1))
    K = results.shape[0]
    g = torch.Generator().manual_seed(cin * 3 + cout + ks)
    n_in = out_coords.shape[0] if transposed else c.shape[0]
    n_out = c.shape[0] if transposed else out_coords.shape[0]
    w = (torch.randn(K, cin, cout, generator=g) / np.sqrt(cin)).cuda()
    gamma = (torch.rand(cout, generator=g) + 0.5).cuda().requires_grad_(True)
    beta = torch.randn(cout, generator=g).cuda().requires_grad_(True)
    x = torch.randn(n_in, cin, generator=g).cuda().requires_grad_(True)
    r = torch.randn(n_out, cout, generator=g).cuda().requires_grad_(True) if res else None
    dy = torch.randn(n_out, cout, generator=g).cuda()
    eps = 1e-5

    def unfused():
        conv = LF.sparse_conv(x, w, kmap, transposed=transposed)
        y = LF.batch_norm(conv, gamma, beta, eps, residual=r, relu=relu)
        grads = torch.autograd.grad(y, [x, gamma, beta] + ([r] if res else []), dy)
        return conv.detach(), y.detach(), grads

    def fused():
        y = LF.conv_bn(x, w, gamma, beta, kmap, transposed=transposed, residual=r, relu=relu, eps=eps)
        grads = torch.autograd.grad(y, [x, gamma, beta] + ([r] if res else []), dy)
        return y.detach(), grads

    conv_u, y_u, g_u = unfused()
    for trace in (False, True):
        if trace:
            ops.conv_timer_begin(8)
        y_f, g_f = fused()
        if trace:
            ms, meta = ops.conv_timer_end()   # the library's conv launch timer sees the two kernels of the composites
            n_out_rows = y_f.shape[0]
            assert meta.tolist() == [[cin, cout, K, n_out_rows], [cout, cin, K, n_in]] and (ms > 0).all()
        assert torch.allclose(y_f, y_u, rtol=1e-4, atol=1e-4), float((y_f - y_u).abs().max())
        assert rel_err(y_f, y_u) < 1e-5
        for a, b, name in zip(g_f, g_u, ("dx", "dgamma", "dbeta", "dres")):
            assert rel_err(a, b) < 2e-4, (name, rel_err(a, b))
    # the epilogue's per-workgroup partial sums against a float64 reduction of the conv output
    plan = kmap.reverse_plan() if transposed else kmap.plan()
    conv_o, y_o, mean, var = ops.conv_bn_fwd(x.detach(), ops.pack_weights(w, 0), plan, K, n_out, cout, gamma.detach(),
                                             beta.detach(), eps, None, False)
    assert torch.equal(conv_o, conv_u)  # same default kernel with and without the statistics epilogue
    c64 = conv_u.double()
    assert torch.allclose(mean.double(), c64.mean(0), rtol=1e-5, atol=1e-6)
    assert torch.allclose(var.double(), c64.var(0, unbiased=False), rtol=1e-4, atol=1e-7)


def test_voxelize_devoxelize_match_oracle(scans_small):
    from latte_amd import ops
    c = scans_small["coords"]
    N = c.shape[0]
    g = torch.Generator().manual_seed(5)
    # stride-4 cells
    cell = c.copy()
    cell[:, :3] = cell[:, :3] // 4 * 4
    vox = ts_ref.spdownsample(ts_ref.spdownsample(c, 2, 1), 2, 2)
    idx = ts_ref.sphashquery(ts_ref.sphash(cell), ts_ref.sphash(vox))
    assert (idx >= 0).all()
    counts = ts_ref.spcount(idx.astype(np.int32), vox.shape[0])
    feats = torch.randn(N, 32, generator=g, requires_grad=True)
    want = ts_ref.spvoxelize(feats, torch.from_numpy(idx), torch.from_numpy(counts))
    gout = torch.randn(vox.shape[0], 32, generator=g)
    want.backward(gout)
    idx_d, cnt_d = dev(idx, torch.int32), dev(counts)
    got = ops.voxelize_fwd(feats.detach().cuda(), idx_d, cnt_d)
    assert torch.allclose(got.cpu(), want.detach(), rtol=RTOL, atol=1e-5)
    gf = ops.voxelize_bwd(gout.cuda(), idx_d, cnt_d, N)
    assert torch.allclose(gf.cpu(), feats.grad, rtol=RTOL, atol=1e-6)
    # trilinear devoxelize at stride 4
    off = ts_ref.get_kernel_offsets(2, 4)
    q = ts_ref.sphashquery(ts_ref.sphash(cell, off), ts_ref.sphash(vox))  # [8,N]
    pc = torch.from_numpy(c.astype(np.float32))
    w_want = ts_ref.calc_ti_weights(pc, torch.from_numpy(q), 4)
    w_got = ops.calc_ti_weights(pc.cuda(), dev(q), 4)
    assert torch.equal(w_got.cpu(), w_want)
    vf = torch.randn(vox.shape[0], 64, generator=g, requires_grad=True)
    idx8 = torch.from_numpy(q.T.copy())
    w8 = w_want.t().contiguous()
    want_p = ts_ref.spdevoxelize(vf, idx8, w8)
    gp = torch.randn(N, 64, generator=g)
    want_p.backward(gp)
    got_p = ops.devoxelize_fwd(vf.detach().cuda(), idx8.int().cuda(), w8.cuda())
    assert torch.allclose(got_p.cpu(), want_p.detach(), rtol=RTOL, atol=1e-5)
    gvf = ops.devoxelize_bwd(gp.cuda(), idx8.int().cuda(), w8.cuda(), vox.shape[0])
    assert torch.allclose(gvf.cpu(), vf.grad, rtol=RTOL, atol=1e-5)


def test_csr_scatter_kernels_match_atomic_kernels(scans_small):
    """lt_csr_build + lt_voxelize_csr_fwd / lt_devoxelize_csr_bwd (gathers, ascending entry order) against the atomic
    scatter kernels and a float64 index_add; CSR structure checked exactly."""
    from latte_amd import ops
    g = torch.Generator().manual_seed(5)
    N, V, C = 20000, 3000, 32
    idx = torch.randint(-1, V, (N,), generator=g).int()
    idx[:50] = -1
    idx[idx == 7] = 8                                       # an empty row
    perm, off = ops.csr_build(idx.cuda(), V)
    perm_c, off_c = perm.cpu().long(), off.cpu().long()
    valid = (idx >= 0).nonzero().squeeze(1)
    order = valid[torch.sort(idx[valid].long(), stable=True)[1]]
    assert torch.equal(perm_c[: order.numel()], order)
    assert torch.equal(off_c, torch.searchsorted(torch.sort(idx[valid].long())[0], torch.arange(V + 1)))
    feats = torch.randn(N, C, generator=g)
    counts = ops.spcount(idx.cuda(), V)
    a = ops.voxelize_fwd(feats.cuda(), idx.cuda(), counts)
    b = ops.voxelize_csr_fwd(feats.cuda(), (perm, off), V)
    want = torch.zeros(V, C, dtype=torch.float64).index_add_(0, idx[valid].long(), feats[valid].double())
    want = want / counts.cpu().clamp(min=1).double().unsqueeze(1)
    assert rel_err(b.cpu(), want) < 1e-6 and rel_err(a.cpu(), want) < 1e-6
    assert torch.equal(b, ops.voxelize_csr_fwd(feats.cuda(), (perm, off), V))
    # devoxelize backward: entries e = 8*point + corner
    Np = 5000
    idx8 = torch.randint(-1, V, (Np, 8), generator=g).int()
    w8 = torch.rand(Np, 8, generator=g)
    w8[idx8 < 0] = 0.0
    gout = torch.randn(Np, C, generator=g)
    csr8 = ops.csr_build(idx8.cuda().view(-1), V)
    a = ops.devoxelize_bwd(gout.cuda(), idx8.cuda(), w8.cuda(), V)
    b = ops.devoxelize_csr_bwd(gout.cuda(), csr8, w8.cuda(), V)
    want = torch.zeros(V, C, dtype=torch.float64)
    for k in range(8):
        m = idx8[:, k] >= 0
        want.index_add_(0, idx8[m, k].long(), (w8[m, k].double().unsqueeze(1) * gout[m].double()))
    assert rel_err(b.cpu(), want) < 1e-6 and rel_err(a.cpu(), want) < 1e-6


@pytest.mark.parametrize("s,C,drop", [(1, 32, 0.0), (4, 96, 0.0), (2, 32, 0.15), (16, 256, 0.0)])
def test_devoxelize_backward_over_the_point_to_voxel_csr(s, C, drop):
    """lt_devoxelize_cells + lt_devoxelize_cell_bwd (no CSR of the 8N-entry map: the points of a cell share their corner
    voxels) against the atomic scatter kernel, the CSR kernel and a float64 index_add.  drop > 0 removes voxels from the
    level: corners go missing (-1) and some points' own cell is no voxel (the atomically added remainder)."""
    from latte_amd import ops
    from latte_amd import point_voxel as pv
    from latte_amd.tensor import PointTensor, SparseTensor
    g = torch.Generator().manual_seed(11 * s + C)
    N = 30000
    pc = torch.cat([torch.rand(N, 3, generator=g) * torch.tensor([60.0, 50.0, 12.0]), torch.randint(0, 2, (N, 1), generator=g).float()], 1)
    cell = torch.cat([torch.floor(pc[:, :3] / s).int() * s, pc[:, 3:].int()], 1)
    corners = torch.cat([cell + torch.tensor([dx, dy, dz, 0], dtype=torch.int32) * s for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)], 0)
    vox = torch.unique(corners, dim=0)
    vox = vox[torch.randperm(vox.shape[0], generator=g)]
    if drop:
        vox = vox[: int(vox.shape[0] * (1 - drop))]
    V = vox.shape[0]
    x = SparseTensor(torch.zeros(V, C).cuda(), vox.cuda(), s)
    z = PointTensor(torch.zeros(N, C).cuda(), pc.cuda())
    idx8, w8 = pv.v2p_index(x, z)
    z.idx_query[x.s], z.weights[x.s] = idx8, w8
    p2v, _ = pv.p2v_index(x, z)
    assert bool((p2v < 0).any()) == bool(drop)
    cells, p2v_csr, p2v_i = pv.v2p_cells(x, z)
    # the table itself: cells[v, c] = CSR row range of the one cell whose corner c is v ((0, 0): none)
    cells_c, idx8_c, p2v_c = cells.cpu(), idx8.cpu().long(), p2v.cpu().long()
    off_c = p2v_csr[1].cpu()
    want_cells = torch.zeros((V, 8, 2), dtype=torch.int32)
    ok = p2v_c >= 0
    for c in range(8):
        m = ok & (idx8_c[:, c] >= 0)
        want_cells[idx8_c[m, c], c, 0] = off_c[p2v_c[m]]
        want_cells[idx8_c[m, c], c, 1] = off_c[p2v_c[m] + 1]
    assert torch.equal(cells_c, want_cells)
    gout = torch.randn(N, C, generator=g).cuda()
    got = ops.devoxelize_cell_bwd(gout, cells, p2v_csr, p2v_i, idx8, w8)
    a = ops.devoxelize_bwd(gout, idx8, w8, V)
    b = ops.devoxelize_csr_bwd(gout, ops.csr_build(idx8.reshape(-1), V), w8, V)
    want = torch.zeros(V, C, dtype=torch.float64)
    for c in range(8):
        m = idx8_c[:, c] >= 0
        want.index_add_(0, idx8_c[m, c], w8.cpu()[m, c].double().unsqueeze(1) * gout.cpu()[m].double())
    tol = 3e-6   # sequential fp32 sums of up to ~4000 terms per voxel at stride 16
    assert rel_err(got.cpu(), want) < tol and rel_err(a.cpu(), want) < tol and rel_err(b.cpu(), want) < tol
    if not drop:   # no atomics involved: the same bits every time
        assert torch.equal(got, ops.devoxelize_cell_bwd(gout, cells, p2v_csr, p2v_i, idx8, w8))


@pytest.mark.parametrize("V,C", [(5000, 32), (777, 96), (33, 256)])
def test_batch_norm_post_activation_add(V, C):
    """relu(bn(x)) + r in one kernel (lt_bn_apply relu = 3: the point branch of SPVCNN, z.F + point_transforms(z.F))
    against the two-step form relu(bn(x)) then a torch add: output and every gradient bitwise (same kernels for the BN
    part, one fp32 add either way); the gradient of r is dy itself."""
    from latte_amd import functional as LF
    g = torch.Generator().manual_seed(V + C)
    x = torch.randn(V, C, generator=g).cuda().requires_grad_(True)
    r = torch.randn(V, C, generator=g).cuda().requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).cuda().requires_grad_(True)
    beta = (0.3 * torch.randn(C, generator=g)).cuda().requires_grad_(True)
    dy = torch.randn(V, C, generator=g).cuda()
    y_two = LF.batch_norm(x, gamma, beta, 1e-5, relu=True) + r
    g_two = torch.autograd.grad(y_two, [x, gamma, beta, r], dy)
    y_one = LF.batch_norm(x, gamma, beta, 1e-5, residual=r, relu=True, post_add=True)
    g_one = torch.autograd.grad(y_one, [x, gamma, beta, r], dy)
    assert torch.equal(y_one, y_two)
    for a, b in zip(g_one, g_two):
        assert torch.equal(a, b)
    assert torch.equal(g_one[3], dy)
    # and against plain torch
    xh = (x - x.mean(0)) / torch.sqrt(x.var(0, unbiased=False) + 1e-5)
    want = torch.relu(xh * gamma + beta) + r
    assert torch.allclose(y_one, want, rtol=1e-4, atol=1e-5)


def test_bn_apply_rejects_unknown_relu_mode():
    from latte_amd import _lib, ops
    x = torch.randn(64, 16, device="cuda")
    m, v = ops.bn_stats(x)
    one = torch.ones(16, device="cuda")
    y = torch.empty_like(x)
    with pytest.raises(RuntimeError):
        _lib.call("lt_bn_apply", ops._p(x), ops._p(m), ops._p(v), ops._p(one), ops._p(one), 1e-5, None, 3, ops._p(y), 64, 16,
                  ops._stream())       # 3 needs a residual
    with pytest.raises(RuntimeError):
        _lib.call("lt_bn_apply", ops._p(x), ops._p(m), ops._p(v), ops._p(one), ops._p(one), 1e-5, None, 2, ops._p(y), 64, 16,
                  ops._stream())


def test_gather_rows_backward_matches_torch_indexing():
    """functional.gather_rows: same forward as x[idx]; the backward (index_add_ into zeros) equals torch's
    index_put_(accumulate=True) up to the order of the few addends of a row."""
    from latte_amd.functional import gather_rows
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3000, 16, generator=g).cuda().requires_grad_(True)
    idx = torch.randint(0, 3000, (7000,), generator=g).cuda()
    dy = torch.randn(7000, 16, generator=g).cuda()
    y = gather_rows(x, idx)
    assert torch.equal(y, x[idx])
    (got,) = torch.autograd.grad(y, x, dy)
    (want,) = torch.autograd.grad(x[idx], x, dy)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)
    # unique indices (the matched ST voxels): no accumulation at all, bitwise
    perm = torch.randperm(3000, generator=g)[:1000].cuda()
    (got,) = torch.autograd.grad(gather_rows(x, perm), x, dy[:1000])
    (want,) = torch.autograd.grad(x[perm], x, dy[:1000])
    assert torch.equal(got, want)
    with torch.no_grad():
        assert torch.equal(gather_rows(x, idx), x[idx])


def test_packed_weight_cache_follows_every_way_a_kernel_changes(scans_small):
    """The MFMA-fragment image of a conv kernel is cached per module (functional._packed).  torch versions in-place ops on
    the parameter, but not writes through `.data` (the reference's update_ema_all / create_ema_model and
    Conv3d.reset_parameters write that way: latte/common/utils/ema_util.py:44-96): those need invalidate_packed(), and an
    optimizer that holds conv kernels (TRAIN.UPDATE_PARAMS == "all") declares every image stale when it steps."""
    from copy import deepcopy
    from latte_amd import functional as LF
    from latte_amd import sparse_modules as spnn
    from latte_amd.tensor import SparseTensor
    coords = torch.from_numpy(scans_small["coords"]).cuda()
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(coords.shape[0], 32, generator=g).cuda()

    def run(conv):
        x = SparseTensor(feats, coords, 1)
        with torch.no_grad():
            return conv(x).feats.clone()

    conv = spnn.Conv3d(32, 48, 3).cuda()
    conv.kernel.requires_grad_(False)
    y0 = run(conv)
    assert conv._packed_cache                                   # an image is cached now
    conv.kernel.data.mul_(2.0)                                  # unversioned write ...
    conv.invalidate_packed()                                    # ... declared by the writer
    assert torch.allclose(run(conv), 2 * y0, rtol=1e-6, atol=1e-7)
    with torch.no_grad():
        conv.kernel.mul_(0.5)                                   # versioned in-place op: seen without help
    assert torch.allclose(run(conv), y0, rtol=1e-6, atol=1e-7)
    sd = {k: v.clone() * 3 for k, v in conv.state_dict().items()}
    conv.load_state_dict(sd)                                    # checkpoint load after a first forward
    assert torch.allclose(run(conv), 3 * y0, rtol=1e-6, atol=1e-6)
    conv.reset_parameters()                                     # re-initialisation
    want = LF.sparse_conv(feats, conv.kernel.detach(), spnn.build_kernel_map(SparseTensor(feats, coords, 1), 3, 1, 1)[0])
    assert torch.equal(run(conv), want)
    # teacher/student pair, everything trainable: the teacher's kernel is written through .data after the student's
    # optimizer step (tools/latte.py:317-326); the step itself declares the cached images stale
    student = spnn.Conv3d(32, 48, 3).cuda()
    teacher = deepcopy(student)
    for p in teacher.parameters():
        p.detach_()
    yt0 = run(teacher)
    opt = torch.optim.SGD(student.parameters(), lr=0.5)
    x = SparseTensor(feats, coords, 1)
    student(x).feats.square().mean().backward()
    opt.step()
    ys = run(student)
    assert not torch.allclose(ys, yt0)
    for pt, ps in zip(teacher.parameters(), student.parameters()):
        pt.data[:] = 0.0 * pt[:].data[:] + 1.0 * ps[:].data[:]  # update_ema_all with alpha = 0
    assert torch.equal(run(teacher), ys)
    # module-tree helper
    teacher.kernel.data.mul_(2.0)
    LF.invalidate_packed(torch.nn.Sequential(teacher))
    assert torch.allclose(run(teacher), 2 * ys, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (96, 96), (128, 96), (48, 16), (64, 256), (384, 256)])
@pytest.mark.parametrize("ks,stride,transposed", [(3, 1, False), (2, 2, False), (2, 2, True)])
def test_spconv_bwd_weight_matches_oracle_autograd(scans_small, cin, cout, ks, stride, transposed):
    """lt_spconv_bwd_weight (pairs as the MFMA reduction index, fixed-order slice reduction) against autograd through the
    oracle's conv3d (oracle/ts_ref.py: per-offset gather -> GEMM -> scatter); 1e-4 relative, bit-reproducible."""
    from latte_amd import ops
    c = scans_small["coords"]
    nbmaps, nbsizes, out_coords, res = ts_ref.build_kmap(c, 1, ks, stride)
    n_fine, n_out = c.shape[0], out_coords.shape[0]
    K = ks ** 3
    g = torch.Generator().manual_seed(cin * 7 + cout)
    n_in, n_o = (n_out, n_fine) if transposed else (n_fine, n_out)
    feats = torch.randn(n_in, cin, generator=g)
    gout = torch.randn(n_o, cout, generator=g)
    w = (torch.randn(K, cin, cout, generator=g) / np.sqrt(K * cin)).requires_grad_(True)
    ts_ref.conv3d(feats, w, nbmaps, nbsizes, n_o, transposed=transposed).backward(gout)
    nb_d, ns_d = dev(nbmaps, torch.int32), dev(nbsizes, torch.int32)
    got = ops.spconv_bwd_weight(feats.cuda(), gout.cuda(), nb_d, ns_d, K, cin, cout, transposed)
    assert rel_err(got.cpu(), w.grad) < 1e-4
    assert torch.equal(got, ops.spconv_bwd_weight(feats.cuda(), gout.cuda(), nb_d, ns_d, K, cin, cout, transposed))


def test_conv3d_module_trains_its_kernel_through_the_hip_backward(scans_small):
    """TRAIN.UPDATE_PARAMS == "all": kernel gradients of spnn.Conv3d (3^3, strided 2^3 and its transposed twin, 4-channel
    stem width through the torch fallback) against the oracle, then one SGD step is visible in the next forward."""
    from latte_amd import sparse_modules as spnn
    from latte_amd.tensor import SparseTensor
    c = scans_small["coords"]
    coords = torch.from_numpy(c).cuda()
    g = torch.Generator().manual_seed(21)
    for cin, cout in ((4, 32), (32, 48)):
        feats = torch.randn(c.shape[0], cin, generator=g)
        down = spnn.Conv3d(cin if cin % 16 == 0 else 32, cout, 2, stride=2).cuda()
        conv = spnn.Conv3d(cin, cout, 3).cuda()
        x = SparseTensor(feats.cuda(), coords, 1)
        y = conv(x)
        gy = torch.randn(y.F.shape, generator=g)
        y.F.backward(gy.cuda())
        nbmaps, nbsizes, out_coords, _ = ts_ref.build_kmap(c, 1, 3, 1)
        w = conv.kernel.detach().cpu().requires_grad_(True)
        ts_ref.conv3d(feats, w, nbmaps, nbsizes, c.shape[0]).backward(gy)
        assert rel_err(conv.kernel.grad.cpu(), w.grad) < 1e-4
        before = y.F.detach().clone()
        torch.optim.SGD(conv.parameters(), lr=0.1).step()
        with torch.no_grad():
            after = conv(SparseTensor(feats.cuda(), coords, 1)).F
        assert not torch.allclose(before, after)          # the packed-weight cache followed the optimizer step
        del down


@pytest.mark.parametrize("cin,c,dual", [(96, 10, True), (96, 16, True), (48, 9, True), (96, 5, False), (128, 32, True)])
def test_head_gather_matches_linear_then_gather(cin, c, dual):
    """lt_head_gather_fwd (heads evaluated at the raw points) against nn.Linear on the voxel rows followed by the
    inverse_to_all gather (xmuda_arch.py:326-335, collate.py:34-55): forward 1e-5, gradients to the features and to the
    head parameters 1e-5, out-of-range / ragged tail handled."""
    from latte_amd import functional as LF
    from latte_amd import ops
    g = torch.Generator().manual_seed(cin + c)
    V, N = 5003, 20011                                       # ragged: N % 16 != 0
    feats = torch.randn(V, cin, generator=g)
    idx = torch.randint(0, V, (N,), generator=g)
    idx[:7] = V - 1
    w1, b1 = torch.randn(c, cin, generator=g) / cin ** 0.5, torch.randn(c, generator=g)
    w2, b2 = (torch.randn(c, cin, generator=g) / cin ** 0.5, torch.randn(c, generator=g)) if dual else (None, None)
    gw1, gw2 = torch.randn(N, c, generator=g), torch.randn(N, c, generator=g)

    def leaf(t):
        return None if t is None else t.clone().requires_grad_(True)
    f_r, w1_r, b1_r, w2_r, b2_r = leaf(feats), leaf(w1), leaf(b1), leaf(w2), leaf(b2)
    want1 = torch.nn.functional.linear(f_r, w1_r, b1_r)[idx]
    loss = (want1 * gw1).sum()
    if dual:
        want2 = torch.nn.functional.linear(f_r, w2_r, b2_r)[idx]
        loss = loss + (want2 * gw2).sum()
    loss.backward()

    dev_ = lambda t: None if t is None else t.cuda().requires_grad_(True)
    f_g, w1_g, b1_g, w2_g, b2_g = dev_(feats), dev_(w1), dev_(b1), dev_(w2), dev_(b2)
    width = ((2 * c if dual else c) + 15) // 16 * 16
    E = torch.zeros(cin, width)
    bias = torch.zeros(width)
    E[:, :c], bias[:c] = w1.t(), b1
    if dual:
        E[:, c:2 * c], bias[c:2 * c] = w2.t(), b2
    packed = ops.pack_weights(E.unsqueeze(0).cuda(), 0)
    idx_g = idx.cuda()
    csr = ops.csr_build(idx_g.int(), V)
    o1, o2 = LF.head_gather(f_g, w1_g, b1_g, w2_g, b2_g, idx_g, csr, packed, bias.cuda())
    assert rel_err(o1.detach().cpu(), want1.detach()) < 1e-5
    l = (o1 * gw1.cuda()).sum()
    if dual:
        assert rel_err(o2.detach().cpu(), want2.detach()) < 1e-5
        l = l + (o2 * gw2.cuda()).sum()
    else:
        assert o2 is None
    l.backward()
    for got, want in ((f_g, f_r), (w1_g, w1_r), (b1_g, b1_r)) + (((w2_g, w2_r), (b2_g, b2_r)) if dual else ()):
        assert rel_err(got.grad.cpu(), want.grad) < 1e-5
    # identity index (idx = None) and a head the loss ignores
    o1n, _ = ops.head_gather_fwd(feats.cuda(), None, packed, bias.cuda(), c, c if dual else 0)
    assert rel_err(o1n.cpu(), torch.nn.functional.linear(feats, w1, b1)) < 1e-5
    if dual:
        f2 = feats.cuda().requires_grad_(True)
        a1, _a2 = LF.head_gather(f2, w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda(), idx_g, None, packed, bias.cuda())
        (a1 * gw1.cuda()).sum().backward()
        f3 = feats.clone().requires_grad_(True)
        (torch.nn.functional.linear(f3, w1, b1)[idx] * gw1).sum().backward()
        assert rel_err(f2.grad.cpu(), f3.grad) < 1e-5


@pytest.mark.parametrize("cout", [32, 16])
def test_stem_kernel_matches_oracle_and_scalar_kernel(scans_small, cout):
    """lt_spconv_stem_fwd (4 input channels, one lane per output row) against the oracle conv and the MFMA table kernel."""
    from latte_amd import ops
    from latte_amd import functional as LF
    c = scans_small["coords"]
    nbmaps, nbsizes, out_coords, res = ts_ref.build_kmap(c, 1, 3, 1)
    g = torch.Generator().manual_seed(cout)
    feats = torch.randn(c.shape[0], 4, generator=g)
    w = torch.randn(27, 4, cout, generator=g) / np.sqrt(27 * 4)
    want = ts_ref.conv3d(feats, w, nbmaps, nbsizes, c.shape[0])
    resd = dev(res, torch.int32)
    plan = ops.kmap_plan(resd)
    got = ops.spconv_stem(feats.cuda(), w.cuda(), plan, 27, c.shape[0])
    assert rel_err(got.cpu(), want) < 1e-5
    old = ops.spconv(feats.cuda(), ops.pack_weights(w.cuda(), 0), resd, cout)
    assert rel_err(got.cpu(), old.cpu()) < 1e-6
    assert torch.equal(got, ops.spconv_stem(feats.cuda(), w.cuda(), plan, 27, c.shape[0]))
    # ragged tail / rows without any neighbour but themselves
    V = 130
    tbl = torch.full((27, V), -1, dtype=torch.int32)
    tbl[13] = torch.arange(V, dtype=torch.int32)
    tbl[0, 5], tbl[26, 129] = 7, 0
    f = torch.randn(V, 4, generator=g)
    o = ops.spconv_stem(f.cuda(), w.cuda(), ops.kmap_plan(tbl.cuda()), 27, V).cpu()
    ref = f @ w[13]
    ref[5] += f[7] @ w[0]
    ref[129] += f[0] @ w[26]
    assert torch.allclose(o, ref, rtol=1e-5, atol=1e-6)
    # and through the module path (what SPVCNN's stem runs)
    km = LF.KernelMap(resd, c.shape[0], c.shape[0], submanifold=True)
    y = LF.sparse_conv(feats.cuda(), w.cuda(), km)
    assert torch.equal(y, got)


def test_conv_epilogue_statistics_with_a_large_mean(scans_small):
    """The fused conv+BN statistics are E[x^2] - E[x]^2 from un-shifted fp32 per-workgroup sums (fp64 combination): the
    relative error of var grows like 1e-7 * (1 + mean^2/var).  Known-answer check at |mean|/std ~ 30 (conv outputs of the
    network sit at O(1)): var within 1e-7 * (1 + ratio^2) * 8 of the shifted two-pass lt_bn_stats, mean within 1e-6."""
    from latte_amd import ops
    from latte_amd import functional as LF
    c = scans_small["coords"]
    _, _, _, res = ts_ref.build_kmap(c, 1, 3, 1)
    resd = dev(res, torch.int32)
    plan = ops.kmap_plan(resd)
    V = c.shape[0]
    g = torch.Generator().manual_seed(12)
    feats = torch.randn(V, 32, generator=g).cuda()
    w = torch.zeros(27, 32, 32)
    w[13] = torch.eye(32)                       # centre offset only: conv output = input + constant shift below
    feats = feats + 30.0
    packed = ops.pack_weights(w.cuda(), 0)
    gamma, beta = torch.ones(32, device="cuda"), torch.zeros(32, device="cuda")
    conv_out, y, mean, var = ops.conv_bn_fwd(feats, packed, plan, 27, V, 32, gamma, beta, 1e-5, None, False)
    assert torch.equal(conv_out, feats)
    m2, v2 = ops.bn_stats(conv_out)
    ratio2 = float((m2 * m2 / v2).max())
    assert 500 < ratio2 < 2000
    assert torch.allclose(mean, m2, rtol=1e-6)
    assert float(((var - v2).abs() / v2).max()) < 1e-7 * (1 + ratio2) * 8


def test_bf16x3_pack_and_launch_sides_use_one_rule(scans_small):
    """Under lt_conv_set_precision(1) the pack side (functional._packed -> bf16 hi|lo image) and the launch side decide with
    ONE predicate (lt_conv_uses_bf16).   Cout = 80 / 112 (5 / 7 column tiles: no bf16 shape) stay fp32 on both
    sides and give the fp32 result bit for bit; a conv the rule claims but the kernel cannot address (operands past 2 GiB),
    or an explicit fp32 variant over a bf16 image, is an error -- never an fp32 kernel reading a bf16 image."""
    from latte_amd import _lib, ops
    from latte_amd import sparse_modules as spnn
    from latte_amd.tensor import SparseTensor
    c = scans_small["coords"]
    g = torch.Generator().manual_seed(3)
    V = c.shape[0]
    try:
        for cin, cout in ((64, 80), (96, 112), (64, 96)):
            conv = spnn.Conv3d(cin, cout, 3).cuda()
            conv.kernel.requires_grad_(False)
            f = torch.randn(V, cin, generator=g).cuda()
            ops.set_conv_precision("f32")
            want = conv(SparseTensor(f, dev(c), 1)).F
            ops.set_conv_precision("bf16x3")
            got = conv(SparseTensor(f, dev(c), 1)).F
            if cout in (80, 112):
                assert not ops.conv_uses_bf16(cin, cout)
                assert torch.equal(got, want), (cin, cout)             # every output column written, by the fp32 kernel
            else:
                assert ops.conv_uses_bf16(cin, cout)
                assert 0 < rel_err(got, want) < 2e-5
        # a claimed conv whose operands the 32-bit buffer offsets cannot reach: refused (LT_ERR_RANGE), output untouched
        ops.set_conv_precision("bf16x3")
        x = SparseTensor(torch.zeros(V, 4, device="cuda"), dev(c), 1)
        kmap, _ = spnn.build_kernel_map(x, 3, 1, 1)
        big = torch.empty((2_200_000, 256), dtype=torch.float32, device="cuda")    # 2.25 GB of rows; never read
        pw = ops.pack_weights(torch.zeros(27, 256, 64, device="cuda"), 4)
        with pytest.raises(_lib.LatteHipError, match="2 GiB"):
            ops.spconv_plan(big, pw, kmap.plan(), 27, V, 64)
        del big
        # an explicit fp32 variant while the weights of this width are bf16 images: refused as well
        f = torch.randn(V, 64, generator=g).cuda()
        pw = ops.pack_weights(torch.zeros(27, 64, 64, device="cuda"), 4)
        with pytest.raises(_lib.LatteHipError, match="fp32 kernel"):
            ops.spconv_plan(f, pw, kmap.plan(), 27, V, 64, variant=2142)
    finally:
        ops.set_conv_precision("f32")
