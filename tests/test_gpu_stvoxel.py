"""GPU parity of the ST-voxel aggregation (HR_Block / CR_Block / refine_ety / scatter) against the golden vectors
captured from the reference's own hr_module.py (tests/golden/hr_*.npz) and against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import hr_ref
from tests.helpers import HR_CASES, hr_roles, load_hr_case

pytestmark = pytest.mark.gpu
RTOL = 1e-3


def to_dev(b):
    out = dict(b)
    out["seg_logit"] = b["seg_logit"].cuda() if not b["seg_logit"].requires_grad else b["seg_logit"]
    return out


@pytest.mark.parametrize("case", HR_CASES)
@pytest.mark.parametrize("tag", ["23", "32"])
def test_hr_cr_block_matches_reference_golden(case, tag):
    from latte_amd.hr_module import CR_Block, HR_Block
    z, cfg, batch = load_hr_case(case)
    ab, at, hb, ht, a_lg = hr_roles(z, batch, tag)
    a_dev = a_lg.cuda().requires_grad_(True)
    ab = dict(ab, seg_logit=a_dev)
    hb = dict(hb, seg_logit=hb["seg_logit"].cuda())
    hr = HR_Block(cfg["temp_wd_size"], cfg["max_range"], cfg["voxel_size"], cfg["conf_perc"])
    d = hr(ab, at, hb, ht)
    assert set(d.keys()) == {"ac_logits", "ac_voxels", "tr_logits", "tr_voxels", "tr_conf_mask", "tr_ety_loss",
                             "tr_pc2img_mask", "tr_img_points", "ac_p2v_indices", "tr_p2v_indices", "hs_pc_ety"}
    for i in range(cfg["B"]):
        for k in ("ac_voxels", "tr_voxels", "ac_p2v_indices", "tr_p2v_indices"):  # bit-exact index structures
            assert np.array_equal(d[k][i].cpu().numpy(), z[f"{tag}_{k}_{i}"]), (k, i)
        for k in ("ac_logits", "tr_logits"):
            assert np.allclose(d[k][i].detach().cpu().numpy(), z[f"{tag}_{k}_{i}"], rtol=RTOL, atol=1e-6), (k, i)
    assert np.allclose(d["hs_pc_ety"].cpu().numpy(), z[f"{tag}_hs_pc_ety"], rtol=RTOL, atol=1e-6)
    got_mask, want_mask = d["tr_conf_mask"].cpu().numpy(), z[f"{tag}_tr_conf_mask"]
    assert got_mask.shape == want_mask.shape
    # an entropy equal to its class threshold to fp32 rounding may fall on either side: at most 2 such entries, and every
    # one of them IS such a tie (entropy within 2 ulp of the class quantile the oracle computes from the same inputs)
    bad = np.nonzero(got_mask != want_mask)[0]
    assert bad.size <= 2
    if bad.size:
        o = hr_ref.hr_forward(dict(ab, seg_logit=a_lg.detach()), at, dict(hb, seg_logit=hb["seg_logit"].cpu()), ht,
                              cfg["temp_wd_size"], cfg["max_range"], cfg["voxel_size"], cfg["conf_perc"])
        lab = torch.cat(o["tr_logits"], 0).argmax(1)
        ety = o["_voxel_ety_all"][:lab.numel()]
        for i in bad:
            assert i < lab.numel()
            thr = float(torch.quantile(ety[lab == lab[i]], cfg["conf_perc"]))
            assert abs(float(ety[i]) - thr) <= 2 * np.spacing(np.float32(thr)), (i, float(ety[i]), thr)
    d["ac_pc_ety"] = torch.from_numpy(z[f"{tag}_ac_pc_ety"]).cuda()
    d["tr_pc_ety"] = torch.from_numpy(z[f"{tag}_tr_pc_ety"]).cuda()
    if (got_mask != want_mask).any():
        d["tr_conf_mask"] = torch.from_numpy(want_mask).cuda()
    loss = CR_Block(reduction="mean", conf_mask=True, ety_weight=True)(d)
    want = float(z[f"{tag}_cr_loss"])
    assert abs(float(loss.detach()) - want) < RTOL * abs(want)
    loss.backward()
    g, gw = a_dev.grad.cpu().numpy(), z[f"{tag}_grad_anchor_logit"]
    assert np.linalg.norm(g - gw) < RTOL * np.linalg.norm(gw)


def test_softmax_entropy_and_refine_known_answers():
    from latte_amd import ops
    from latte_amd.hr_module import refine_ety
    z, cfg, _ = load_hr_case("tiny")
    g = torch.Generator().manual_seed(0)
    logits = 3 * torch.randn(5000, 9, generator=g)
    probs, ety = ops.softmax_entropy(logits.cuda())
    want_p = torch.softmax(logits, 1)
    assert torch.allclose(probs.cpu(), want_p, rtol=1e-5, atol=1e-7)
    assert torch.allclose(ety.cpu(), hr_ref.prob_2_entropy(want_p).sum(1), rtol=1e-4, atol=1e-6)
    m = refine_ety(torch.from_numpy(z["refine_ety_in_ety"]).cuda(), torch.from_numpy(z["refine_ety_in_label"]).cuda(), 0.75)
    assert np.array_equal(m.cpu().numpy(), z["refine_ety_mask"])
    # refine_pseudo_labels: golden vector captured from the reference's refine_pseudo_labels.py:10-30
    from latte_amd.tta_step import refine_pseudo_labels
    r = refine_pseudo_labels(torch.from_numpy(z["rpl_in_prob"]).cuda(), torch.from_numpy(z["rpl_in_label"]).cuda())
    assert np.array_equal(r.cpu().numpy(), z["rpl_out"])
    # larger random case against the oracle, incl. an empty class and even/odd class sizes (lower median)
    g = torch.Generator().manual_seed(4)
    prob = torch.rand(20001, generator=g)
    lab = torch.randint(0, 7, (20001,), generator=g)
    lab[lab == 3] = 5
    want = hr_ref.refine_pseudo_labels(prob, lab)
    got = refine_pseudo_labels(prob.cuda(), lab.cuda(), num_classes=9)
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("reduce", ["mean", "sum", "max"])
def test_scatter_shim_matches_oracle(reduce):
    import latte_amd
    latte_amd.install_shims()
    import torch_scatter.scatter as ts_scatter  # exactly how the reference imports it (hr_module.py:5)
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, 700, (20000,), generator=g)
    idx[idx == 13] = 14  # a hole: row 13 stays zero
    src = torch.randn(20000, 9, generator=g, requires_grad=True)
    n = int(idx.max()) + 1
    if reduce == "max":
        want = torch.full((n, 9), -float("inf")).scatter_reduce(0, idx[:, None].expand(-1, 9), src, "amax")
        want = torch.where(torch.isinf(want), torch.zeros_like(want), want)
    else:
        want = torch.zeros(n, 9).index_add(0, idx, src)
        if reduce == "mean":
            want = want / torch.bincount(idx, minlength=n).clamp(min=1)[:, None]
    gw = torch.randn(n, 9, generator=g)
    want.backward(gw)
    s2 = src.detach().cuda().requires_grad_(True)
    got = ts_scatter(s2, idx.cuda(), dim=0, reduce=reduce)
    assert torch.allclose(got.detach().cpu(), want.detach(), rtol=1e-5, atol=1e-6)
    got.backward(gw.cuda())
    assert torch.allclose(s2.grad.cpu(), src.grad, rtol=1e-5, atol=1e-6)
    # 1-D source (the entropy call site hr_module.py:252)
    e = torch.rand(20000, generator=g)
    got1 = ts_scatter(e.cuda(), idx.cuda(), dim=0, reduce="mean")
    assert torch.allclose(got1.cpu(), hr_ref.scatter_mean(e, idx), rtol=1e-5, atol=1e-7)


def test_hr_block_raises_like_reference_when_nothing_corresponds():
    from latte_amd.hr_module import HR_Block
    z, cfg, batch = load_hr_case("tiny")
    ab, at, hb, ht, a_lg = hr_roles(z, batch, "32")
    ab = dict(ab, seg_logit=a_lg.cuda(), ori_points=[p + 25.0 for p in ab["ori_points"]])  # anchors moved away
    hb = dict(hb, seg_logit=hb["seg_logit"].cuda())
    with pytest.raises(AssertionError):
        HR_Block(cfg["temp_wd_size"], cfg["max_range"], cfg["voxel_size"], cfg["conf_perc"])(ab, at, hb, ht)


@pytest.mark.parametrize("cp", [0.9, None])
@pytest.mark.parametrize("N,C", [(5000, 10), (257, 17), (64, 2)])
def test_pseudo_label_fuse_matches_oracle(N, C, cp):
    """lt_pseudo_label_fuse (two kernels) against the oracle's restatement of latte.py:207-236: entropies and confidences
    to 1e-5, labels identical wherever the two best fused probabilities are further apart than the fp32 noise."""
    from latte_amd import ops
    from oracle.step_ref import pseudo_label_fuse_ref
    g = torch.Generator().manual_seed(N + C)
    mask = torch.rand(N, generator=g) < 0.4
    Nf = int(mask.sum())
    l2 = 3 * torch.randn(Nf, C, generator=g)
    l3 = 3 * torch.randn(N, C, generator=g)
    st2 = torch.rand(Nf, generator=g) * (torch.rand(Nf, generator=g) < 0.7)      # 0 = the point has no ST voxel
    st3 = torch.rand(N, generator=g) * (torch.rand(N, generator=g) < 0.7)
    want_pc2, want_pc3, prob = pseudo_label_fuse_ref(l2, l3, mask, st2.clone(), st3.clone(), cp)
    fov_idx = mask.nonzero().squeeze(1).cuda()
    thr2 = thr3 = None
    if cp is not None:
        thr2, thr3 = torch.quantile(st2.cuda(), cp), torch.quantile(st3.cuda(), cp)
    pc2, pc3, conf, label = ops.pseudo_label_fuse(l2.cuda(), l3.cuda(), fov_idx, st2.cuda(), st3.cuda(), thr2, thr3)
    assert label.dtype == torch.int32
    # substituted entries are copies (bit-exact); computed entropies agree to fp32 rounding
    assert torch.allclose(pc2.cpu(), want_pc2, rtol=1e-5, atol=1e-6)
    assert torch.allclose(pc3.cpu(), want_pc3, rtol=1e-5, atol=1e-6)
    sub3 = (st3 != 0) if cp is None else (st3 != 0) & (st3 <= thr3.cpu())
    assert torch.equal(pc3.cpu()[sub3], st3[sub3])
    assert torch.allclose(conf.cpu(), prob.amax(1), rtol=1e-5, atol=1e-7)
    top2 = prob.topk(2, 1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert clear.float().mean() > 0.99
    assert torch.equal(label.cpu().long()[clear], prob.argmax(1)[clear])


def test_pseudo_label_fuse_edge_cases():
    from latte_amd import ops
    dev = "cuda"
    # no point inside the FOV: 3-D teacher alone
    l3 = torch.randn(100, 6, device=dev)
    e = torch.zeros(0, device=dev)
    pc2, pc3, conf, label = ops.pseudo_label_fuse(torch.zeros(0, 6, device=dev), l3, torch.zeros(0, dtype=torch.int64, device=dev),
                                                  e, torch.zeros(100, device=dev))
    p = torch.softmax(l3, 1)
    assert pc2.numel() == 0 and torch.allclose(conf, p.amax(1), rtol=1e-5) and torch.equal(label.long(), p.argmax(1))
    # empty scan
    out = ops.pseudo_label_fuse(torch.zeros(0, 6, device=dev), torch.zeros(0, 6, device=dev),
                                torch.zeros(0, dtype=torch.int64, device=dev), e, e)
    assert all(o.numel() == 0 for o in out)
    with pytest.raises(RuntimeError):
        ops.pseudo_label_fuse(torch.zeros(0, 65, device=dev), torch.zeros(4, 65, device=dev),
                              torch.zeros(0, dtype=torch.int64, device=dev), e, torch.zeros(4, device=dev))


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 208896])
@pytest.mark.parametrize("q", [0.0, 0.5, 0.9, 1.0])
def test_quantile_matches_torch(n, q):
    """lt_quantile (keys-only radix sort + one interpolation) against torch.quantile: same value (ATen's fp32 rank and
    lerp arithmetic are reproduced), with negative values, zeros and ties in the input."""
    from latte_amd import ops
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g)
    x[torch.rand(n, generator=g) < 0.3] = 0.0                     # many points have no ST voxel: entropy 0
    x = x.cuda()
    got = ops.quantile(x, q)
    want = torch.quantile(x, q)
    assert got.shape == want.shape == torch.Size([])
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-7), (float(got), float(want))


@pytest.mark.parametrize("C", [1, 9, 16, 17])
@pytest.mark.parametrize("q", [0.0, 0.5, 0.75, 1.0])
def test_class_quantile_and_median_masks_match_torch_on_hard_inputs(C, q):
    """the per-class thresholds (radix selection for <= 16 classes, radix sort above) against torch.quantile / torch.median:
    classes with 0, 1, 2 elements, one class of identical values, ties across the bins' boundaries, ignored labels"""
    from latte_amd import ops
    g = torch.Generator().manual_seed(100 * C + int(q * 100))
    n = 40000
    lab = torch.randint(0, C, (n,), generator=g)
    ety = torch.rand(n, generator=g) * 2.5
    ety[torch.rand(n, generator=g) < 0.2] = 0.0                        # no ST voxel: exactly 0
    ety[torch.rand(n, generator=g) < 0.1] = 1.0                        # a tie at a power of two (a bin boundary of every level)
    if C >= 9:
        lab[lab == 3] = 4                                              # class 3: empty
        lab[:1] = 3                                                    # ... no: ONE element
        lab[lab == 5] = 6
        lab[1:3] = 5                                                   # class 5: two elements
        ety[lab == 7] = 0.625                                          # class 7: all values equal
        lab[lab == 8] = 2                                              # class 8: empty
    lab[torch.rand(n, generator=g) < 0.05] = -100                      # ignore label: kept, in no class
    ety_d, lab_d = ety.cuda(), lab.int().cuda()
    got = ops.class_quantile_mask(ety_d, lab_d, C, q).bool().cpu()
    keep = ops.class_median_mask(ety_d, lab_d, C, 0.9).bool().cpu()
    want, want_keep = torch.ones(n, dtype=torch.bool), torch.ones(n, dtype=torch.bool)
    for c in range(C):
        sel = lab == c
        if not sel.any():
            continue
        t = torch.quantile(ety_d[sel.cuda()], q).cpu()                 # on the device, like the product (same lerp)
        want[sel] = ~(ety[sel] > t)
        med = min(float(ety[sel].median()), 0.9)
        want_keep[sel] = ~(ety[sel] < med)
    assert torch.equal(got, want)
    assert torch.equal(keep, want_keep)


def test_quantile_rejects_bad_arguments():
    from latte_amd import ops
    with pytest.raises(RuntimeError):
        ops.quantile(torch.zeros(0, device="cuda"), 0.5)
    with pytest.raises(RuntimeError):
        ops.quantile(torch.zeros(8, device="cuda"), 1.5)


def test_class_masks_ignore_labels_outside_the_class_range():
    """refine_ety / refine_pseudo_labels with an ignore label (-100) or a label >= num_classes among the inputs: those
    entries belong to no class -- never masked, and no out-of-bounds threshold read."""
    from latte_amd import ops
    g = torch.Generator().manual_seed(2)
    n = 4096
    ety = torch.rand(n, generator=g)
    lab = torch.randint(0, 5, (n,), generator=g)
    odd = torch.rand(n, generator=g) < 0.1
    lab_odd = lab.clone()
    lab_odd[odd] = -100
    lab_odd[odd & (torch.arange(n) % 2 == 0)] = 7
    m_ref = ops.class_quantile_mask(ety[~odd].cuda(), lab[~odd].int().cuda(), 5, 0.75).cpu()
    m = ops.class_quantile_mask(ety.cuda(), lab_odd.int().cuda(), 5, 0.75).cpu()
    assert bool(m[odd].all()) and torch.equal(m[~odd], m_ref)
    k_ref = ops.class_median_mask(ety[~odd].cuda(), lab[~odd].int().cuda(), 5, 0.9).cpu()
    k = ops.class_median_mask(ety.cuda(), lab_odd.int().cuda(), 5, 0.9).cpu()
    assert bool(k[odd].all()) and torch.equal(k[~odd], k_ref)


def test_sort_segments_checks_the_key_bits_promise():
    """ lt_sort_segments sorts key_bits + 1 bits only; an unmasked key at or beyond 2^key_bits (a stale key_bits,
    an anchor id beyond its bits) would come out mis-sorted without a trace.  The device checks every key: LT_ERR_RANGE."""
    from latte_amd import _lib, ops
    g = torch.Generator().manual_seed(9)
    keys = torch.randint(0, 1 << 20, (5000,), generator=g)
    keys[::7] = torch.iinfo(torch.int64).max                      # masked elements are fine
    se, inv, uk, seg = ops.sort_segments(keys.cuda(), 20)
    want = torch.unique(keys[keys != torch.iinfo(torch.int64).max])
    assert torch.equal(uk.cpu(), want)
    bad = keys.clone()
    bad[123] = (1 << 20) + 5
    with pytest.raises(_lib.LatteHipError, match="key_bits"):
        ops.sort_segments(bad.cuda(), 20)
    ops.sort_segments(bad.cuda(), 21)                              # the same keys under an honest promise


@pytest.mark.parametrize("case", ["tiny", "b1_32k", "nusc_cp05"])
def test_cr_block_kernels_on_a_plain_dict_match_reference_golden(case):
    """CR_Block (lt_cr_loss_fwd / lt_cr_loss_bwd) fed the way a caller outside this package would: a plain dict of per-anchor
    LISTS (the reference's HR_Block output, here the golden arrays themselves) -- no pre-split tensors, no device offsets.
    Loss against the reference's own CR_Block value; gradient against the oracle's autograd; every flag combination against
    the oracle (ety_weight / conf_mask off, self_corr)."""
    from latte_amd.hr_module import CR_Block
    z, cfg, _ = load_hr_case(case)
    tag = "32"
    B = cfg["B"]
    ac = [torch.from_numpy(z[f"{tag}_ac_logits_{i}"]).cuda().requires_grad_(True) for i in range(B)]
    base = {
        "ac_logits": ac, "tr_logits": [torch.from_numpy(z[f"{tag}_tr_logits_{i}"]).cuda() for i in range(B)],
        "ac_voxels": [torch.from_numpy(z[f"{tag}_ac_voxels_{i}"]).cuda() for i in range(B)],
        "tr_voxels": [torch.from_numpy(z[f"{tag}_tr_voxels_{i}"]).cuda() for i in range(B)],
        "ac_p2v_indices": [torch.from_numpy(z[f"{tag}_ac_p2v_indices_{i}"]).cuda() for i in range(B)],
        "tr_p2v_indices": [torch.from_numpy(z[f"{tag}_tr_p2v_indices_{i}"]).cuda() for i in range(B)],
        "tr_conf_mask": torch.from_numpy(z[f"{tag}_tr_conf_mask"]).cuda(),
        "ac_pc_ety": torch.from_numpy(z[f"{tag}_ac_pc_ety"]).cuda(), "tr_pc_ety": torch.from_numpy(z[f"{tag}_tr_pc_ety"]).cuda(),
    }
    loss = CR_Block(reduction="mean", conf_mask=True, ety_weight=True)(base)
    want = float(z[f"{tag}_cr_loss"])
    assert abs(float(loss.detach()) - want) < 1e-4 * abs(want)
    loss.backward()
    cpu = {k: ([t.detach().cpu() for t in v] if isinstance(v, list) else v.cpu()) for k, v in base.items()}
    for kw in (dict(conf_mask=True, ety_weight=True), dict(conf_mask=False, ety_weight=True),
               dict(conf_mask=True, ety_weight=False), dict(conf_mask=False, ety_weight=False)):
        for self_corr in (False, True):
            ac_ref = [t.clone().requires_grad_(True) for t in cpu["ac_logits"]]
            ref = hr_ref.cr_forward(dict(cpu, ac_logits=ac_ref), self_corr=self_corr, **kw)
            for t in ac:
                t.grad = None
            got = CR_Block(reduction="mean", **kw)(base, self_corr=self_corr)
            assert abs(float(got.detach()) - float(ref.detach())) < 1e-4 * abs(float(ref.detach())), (kw, self_corr)
            ref.backward()
            (got * 2.0).backward()                       # the incoming gradient is used
            for g, r in zip(ac, ac_ref):
                assert float((g.grad.cpu() - 2.0 * r.grad).norm()) <= 1e-4 * float((2.0 * r.grad).norm()) + 1e-12, (kw, self_corr)
    # zero probabilities (Q6): clamped to 1e-8 in the loss, no gradient through the clamped anchor entries
    a0 = base["ac_logits"][0].detach().clone()
    a0[:5, 0] = 0.0
    a0.requires_grad_(True)
    t0 = base["tr_logits"][0].clone()
    t0[5:9, 1] = 0.0
    d2 = dict(base, ac_logits=[a0] + base["ac_logits"][1:], tr_logits=[t0] + base["tr_logits"][1:])
    cpu2 = {k: ([t.detach().cpu() for t in v] if isinstance(v, list) else v.cpu()) for k, v in d2.items()}
    a0_ref = cpu2["ac_logits"][0].clone().requires_grad_(True)
    ref = hr_ref.cr_forward(dict(cpu2, ac_logits=[a0_ref] + cpu2["ac_logits"][1:]), conf_mask=False, ety_weight=True)
    got = CR_Block(reduction="mean", conf_mask=False, ety_weight=True)(d2)
    assert np.isfinite(float(got.detach())) and abs(float(got.detach()) - float(ref.detach())) < 1e-4 * abs(float(ref.detach()))
    got.backward()
    ref.backward()
    assert bool((a0.grad[:5, 0] == 0).all())
    assert float((a0.grad.cpu() - a0_ref.grad).norm()) <= 1e-4 * float(a0_ref.grad.norm())
