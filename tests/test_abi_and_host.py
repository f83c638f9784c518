"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol include/*.h declares,
the ctypes table covers them all, the product path refuses to run without a GPU, and host-side helpers behave."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in ("latte_hip.h", "latte_hip_st.h", "latte_hip_net.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(lt_[a-z0-9_]+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    from latte_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared()
    assert len(declared) > 40
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes signature table binds exactly the declared surface
    assert set(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ declared
    assert _lib.load().lt_version() >= 100


def test_size_queries_need_no_gpu():
    from latte_amd import _lib
    lib = _lib.load()
    assert lib.lt_table_bytes(1000) >= 2 * 1000 * 12
    assert lib.lt_kmap_plan_bytes(27, 187419) >= 187419 * 4 + 27 * 187419 * 4
    assert lib.lt_spconv_packed_floats(27, 96, 96) == 27 * 96 * 96
    for fn in ("lt_unique_workspace_bytes", "lt_downsample_workspace_bytes", "lt_kmap_workspace_bytes",
               "lt_sort_segments_workspace_bytes", "lt_class_quantile_workspace_bytes", "lt_kmap_order_workspace_bytes",
               "lt_quantile_workspace_bytes",
               "lt_kmap_plan_workspace_bytes"):
        assert getattr(lib, fn)(100000) > 0


def test_product_path_has_no_cpu_fallback():
    from latte_amd import _lib, ops
    with pytest.raises(_lib.LatteHipError):
        ops.sphash(torch.zeros(4, 4, dtype=torch.int32))          # CPU tensor -> refused, not computed
    with pytest.raises(_lib.LatteHipError):
        ops.spconv(torch.zeros(4, 16), torch.zeros(27 * 16 * 16), torch.zeros(27, 4, dtype=torch.int32), 16)


def test_argument_errors_come_back_as_messages():
    from latte_amd import _lib
    lib = _lib.load()
    rc = lib.lt_spconv_plan_fwd(None, None, None, None, 10, 10, 20, 16, 27, None)   # Cin % 16 != 0
    assert rc == -1 and b"multiple of 16" in lib.lt_last_error()
    rc = lib.lt_kmap_plan(None, 40, 10, None, 0, None, 0, None)                     # K > 32
    assert rc == -1 and b"bad sizes" in lib.lt_last_error()


def test_kernel_offsets_follow_torchsparse_order():
    from latte_amd.functional import get_kernel_offsets
    from oracle import ts_ref
    for size, stride in ((3, 1), (3, 4), (2, 1), (2, 8)):
        assert np.array_equal(get_kernel_offsets(size, stride, 1).numpy(), ts_ref.get_kernel_offsets(size, stride))
    o3 = get_kernel_offsets(3, 1, 1).numpy()
    assert o3[0].tolist() == [-1, -1, -1] and o3[1].tolist() == [0, -1, -1] and o3[13].tolist() == [0, 0, 0]  # x fastest
    o2 = get_kernel_offsets(2, 1, 1).numpy()
    assert o2[1].tolist() == [0, 0, 1] and o2[4].tolist() == [1, 0, 0]                                        # z fastest


def test_hr_window_plan_matches_reference_rule():
    """frames of the same scene within floor(temp_wd_size/2) scan ids (latte/models/hr_module.py:192-196)"""
    from latte_amd.hr_module import HR_Block
    hr = HR_Block(temp_wd_size=3, max_range=40, voxel_size=0.2, conf_perc=0.9)
    assert hr.temp_bond == 1 and hr.min_offset == -201
    pts = [torch.zeros(5, 3), torch.zeros(7, 3), torch.zeros(3, 3), torch.zeros(4, 3)]
    b = dict(scene_name=["scene-1", "scene-1", "scene-1", "scene-2"], scan_ID=[0, 1, 2, 0], ori_points=pts)
    segs, h_start, h_sizes = hr._plan(b, "ori_points", b, "ori_points")
    assert [(a, f) for a, f, _, _, _ in segs] == [(0, 0), (0, 1), (1, 0), (1, 1), (1, 2), (2, 1), (2, 2), (3, 3)]
    assert all(src == row for _, _, src, _, row in segs)      # contiguous windows: logits rows == point rows
    assert h_start.tolist() == [0, 5, 12, 15, 19] and h_sizes == [5, 7, 3, 4]
    # non-contiguous windows (a shuffled loader): the points come from the matching frames, the logit rows are taken
    # sequentially from the first matching frame on, one slice per frame (hr_module.py:203-220, quirk Q5)
    b2 = dict(b, scene_name=["scene-1", "scene-2", "scene-1", "scene-2"])
    segs2, _, _ = hr._plan(b2, "ori_points", b2, "ori_points")
    assert segs2 == [(0, 0, 0, 5, 0), (1, 1, 5, 7, 5), (1, 3, 15, 4, 12), (2, 2, 12, 3, 12), (3, 1, 5, 7, 5), (3, 3, 15, 4, 12)]


def test_synthetic_workloads_cover_baseline_configs():
    from latte_amd import synth
    for name, n in (("synthia_kitti_32k_b1", 32768), ("nuscenes_34k_w5", 34816), ("a2d2_kitti_128k", 131072)):
        w = synth.workload(name)
        p, i = synth.make_scan(w["n"], 0, beams=w["beams"], full_360=w["full_360"])
        assert p.shape == (n, 3) and i.shape == (n,) and p.dtype == np.float32
    p2, _ = synth.make_scan(34816, 0, beams=32, full_360=True)
    p3, _ = synth.make_scan(34816, 0, beams=32, full_360=True)
    assert np.array_equal(p2, p3)   # deterministic


def test_cross_entropy_matches_torch():
    """tta_step.cross_entropy (gather/sum formulation) against F.cross_entropy: value and gradient, with ignored labels and
    with class weights"""
    import torch
    import torch.nn.functional as F
    from latte_amd.tta_step import cross_entropy
    g = torch.Generator().manual_seed(5)
    for weight in (None, torch.rand(7, generator=g) + 0.5):
        x = torch.randn(300, 7, generator=g, requires_grad=True)
        y = torch.randint(0, 7, (300,), generator=g)
        y[::5] = -100
        a = cross_entropy(x, y, weight)
        ga, = torch.autograd.grad(a, x)
        b = F.cross_entropy(x, y, weight=weight)
        gb, = torch.autograd.grad(b, x)
        assert abs(float(a) - float(b)) < 1e-6 * abs(float(b))
        assert float((ga - gb).abs().max()) < 1e-7


def test_ema_util_mirrors_return_what_the_reference_returns():
    """configure_model_norm / configure_model_all / create_ema_model / update_ema_* of latte_amd.tta_step against the values
    the reference's own latte/common/utils/ema_util.py produced on the same module tree (tests/golden/ema_util_returns.json,
    made by tests/golden/make_ema_util_golden.py): the nested [model, [lists]] return value its callers unpack
    (latte/tta/tta.py:130-136), the trainable set, dropout, the BatchNorm statistics mode, the EMA warm-up rule."""
    import json
    import os
    from latte_amd.spvcnn import Net3DSeg
    from latte_amd import tta_step as T
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ema_util_returns.json")))

    def net():
        torch.manual_seed(0)
        return Net3DSeg(9, True, "SPVCNN", {"cr": 0.5})

    def describe(model):
        return {"trainable": sorted(n for n, p in model.named_parameters() if p.requires_grad),
                "training": bool(model.training),
                "dropout_p": {n: float(m.p) for n, m in model.named_modules() if "dropout" in type(m).__name__.lower()},
                "bn_track": sorted(n for n, m in model.named_modules()
                                   if "batchnorm" in type(m).__name__.lower() and m.track_running_stats)}

    for reset in (True, False):
        for fn in ("configure_model_norm", "configure_model_all"):
            r = getattr(T, fn)(net(), reset)
            want = gold[f"{fn}_{reset}"]
            assert isinstance(r, list) and len(r) == 2 and isinstance(r[0], torch.nn.Module)
            assert r[1] == want["lists"], fn
            assert describe(r[0]) == want["model"], fn
    model, _ = T.configure_model_norm(net(), True)          # how the reference's callers unpack it
    r = T.create_ema_model(model)
    assert r[1] == gold["create_ema_model"]["lists"] and describe(r[0]) == gold["create_ema_model"]["model"]
    assert any(p.requires_grad for p in r[0].parameters()) == gold["create_ema_model"]["requires_grad_any"]
    for fn in ("update_ema_norm", "update_ema_all"):
        for it in (None, 3):
            s, e = net(), net()
            with torch.no_grad():
                for p in s.parameters():
                    p.add_(1.0)
            getattr(T, fn)(e, s, 0.99, it)
            ref = net()
            for (n, pe), (_, pr) in zip(e.named_parameters(), ref.named_parameters()):
                assert abs(float((pe - pr).abs().max()) - gold[f"{fn}_{it}"]["moved"][n]) < 1e-5, (fn, it, n)
