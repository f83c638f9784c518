import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def scans_small():
    """3 quantized synthetic scans (2048 points each) collated: coords [V,4] int32, feats [V,4] fp32."""
    from latte_amd import synth
    from oracle import ts_ref
    cs, fs, invs = [], [], []
    for j in range(3):
        pts, inten = synth.make_scan(2048, j, beams=16)
        c, f, inds, inv = ts_ref.quantize_scan(pts, inten, scale=20)
        cs.append(c)
        fs.append(f)
        invs.append(inv)
    coords, feats = ts_ref.sparse_collate(cs, fs)
    return dict(coords=coords, feats=feats, inverse=invs, sizes=[c.shape[0] for c in cs])
