#!/usr/bin/env python3
"""lt_class_quantile_mask / lt_class_median_mask / lt_quantile at the step's sizes, against torch: value check + time per
call.  LT_QUANTILE_SORT=1 times the radix-sort path instead of the selection path.   python tools/quantile_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from latte_amd import ops


def us(fn, reps=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


torch.manual_seed(0)
print("path:", "sort" if os.environ.get("LT_QUANTILE_SORT") == "1" else "select")
for n, C, scale in ((150000, 10, 3.0), (300000, 10, 3.0), (786432, 10, 3.0), (208896, 16, 3.0), (786432, 10, 0.01)):
    # scale 0.01: a network that says "uniform" everywhere -- all entropies within 1e-4 of log2(C)/3.3 (one bin of every level)
    p = torch.softmax(scale * torch.randn(n, C, device="cuda"), 1)
    ety = -(p * torch.log2(p + 1e-30)).sum(1) / 3.3
    lab = p.argmax(1).int()
    m = ops.class_quantile_mask(ety, lab, C, 0.9)
    want = torch.ones(n, dtype=torch.bool, device="cuda")
    for c in range(C):
        sel = lab == c
        if sel.any():
            want[sel] = ~(ety[sel] > torch.quantile(ety[sel], 0.9))
    bad = int((m.bool() != want).sum())
    conf = p.max(1).values
    k = ops.class_median_mask(conf, lab, C, 0.9)
    wk = torch.ones(n, dtype=torch.bool, device="cuda")
    for c in range(C):
        sel = lab == c
        if sel.any():
            t = min(float(conf[sel].median()), 0.9)
            wk[sel] = ~(conf[sel] < t)
    bad2 = int((k.bool() != wk).sum())
    qv = float(ops.quantile(ety, 0.9))
    qw = float(torch.quantile(ety, 0.9)) if n < (1 << 24) else float("nan")
    print(f"n {n:7d} C {C:2d}  class quantile {us(lambda: ops.class_quantile_mask(ety, lab, C, 0.9)):7.1f} us (mismatch {bad})   "
          f"class median {us(lambda: ops.class_median_mask(conf, lab, C, 0.9)):7.1f} us (mismatch {bad2})   "
          f"quantile {us(lambda: ops.quantile(ety, 0.9)):7.1f} us ({qv!r} vs torch {qw!r})")
