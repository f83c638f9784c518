#!/usr/bin/env python3
"""Micro-bench of the BatchNorm kernels (stats, apply, backward reduce + apply) at U-Net-like layer shapes:
us per call and GB/s of algorithmic traffic.   LATTE_HIP_LIB=<other .so> python tools/bn_bench.py   for an A/B."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from latte_amd import ops  # noqa: E402

SHAPES = [(209000, 32), (209000, 96), (120000, 32), (120000, 64), (50000, 64), (50000, 128), (15000, 128), (15000, 256),
          (4000, 256), (1200, 256)]


def timed(fn, n=30):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


def main():
    tot = {"apply": 0.0, "apply_res": 0.0, "bwd_y": 0.0, "bwd_x": 0.0, "stats": 0.0}
    for V, C in SHAPES:
        x = torch.randn(V, C, device="cuda")
        res = torch.randn(V, C, device="cuda")
        dy = torch.randn(V, C, device="cuda")
        gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
        mean, var = ops.bn_stats(x)
        y = ops.bn_apply(x, mean, var, gamma, beta, 1e-5, None, True)
        nbytes = V * C * 4
        rows = [("stats", lambda: ops.bn_stats(x), 1),
                ("apply", lambda: ops.bn_apply(x, mean, var, gamma, beta, 1e-5, None, True, out=y), 2),
                ("apply_res", lambda: ops.bn_apply(x, mean, var, gamma, beta, 1e-5, res, True, out=y), 3),
                ("bwd_y", lambda: ops.bn_bwd(dy, x, y, mean, var, gamma, 1e-5, True, need_dres=True), 7),
                ("bwd_x", lambda: ops.bn_bwd(dy, x, None, mean, var, gamma, 1e-5, True, beta=beta), 5)]
        line = f"V={V:7d} C={C:4d}"
        for name, fn, passes in rows:
            us = timed(fn)
            tot[name] += us
            line += f"  {name} {us:7.1f} us {passes * nbytes / us / 1e3:6.0f} GB/s"
        print(line, flush=True)
    print("sum us:", {k: round(v, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()
