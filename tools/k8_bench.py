#!/usr/bin/env python3
"""The 2^3 stride-2 convs and their transposes (spvcnn.py:36,42,48,54,60,69,78,87) at the a2d2_kitti_128k sizes: the
LDS-staged-weights kernel (spconv_ls.hip, variant 5000 = the default for these maps) against the general split-j kernel
(variant 5001), per U-Net level and direction, HIP events, best of 3 rounds.

  python tools/k8_bench.py [--out gpurun_out/k8.json]
direction "up"   = one pair per output row (transposed conv forward, strided conv backward-input): reverse plan
direction "down" = ~2-3 pairs per output row (strided conv forward, transposed conv backward-input): forward plan
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (level stride, direction, Cin, Cout): the step's 2^3 launches (profiles/r02_conv_table_a2d2.json)
SHAPES = [(1, "up", 96, 96), (1, "up", 32, 32), (2, "up", 128, 96), (2, "up", 32, 32), (4, "up", 256, 128), (4, "up", 64, 64),
          (8, "up", 256, 256), (8, "up", 128, 128),
          (1, "down", 32, 32), (1, "down", 96, 96), (2, "down", 32, 32), (2, "down", 96, 128), (4, "down", 64, 64),
          (4, "down", 128, 256), (8, "down", 128, 128), (8, "down", 256, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default=None, help="up | down")
    ap.add_argument("--tag", default="")
    ap.add_argument("--stamps", action="store_true", help="in-kernel time stamps of the LDS-staged kernel")
    ap.add_argument("--shape", default=None, help="stride,Cin,Cout: only this shape")
    ap.add_argument("--diag", action="store_true", help="also time the diagnostic variants: no stores (1), cache-resident rows (2), "
                    "no row loads (8), neither (9)")
    args = ap.parse_args()
    import torch
    from latte_amd import ops, synth
    from latte_amd import sparse_modules as spnn
    from latte_amd.tensor import SparseTensor
    from latte_amd.workload import build_batch
    from latte_amd.point_voxel import initial_voxelize
    from latte_amd.tensor import PointTensor

    dev = torch.device("cuda", 0)
    w = synth.workload("a2d2_kitti_128k")
    lidar = build_batch(w, device=dev)["lidar"]
    z = PointTensor(lidar.F, lidar.C.float())
    x = initial_voxelize(z, 0.05, 0.05)
    levels = {}
    s = 1
    while s <= 8:
        km, oc = spnn.build_kernel_map(x, 2, 2, 1)
        levels[s] = (km, x.C.shape[0], oc.shape[0])
        nx = SparseTensor(torch.zeros(oc.shape[0], 4, device=dev), oc, 2 * s)
        nx.cmaps, nx.kmaps = x.cmaps, x.kmaps
        nx.cmaps[nx.stride] = oc
        x, s = nx, 2 * s

    def timed(run):
        for _ in range(2):
            run()
        best = float("inf")
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, 1e3 * e0.elapsed_time(e1) / args.reps)
        return best

    rows = []
    for s, direction, cin, cout in SHAPES:
        if args.only and direction != args.only:
            continue
        if args.shape and [int(v) for v in args.shape.split(",")] != [s, cin, cout]:
            continue
        km, v_fine, v_coarse = levels[s]
        if direction == "up":      # coarse rows in, fine rows out: one pair per output row
            plan, vin, vout = km.reverse_plan(), v_coarse, v_fine
        else:
            plan, vin, vout = km.plan(), v_fine, v_coarse
        feats = torch.randn(vin, cin, device=dev)
        pw = ops.pack_weights(torch.randn(8, cin, cout, device=dev) * 0.05, 0)
        M = v_fine
        ref = ops.spconv_plan(feats, pw, plan, 8, vout, cout, variant=5001)
        got = ops.spconv_plan(feats, pw, plan, 8, vout, cout, variant=5000)
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, (s, direction, cin, cout, err)
        row = dict(tag=args.tag, stride=s, direction=direction, Cin=cin, Cout=cout, Vin=vin, Vout=vout, pairs=M, err_vs_general=err)
        extra = [(f"ls_dbg{d}", 5010 + d) for d in (1, 2, 8, 9)] if args.diag else []
        for name, v in [("ls", 5000), ("general", 5001)] + extra:
            us = timed(lambda: ops.spconv_plan(feats, pw, plan, 8, vout, cout, variant=v))
            row[name + "_us"] = round(us, 1)
            row[name + "_tflops"] = round(2.0 * M * cin * cout / us / 1e6, 1)
            row[name + "_GBs"] = round(4.0 * (M * cin + vout * cout) / us / 1e3, 1)     # gathered rows + written rows
        if args.stamps:
            import ctypes
            import numpy as np
            from latte_amd import _lib
            for d in (4, 13):
                cap = 8 * 8 * 8192
                buf = torch.zeros(cap, dtype=torch.int64, device=dev)
                _lib.call("lt_conv_debug_stamps", ctypes.c_void_p(buf.data_ptr()), cap)
                ops.spconv_plan(feats, pw, plan, 8, vout, cout, variant=5010 + d)
                torch.cuda.synchronize()
                _lib.call("lt_conv_debug_stamps", None, 0)
                st = buf.cpu().numpy().reshape(-1, 8)
                st = st[st[:, 0] != 0]
                life = st[:, 3] - st[:, 0]
                pro, loop, tail = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
                mma, stage, epi, adv = st[:, 4], st[:, 5], st[:, 6] & 0xFFFFFFFF, st[:, 6] >> 32
                hw, xcc = st[:, 7] & 0xFFFFFFFF, (st[:, 7] >> 32) & 0xF
                key = (((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)) * 4 + ((hw >> 4) & 3)
                span = max(st[xcc == x, 3].max() - st[xcc == x, 0].min() for x in np.unique(xcc))
                occ = np.array([life[key == k].sum() for k in np.unique(key)]) / span
                print(f"  stamps dbg={d}: waves {len(st)} span {span} cyc; life mean {life.mean():.0f} max {life.max():.0f}; "
                      f"prologue {pro.mean():.0f} loop {loop.mean():.0f} tail {tail.mean():.0f} | in loop: mma {mma.mean():.0f} "
                      f"stage {stage.mean():.0f} epilogue {epi.mean():.0f} advance {adv.mean():.0f}; SIMDs {len(np.unique(key))} "
                      f"waves/SIMD mean {occ.mean():.2f} max {occ.max():.2f}; start spread {st[:, 0].max() - st[:, 0].min()}",
                      flush=True)
        rows.append(row)
        print(json.dumps(row), flush=True)
    if args.out:
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
