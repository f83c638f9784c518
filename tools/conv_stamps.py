#!/usr/bin/env python3
"""In-kernel time stamps of the split-k conv kernel (variant 70000 + code): where a wave's lifetime goes
(prologue / main loop / epilogue), cycles per slab in the main loop, the in-kernel clock, and how busy each CU is.

  python tools/conv_stamps.py --stride 8 --channels 256 --code 1244
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=6)
    ap.add_argument("--stride", type=int, default=8)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--code", type=int, default=1244)
    ap.add_argument("--dbg", type=int, default=4)
    ap.add_argument("--points", type=int, default=34816)
    ap.add_argument("--beams", type=int, default=32)
    ap.add_argument("--front", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    from latte_amd import _lib, ops, synth
    from latte_amd import sparse_modules as spnn
    from latte_amd.point_voxel import initial_voxelize
    from latte_amd.quantize import quantize_scan_gpu
    from latte_amd.tensor import PointTensor, SparseTensor

    dev = torch.device("cuda", 0)
    coords = []
    for b in range(args.scans):
        p, i = synth.make_scan(args.points, b, beams=args.beams, full_360=not args.front)
        c, f, _, _ = quantize_scan_gpu(torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev))
        coords.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int32, device=dev)], 1))
    C4 = torch.cat(coords, 0)
    x = initial_voxelize(PointTensor(torch.zeros(C4.shape[0], 4, device=dev), C4.float()), 0.05, 0.05)
    s = 1
    while s < args.stride:
        _, oc = spnn.build_kernel_map(x, 2, 2, 1)
        nx = SparseTensor(torch.zeros(oc.shape[0], 4, device=dev), oc, 2 * s)
        nx.cmaps, nx.kmaps = x.cmaps, x.kmaps
        nx.cmaps[nx.stride] = oc
        x, s = nx, 2 * s
    kmap, _ = spnn.build_kernel_map(x, 3, 1, 1)
    nbr = kmap.results
    V = nbr.shape[1]
    M = int((nbr >= 0).sum())
    plan = ops.kmap_plan(nbr)
    C = args.channels
    feats = torch.randn(V, C, device=dev)
    w = torch.randn(27, C, C, device=dev) * 0.05
    packed = ops.pack_weights(w, 0)
    W = args.code % 10
    ntw = (args.code // 10) % 10
    cap = 8 * 8 * (V // 16 + 64) * max(C // (16 * ntw), 1) * 4
    buf = torch.zeros(cap, dtype=torch.int64, device=dev)
    _lib.call("lt_conv_debug_stamps", ctypes.c_void_p(buf.data_ptr()), cap)
    variant = (3 + args.dbg) * 10000 + args.code
    for _ in range(20):  # warm clocks
        ops.spconv_plan(feats, packed, plan, 27, V, C, variant=args.code)
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.spconv_plan(feats, packed, plan, 27, V, C, variant=variant)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1)
    _lib.call("lt_conv_debug_stamps", None, 0)
    st = buf.cpu().numpy().reshape(-1, 8)
    st = st[st[:, 0] != 0]
    t0, t1, t2, t3, r0, r3 = (st[:, i].astype(np.int64) for i in range(6))
    tI, tC = r0 >> 40, r3 >> 40
    r0, r3 = r0 & ((1 << 40) - 1), r3 & ((1 << 40) - 1)
    U = st[:, 6] & 0xFFFFFFFF
    nact = st[:, 6] >> 32
    hw = st[:, 7] & 0xFFFFFFFF
    xcc = (st[:, 7] >> 32) & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    simd = (hw >> 4) & 0x3
    print(f"V={V} M={M} C={C} code={args.code}: {us:.1f} us (stamped launch), {2.0 * M * C * C / us / 1e6:.1f} TFLOP/s, waves {len(st)}")
    span_rt = (r3.max() - r0.min()) / 100.0  # us (100 MHz)
    span_cy = t3.max() - t0.min()
    # per-XCD counters are not synchronised: clock from per-wave deltas
    clk = np.median((t3 - t0)[(r3 - r0) > 200] / ((r3 - r0)[(r3 - r0) > 200] / 100.0)) if ((r3 - r0) > 200).any() else float("nan")
    print(f"kernel span by s_memrealtime {span_rt:.1f} us; in-kernel clock (median of long waves) {clk / 1e3:.3f} GHz")
    life, pro, loop, epi = t3 - t0, t1 - t0, t2 - t1, t3 - t2
    slabs = np.maximum((U - (np.arange(len(st)) % 1) * 0) // 1, 0)
    print(f"wave lifetime cycles: mean {life.mean():.0f}  prologue {pro.mean():.0f} ({100 * pro.sum() / life.sum():.1f} %)  "
          f"main loop {loop.mean():.0f} ({100 * loop.sum() / life.sum():.1f} %)  epilogue {epi.mean():.0f} ({100 * epi.sum() / life.sum():.1f} %)")
    # slabs of this wave: ceil((U - wave)/W): approximate with U / W
    per_slab = loop.sum() / max((U / W).sum(), 1)
    mf = 2 * ntw * 4  # MFMAs per slab at full strip activity (SR=2)
    print(f"main loop: {per_slab:.0f} wave-cycles per slab ({mf} MFMAs = {mf * 32} MFMA-pipe cycles when both strips are active)")
    print(f"main loop split: issue part {tI.sum() / max((U / W).sum(), 1):.0f} cycles per slab, compute part {tC.sum() / max((U / W).sum(), 1):.0f}")
    # per-SIMD concurrency: sum of wave lifetimes on a SIMD / span
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simd_key = key * 4 + simd
    uniq = np.unique(simd_key)
    print(f"SIMDs seen {len(uniq)}, CUs seen {len(np.unique(key))}")
    occ = np.array([life[simd_key == k].sum() for k in uniq]) / span_cy
    print(f"resident waves per SIMD (time-average over the kernel span): mean {occ.mean():.2f} min {occ.min():.2f} max {occ.max():.2f}")
    # end time distribution per XCD (tail)
    for xcd in range(8):
        m = xcc == xcd
        if m.any():
            print(f"  XCD {xcd}: waves {m.sum():5d}  span {(t3[m].max() - t0[m].min()):8d} cyc  sum(U) {int(U[m].sum() / W):7d} "
                  f"last-start {(t0[m].max() - t0[m].min()):8d}")
    # timeline: fraction of waves in main loop over time (20 bins), using realtime per XCD is not comparable -> use XCD 0
    m = xcc == xcc[0]
    a, b = t0[m].min(), t3[m].max()
    edges = np.linspace(a, b, 21)
    print("XCD", int(xcc[0]), "waves in prologue / main / epilogue per time bin:")
    for i in range(20):
        mid = 0.5 * (edges[i] + edges[i + 1])
        n_pro = ((t0[m] <= mid) & (mid < t1[m])).sum()
        n_main = ((t1[m] <= mid) & (mid < t2[m])).sum()
        n_epi = ((t2[m] <= mid) & (mid < t3[m])).sum()
        print(f"   {i:2d}: {n_pro:4d} {n_main:4d} {n_epi:4d}")


if __name__ == "__main__":
    main()
