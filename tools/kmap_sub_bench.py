#!/usr/bin/env python3
"""3^3 submanifold kernel map at the a2d2 sizes: lt_kmap_build (27 probes per row) against lt_kmap_build_sub (14 probes, the
rest mirrored), per level of the backbone.   python tools/kmap_sub_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from latte_amd import functional as LF
from latte_amd import ops, synth
from latte_amd.workload import build_batch

w = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "a2d2_kitti_128k")
batch = build_batch(w, device="cuda")
c = batch["lidar"].C.int().contiguous()
from latte_amd.point_voxel import initial_voxelize
from latte_amd.tensor import PointTensor
x0 = initial_voxelize(PointTensor(batch["lidar"].F, batch["lidar"].C.float()), 1.0, 1.0)
c = x0.C.int().contiguous()


def us(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


s = 1
while c.shape[0] > 1000 and s <= 16:
    off = LF.get_kernel_offsets(3, s, 1, device=c.device)
    full = ops.kmap_build(c, c, off)
    half = ops.kmap_build(c, c, off, submanifold=True)
    assert torch.equal(full, half)
    print(f"stride {s:2d}  {c.shape[0]:7d} rows  27 probes {us(lambda: ops.kmap_build(c, c, off)):7.1f} us   "
          f"14 probes + mirror {us(lambda: ops.kmap_build(c, c, off, submanifold=True)):7.1f} us")
    c = ops.downsample_coords(c, 2 * s)
    s *= 2
