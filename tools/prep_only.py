#!/usr/bin/env python3
"""SPVCNN.prepare() alone, N times (for a rocprofv3 kernel trace of the backbone's index building on an idle GPU):
  rocprofv3 --kernel-trace --stats ... -- python tools/prep_only.py [workload] [runs]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from latte_amd import synth
from latte_amd.spvcnn import Net3DSeg
from latte_amd.tensor import SparseTensor
from latte_amd.workload import build_batch
w = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "a2d2_kitti_128k")
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": 1.0}).cuda()
x = build_batch(w, device="cuda")["lidar"]
for _ in range(runs):
    model.net_3d.prepare(SparseTensor(x.F, x.C, x.s))
torch.cuda.synchronize()
