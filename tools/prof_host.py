#!/usr/bin/env python3
"""cProfile of the host side of one TTA step (diagnostic)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from latte_amd import synth
from latte_amd.spvcnn import Net3DSeg
from latte_amd.tta_step import LatteStep
from latte_amd.workload import build_batch, hr_cfg_of
w = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "nuscenes_34k_w5")
torch.manual_seed(1)
model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": 1.0}).cuda()
step = LatteStep(model, w["C"], hr_cfg_of(w))
batch = build_batch(w, device="cuda")
PRE = len(sys.argv) > 2 and sys.argv[2] == "prefetch"   # main thread only: the helper thread's work is not profiled
for _ in range(3):
    step.step(batch, next_batch=batch if PRE else None)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step.step(batch, next_batch=batch if PRE else None)
torch.cuda.synchronize()
pr.disable()
step.close()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(60)
st.sort_stats("tottime").print_stats(45)
