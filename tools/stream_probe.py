#!/usr/bin/env python3
"""Experiment: two no-grad SPVCNN forwards over the same batch, sequential vs on two HIP streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from latte_amd import synth
from latte_amd.spvcnn import Net3DSeg
from latte_amd.tta_step import LatteStep
from latte_amd.workload import build_batch, hr_cfg_of
w = synth.workload("nuscenes_34k_w5")
torch.manual_seed(1)
model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": 1.0}).cuda()
step = LatteStep(model, w["C"], hr_cfg_of(w))
batch = build_batch(w, device="cuda")
m = step.ema_model
with torch.no_grad():
    for _ in range(3):
        m(batch)   # maps cached on the batch after the first call
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        a = m(batch); b = m(batch)
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) / 5
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(2):
        with torch.cuda.stream(s1): a = m(batch)
        with torch.cuda.stream(s2): b = m(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        with torch.cuda.stream(s1): a = m(batch)
        with torch.cuda.stream(s2): b = m(batch)
    torch.cuda.synchronize()
    par = (time.perf_counter() - t0) / 5
    # host-only cost estimate: one forward's Python time when the GPU queue is empty is not observable directly;
    # report both
print(f"two forwards sequential {1e3*seq:.2f} ms, two streams {1e3*par:.2f} ms")
