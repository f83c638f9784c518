#!/usr/bin/env python3
"""Host-issue vs GPU-completion timeline of the TTA step, without synchronising inside the step (diagnostic).
For every phase boundary: host time at which the boundary was reached, and the time the main stream reached it."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from latte_amd import synth
from latte_amd.spvcnn import Net3DSeg
from latte_amd.tta_step import LatteStep
from latte_amd.workload import build_batch, hr_cfg_of

name = sys.argv[1] if len(sys.argv) > 1 else "nuscenes_34k_w5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w = synth.workload(name)
torch.manual_seed(1)
model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": 1.0}).cuda()
step = LatteStep(model, w["C"], hr_cfg_of(w))
batch = build_batch(w, device="cuda")
for _ in range(3):
    step.step(batch)
torch.cuda.synchronize()
step.trace = []
e0 = torch.cuda.Event(enable_timing=True)
e0.record()
t0 = time.perf_counter()
for _ in range(steps):
    step.step(batch)
step._tick("end")
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"{steps} steps: {1e3 * (t1 - t0) / steps:.2f} ms/step")
print(f"{'phase reached':16s} {'host ms':>9s} {'gpu ms':>9s} {'gpu-host':>9s}")
for nm, th, ev in step.trace:
    print(f"{nm:16s} {1e3 * (th - t0):9.2f} {e0.elapsed_time(ev):9.2f} {e0.elapsed_time(ev) - 1e3 * (th - t0):9.2f}")
