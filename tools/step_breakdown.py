#!/usr/bin/env python3
"""Wall-clock breakdown of one TTA step by segment (synchronising timers; diagnostic only, not the bench)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from latte_amd import synth
from latte_amd.spvcnn import Net3DSeg
from latte_amd.tta_step import LatteStep
from latte_amd.workload import build_batch, hr_cfg_of

name = sys.argv[1] if len(sys.argv) > 1 else "nuscenes_34k_w5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = synth.workload(name)
torch.manual_seed(1)
model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": 1.0}).cuda()
step = LatteStep(model, w["C"], hr_cfg_of(w))
batch = build_batch(w, device="cuda")
for _ in range(3):
    step.step(batch)
step.timers = {}
for _ in range(steps):
    step.step(batch)
tot = sum(step.timers.values())
print(json.dumps({k: round(1e3 * v / steps, 2) for k, v in step.timers.items()} | {"total_ms": round(1e3 * tot / steps, 2)}))
