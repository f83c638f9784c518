#!/usr/bin/env python3
"""List the torch-level host synchronisations of one TTA step (torch.cuda.set_sync_debug_mode), with call sites."""
import os, sys, warnings, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from latte_amd import synth
from latte_amd.spvcnn import Net3DSeg
from latte_amd.tta_step import LatteStep
from latte_amd.workload import build_batch, hr_cfg_of
w = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "tiny")
torch.manual_seed(1)
model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": 1.0}).cuda()
step = LatteStep(model, w["C"], hr_cfg_of(w))
batch = build_batch(w, device="cuda")
for _ in range(2):
    step.step(batch)
torch.cuda.synchronize()
sites = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    st = [f for f in traceback.extract_stack() if "/latte_amd/" in f.filename or f.filename.endswith("bench.py")]
    key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st[-3:]))
    sites[key] += 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step.step(batch)
torch.cuda.set_sync_debug_mode("default")
for k, n in sites.most_common():
    print(n, k)
print("total torch-level syncs per step:", sum(sites.values()))
