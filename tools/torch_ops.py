#!/usr/bin/env python3
"""Which lines of latte_amd issue the torch (aten) kernels of a step: torch.profiler with stacks, device time of every
aten op attributed to the innermost latte_amd frame.   python tools/torch_ops.py [--top 40]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--workload", default="nuscenes_34k_w5")
    args = ap.parse_args()
    from latte_amd import synth
    from latte_amd.spvcnn import Net3DSeg
    from latte_amd.tta_step import LatteStep
    from latte_amd.workload import build_batch, hr_cfg_of
    w = synth.workload(args.workload)
    torch.manual_seed(1)
    dev = torch.device("cuda", 0)
    model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": 1.0}).to(dev)
    step = LatteStep(model, w["C"], hr_cfg_of(w))
    batch = build_batch(w, device=dev)
    for _ in range(3):
        step.step(batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step.step(batch)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.key_averages(group_by_input_shape=True):
        t = ev.self_device_time_total
        if not ev.key.startswith("aten::") or t <= 0:
            continue
        where = "?"
        for fr in ev.stack or []:
            if "latte_amd" in fr and "site-packages" not in fr:
                where = fr.strip().replace(ROOT + "/", "")
                break
        if where == "?":
            where = str(ev.input_shapes)[:120]
        k = (ev.key, where)
        agg[k][0] += ev.count
        agg[k][1] += t
    tot = sum(v[1] for v in agg.values())
    print(f"aten device time in one step: {tot / 1e3:.2f} ms")
    for (name, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f"{t:8.1f} us {n:4d}  {name:28s} {where}")


if __name__ == "__main__":
    main()
