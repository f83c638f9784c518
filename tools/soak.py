#!/usr/bin/env python3
"""Soak run of the TTA step: N iterations over a rotation of different batches with the prefetch one iteration ahead
(side stream + helper thread), losses checked finite every step, parameters checked finite at the end.

  python tools/soak.py [--steps 150] [--workload nuscenes_34k_w5] [--batches 3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--workload", default="nuscenes_34k_w5")
    ap.add_argument("--batches", type=int, default=3)
    args = ap.parse_args()
    import math
    import torch
    from latte_amd import synth
    from latte_amd.spvcnn import Net3DSeg
    from latte_amd.tta_step import LatteStep
    from latte_amd.workload import build_batch, hr_cfg_of
    dev = torch.device("cuda", 0)
    w = synth.workload(args.workload)
    torch.manual_seed(1)
    model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": 1.0}).to(dev)
    step = LatteStep(model, w["C"], hr_cfg_of(w))
    batches = [build_batch(w, device=dev, scan_offset=100 * i, seed=i) for i in range(args.batches)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    t_mark, i_mark = t0, 0
    for i in range(args.steps):
        b = batches[i % len(batches)]
        nb = batches[(i + 1) % len(batches)] if i + 1 < args.steps else None
        losses = step.step(b, next_batch=nb)
        if i % 25 == 0 or i == args.steps - 1:
            vals = {k: float(v) for k, v in losses.items()}    # syncs
            assert all(math.isfinite(v) for v in vals.values()), (i, vals)
            last = vals
            now = time.perf_counter()
            print(i, {k: round(v, 5) for k, v in vals.items()}, f"{1e3 * (now - t_mark) / max(i - i_mark, 1):.2f} ms/step", flush=True)
            t_mark, i_mark = now, i
    step.close()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for n, p in list(step.model.named_parameters()) + list(step.ema_model.named_parameters()):
        assert torch.isfinite(p).all(), n
    print(f"{args.steps} steps over {len(batches)} batches of {args.workload}: {1e3 * dt / args.steps:.2f} ms per step including "
          f"the first step's warm-up, "
          f"all losses and parameters finite; last {last}")


if __name__ == "__main__":
    main()
