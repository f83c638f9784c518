#!/usr/bin/env python3
"""Headline benchmark: points/s of one Latte TTA step (3-D side) on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = teacher SPVCNN forward + student SPVCNN forward + 2x HR_Block (ST-voxel aggregation) + pseudo-label fusion +
2x CR_Block + cross-entropy + backward to the BatchNorm parameters + (N>1: one flat RCCL all-reduce of those grads) +
Adam + EMA teacher update -- latte/tta/tools/latte.py:156-323 without the 2-D SegFormer.
Workload at every N: BASELINE.json configs[1] (nuScenes-like 34 816-point 360-degree scans, 5-frame ST window, 6 scans
per step and GPU, 16 classes, SPVCNN cr=1.0, fp32); every rank owns its own sequence (weak scaling, no data-path
collective).  Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the dominant kernel family (sparse-conv fp32 MFMA kernel k_spconv_sj): algorithmic flops / HIP-event time of its
                launches of a sample of the timed steps (the library's conv launch timer, lt_conv_timer_begin/end)
  cpu_baseline  the CPU oracle (oracle/step_ref.py, a port: the reference's torchsparse backend is not installable)
                timed on this box's host cores on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense, spec
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="nuscenes_34k_w5")
    ap.add_argument("--cr", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scans", type=int, default=3, help="scans in the CPU-oracle sample")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not time the conv launches with HIP events")
    ap.add_argument("--all-kernel-events", action="store_true", help="time the conv launches of every timed step")
    ap.add_argument("--no-side-stream", action="store_true", help="run the input-only, sync-heavy work (index structures, "
                    "HR geometry) on the main stream too")
    ap.add_argument("--no-prefetch", action="store_true", help="build the index structures of a batch inside its own step "
                    "instead of one step ahead")
    ap.add_argument("--dump-conv", default=None, help="write the per-launch conv table of one step (JSON) here")
    ap.add_argument("--fwd-only", action="store_true", help="time teacher forward only (BASELINE config 5 style)")
    return ap.parse_args()


def cpu_baseline(w, cr, n_scans, seed=1):
    """Oracle TTA step (teacher fwd, student fwd, 2x HR, 2x CR, CE, backward) on `n_scans` scans of the workload."""
    import numpy as np
    import torch
    from latte_amd import synth
    from latte_amd.spvcnn import Net3DSeg
    from latte_amd.workload import hr_cfg_of
    from oracle import step_ref

    torch.manual_seed(seed)
    model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": cr})
    sd = model.state_dict()
    names = [f"{mn}.{pn}"[len("net_3d."):] for mn, mod in model.named_modules()
             if isinstance(mod, torch.nn.BatchNorm1d) and mn.startswith("net_3d.") for pn, _ in mod.named_parameters()]
    scans = [synth.make_scan(w["n"], j, beams=w["beams"], full_360=w["full_360"]) for j in range(n_scans)]
    ori = [torch.from_numpy(p) for p, _ in scans]
    fov = [(p[:, 0] > 0) & (p[:, 1].abs() < 0.8 * p[:, 0]) for p in ori]
    n_fov = int(sum(int(m.sum()) for m in fov))
    g = torch.Generator().manual_seed(0)
    l2s = 3 * torch.randn(n_fov, w["C"], generator=g)
    l2t = 3 * torch.randn(n_fov, w["C"], generator=g)
    cores = torch.get_num_threads()
    t0 = time.perf_counter()
    coords, feats, invs, sizes = step_ref.quantize_batch(scans)
    step_ref.tta_step_ref(sd, sd, coords, torch.from_numpy(feats), invs, sizes, ori, fov,
                          [synth.make_pose(j) for j in range(n_scans)], ["scene-1000"] * n_scans, list(range(n_scans)),
                          l2s, l2t, hr_cfg_of(w), grad_names=names)
    dt = time.perf_counter() - t0
    pts = n_scans * w["n"]
    return {"value": pts / dt, "unit": "points/s", "cores": int(cores), "kind": "port",
            "sample": f"one oracle TTA step (oracle/step_ref.py: quantize, teacher fwd, student fwd, 2x HR, 2x CR, CE, "
                      f"backward) on {n_scans} scans x {w['n']} points of the same workload, {dt:.1f} s, torch-CPU "
                      f"fp32 with {cores} threads; the reference's own backend (torchsparse v1.4.0 CUDA) cannot run here"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without the launcher: start it as a child (nothing has touched the GPU yet) and leave with its code
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    import torch
    import torch.distributed as dist
    from latte_amd import _lib, ops, synth
    from latte_amd.spvcnn import Net3DSeg
    from latte_amd.tta_step import LatteStep, shard_scan_offset
    from latte_amd.workload import build_batch, hr_cfg_of

    if os.environ.get("LATTE_BENCH_TRACE"):  # where a stuck multi-rank run stands: Python stacks of every rank after 60 s
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LATTE_BENCH_TRACE"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    _lib.load()
    # LATTE_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share GPUs)
    backend = os.environ.get("LATTE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        pg = dist.group.WORLD

    w = synth.workload(args.workload)
    torch.manual_seed(1)  # cfg.RNG_SEED (latte/common/config/base.py:173); same weights on every rank
    model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": args.cr}).to(dev)
    step = LatteStep(model, w["C"], hr_cfg_of(w), process_group=pg, side_stream=not args.no_side_stream)
    batch = build_batch(w, device=dev, scan_offset=shard_scan_offset(rank, w["B"]), seed=rank)
    n_points = batch["n_points"]

    def one_step():
        if args.fwd_only:
            with torch.no_grad():
                return step.ema_model(batch)
        # the synthetic workload feeds the same scans every step: `next_batch=batch` makes the step issue the NEXT
        # iteration's input-only structures (rebuilt from the coordinates, not reused) while its own backward runs
        return step.step(batch, next_batch=None if args.no_prefetch else batch)

    for _ in range(args.warmup):
        one_step()
    timed_kernels = not args.no_kernel_events
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # conv launches are timed (HIP events attached to the dispatches) in ONE of the timed steps, the second
    # (all of them with --all-kernel-events) -- an event pair costs ~10 us of launch path, 125 conv launches per step
    sampled = [i for i in range(args.steps) if args.all_kernel_events or i == 1 or args.steps < 2]
    if timed_kernels:
        ops.conv_timer_begin(1 << 16)
        ops.conv_timer_arm(False)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if timed_kernels and i in sampled:
            ops.conv_timer_arm(True)
            one_step()
            ops.conv_timer_arm(False)
        else:
            one_step()
    dt_host = time.perf_counter() - t0  # the host has issued every step (the GPU may still be working)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    trace = ops.conv_timer_end() if timed_kernels else None
    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    pairs = None
    if timed_kernels:
        # pair counts M of every conv launch of a step, from one more (untimed) step on every rank: the launch sequence
        # and the kernel maps are the same in every step (same scans), only the BN parameters move
        ops.CONV_PAIRS = []
        one_step()
        torch.cuda.synchronize()
        pairs = [int(m.item()) for m in ops.CONV_PAIRS]
        ops.CONV_PAIRS = None

    step.close()
    if rank == 0:
        roof = None
        if timed_kernels:
            ms, meta = trace
            n_s = len(sampled)
            per_step = len(ms) // n_s
            assert len(pairs) == per_step and per_step * n_s == len(ms), (len(pairs), per_step, len(ms))
            flops_step = sum(2.0 * m * int(cin) * int(cout) for m, (cin, cout, _k, _v) in zip(pairs, meta[:per_step]))
            ms_total = float(ms.astype("float64").sum())
            if args.dump_conv:
                us = 1e3 * ms.reshape(n_s, per_step).mean(0)
                table = [dict(i=i, Cin=int(c[0]), Cout=int(c[1]), K=int(c[2]), Vout=int(c[3]), M=m, us=round(float(u), 1),
                              tflops=round(2.0 * m * int(c[0]) * int(c[1]) / float(u) / 1e6, 2))
                         for i, (m, c, u) in enumerate(zip(pairs, meta[:per_step], us))]
                json.dump(table, open(args.dump_conv, "w"))
            launches = len(ms)
            flops_total = flops_step * n_s
            achieved = flops_total / (ms_total * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                    "kernel": "k_spconv_sj<SR,NTW,W> (split-j; +k_spconv_scalar for the 4-channel stem): fp32 MFMA sparse "
                              "conv, every launch of the sampled steps of the timed region",
                    "launches": launches, "avg_launch_us": round(1e3 * ms_total / launches, 2),
                    "flops_per_launch": round(flops_total / launches),
                    "timed_steps": sampled,
                    "share_of_step": round(ms_total * 1e-3 / n_s / (dt / args.steps), 4)}
            pmc = os.path.join(ROOT, "profiles", "conv_traffic.json")
            if os.path.exists(pmc) and args.workload == "nuscenes_34k_w5" and args.cr == 1.0 and not args.fwd_only:
                try:
                    roof["traffic"] = json.load(open(pmc)).get("hbm_bytes_per_launch")
                except Exception:
                    pass
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(w, args.cr, args.cpu_scans)
        out = {
            "metric": "points/sec per sparse-conv 3D fwd+ST-voxel TTA step" if not args.fwd_only else "points/sec, 3D fwd only",
            "value": n_points * world * args.steps / dt, "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scans_per_gpu": w["B"], "points_per_scan": w["n"], "classes": w["C"],
                       "st_window": w["W"], "st_voxel_m": w["vs"], "max_range_m": w["R"], "spvcnn_cr": args.cr,
                       "voxels_per_gpu": batch["n_voxels"], "parallelism": f"dp{world} (sequences sharded)"},
            "host_issue_ms_per_step": round(1e3 * dt_host / args.steps, 3),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
