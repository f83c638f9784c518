#!/usr/bin/env python3
"""Headline benchmark: points/s of one Latte TTA step (3-D side) on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = teacher SPVCNN forward + student SPVCNN forward + 2x HR_Block (ST-voxel aggregation) + pseudo-label fusion +
2x CR_Block + cross-entropy + backward to the BatchNorm parameters + (N>1: one flat RCCL all-reduce of those grads) +
Adam + EMA teacher update -- latte/tta/tools/latte.py:156-323 without the 2-D SegFormer.  Inputs are
resident in HBM before the timed region; index structures are rebuilt from the coordinates in every step.

Workloads (latte_amd/synth.py restates BASELINE.json's configs):
  N = 1   a2d2_kitti_128k  BASELINE configs[2], the largest single-GPU config: 6 x 131 072-point scans, 5 cm input voxels,
          ST voxels 0.1 m / R 40 m / 3-frame window, 10 classes, SPVCNN cr = 1.0, fp32.  configs[0] / configs[1]
          (synthia_kitti_32k, nuscenes_34k_w5) are measured right after it and reported in `other_workloads`.
  N > 1   the SAME workload per GPU (weak scaling, what `value` and the driver's efficiency are computed on): every rank
          owns its own sequence of 6 a2d2 scans -- independent sequences shard across the GPUs with no data-path
          collective; the only exchange is one flat RCCL all-reduce of the 13 248 BatchNorm grads per step.
          Then, in the extra object `strong_scaling_multiseq_64`, BASELINE configs[3]: 64 scans per step = 8 sequences x 8
          frames of 34 816 points, whole sequences dealt to the ranks (4/2/1 per GPU at N = 2/4/8), total work fixed, with
          rank 0's one-GPU time of the same 64-scan step measured in the same run.
  --workload conv_1m       BASELINE configs[4]: 3^3 conv over >= 1 M voxels, C = 16..256, plus one U-Net forward over them.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      the dominant kernel family (fp32-MFMA sparse conv): algorithmic flops / HIP-event time of its launches in a
                sampled timed step (the library's conv launch timer: events on the stream the kernel is launched on);
                `hbm_classes`: achieved GB/s of the HBM-bound kernel classes (hash/unique, kernel-map query, voxelize,
                devoxelize, BatchNorm, segmented reduction, join): algorithmic bytes of SURVEY 8d / HIP-event time, each
                kernel run on the tensors of this workload
  cpu_baseline  the CPU oracle (oracle/step_ref.py, a port: the reference's torchsparse backend is not installable)
                timed on this box's host cores on a bounded sample of the same workload, with OMP_NUM_THREADS = 4 (the
                value the reference pins, latte/tta/tools/latte.py:7) and with all cores the box gives
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
PEAK_HBM_GBS = 8000.0         # same guide: HBM3E 8 TB/s spec
METRIC = "points/sec per sparse-conv 3D fwd+ST-voxel TTA step"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, help="default: a2d2_kitti_128k (N > 1: per GPU, weak scaling; the strong-scaling "
                    "multiseq_64 run follows as an extra object); multiseq_64 = 64 scans dealt to the ranks by sequence; "
                    "conv_1m = the sparse-conv micro-benchmark")
    ap.add_argument("--cr", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scans", type=int, default=None, help="scans in the CPU-oracle sample (default: sized to ~20 s)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the configs[0]/[1] lines of other_workloads")
    ap.add_argument("--no-hbm-classes", action="store_true", help="skip the per-class HBM GB/s measurements")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not time the conv launches with HIP events")
    ap.add_argument("--all-kernel-events", action="store_true", help="time the conv launches of every timed step")
    ap.add_argument("--no-side-stream", action="store_true", help="run the input-only, sync-heavy work (index structures, "
                    "HR geometry) on the main stream too")
    ap.add_argument("--no-prefetch", action="store_true", help="build the index structures of a batch inside its own step "
                    "instead of one step ahead")
    ap.add_argument("--no-from-raw", action="store_true", help="skip the extra measurement whose timed region starts at raw "
                    "host points (H2D + input voxelization + collate on the side stream, one step ahead)")
    ap.add_argument("--dump-conv", default=None, help="write the per-launch conv table of one step (JSON) here")
    ap.add_argument("--fwd-only", action="store_true", help="time teacher forward only")
    ap.add_argument("--conv-precision", default="f32", choices=["f32", "bf16x3"], help="arithmetic of the sparse convs: fp32 "
                    "MFMA (default, the headline) or the opt-in bf16x3 split-precision kernel (lt_conv_set_precision)")
    ap.add_argument("--no-bf16x3-line", action="store_true", help="skip the extra bf16x3 measurement of the default run")
    ap.add_argument("--no-strong-line", action="store_true", help="N > 1: skip the strong-scaling run of multiseq_64 (configs[3])")
    ap.add_argument("--no-scaling-base", action="store_true", help="N > 1: skip rank 0's one-GPU run of the whole 64-scan step")
    ap.add_argument("--emulate-shard", type=int, default=0, help="diagnostic, one GPU: time the step of rank 0's shard of "
                    "multiseq_64 split over this many ranks (what one rank of an N-GPU run computes; no collective)")
    ap.add_argument("--sync-bn", action="store_true", help="N > 1: cross-rank BatchNorm statistics (exact parity with the "
                    "unsplit batch: ~150 small all-reduces per step) instead of per-GPU statistics")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_leg(w, cr, n_scans, threads, seed=1):
    """Oracle TTA step (quantize, teacher fwd, student fwd, 2x HR, 2x CR, CE, backward) on `n_scans` scans of the workload
    with `threads` torch-CPU threads."""
    import torch
    from latte_amd import synth
    from latte_amd.spvcnn import Net3DSeg
    from latte_amd.workload import hr_cfg_of
    from oracle import step_ref

    torch.set_num_threads(int(threads))
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
    coords, feats, invs, sizes = step_ref.quantize_batch(scans)   # outside the timer, like the GPU step's `value` (inputs
    t0 = time.perf_counter()                                      # already voxelized; `from_raw` times it on the GPU side)
    step_ref.tta_step_ref(sd, sd, coords, torch.from_numpy(feats), invs, sizes, ori, fov,
                          [synth.make_pose(j) for j in range(n_scans)], ["scene-1000"] * n_scans, list(range(n_scans)),
                          l2s, l2t, hr_cfg_of(w), grad_names=names)
    dt = time.perf_counter() - t0
    return {"value": n_scans * w["n"] / dt, "unit": "points/s", "cores": int(threads), "seconds": round(dt, 2),
            "scans": n_scans, "points": n_scans * w["n"]}


def cpu_baseline(w, cr, n_scans=None):
    """Two legs on the same bounded sample: all cores the box gives this process, and the reference's OMP_NUM_THREADS=4."""
    import torch
    all_cores = int(os.environ.get("LATTE_BENCH_CPU_THREADS", 0)) or int(torch.get_num_threads())
    if n_scans is None:
        # the step's whole batch when that is <= ~1 M points (measured on the GPU box: ~40 k points/s on 16 threads, ~34 k on
        # 4 -> ~20 s + ~24 s for the 6 x 131 072-point batch), else as many whole scans as fit 1 M points
        n_scans = max(1, min(w["B"], 1_000_000 // w["n"]))
    full = _cpu_leg(w, cr, n_scans, all_cores)
    omp4 = _cpu_leg(w, cr, n_scans, min(4, all_cores))
    torch.set_num_threads(all_cores)
    sample = (f"one oracle TTA step (oracle/step_ref.py: teacher fwd, student fwd, 2x HR, 2x CR, CE, backward; inputs already voxelized) on "
              f"{n_scans} scan(s) x {w['n']} points of the same workload (the step's batch is {w['B']} scans), torch-CPU "
              f"fp32: {full['seconds']} s with {all_cores} threads, {omp4['seconds']} s with {omp4['cores']} threads "
              f"(the reference pins OMP_NUM_THREADS=4); the reference's own backend (torchsparse v1.4.0 CUDA) cannot run here")
    return {"value": full["value"], "unit": "points/s", "cores": all_cores, "kind": "port", "sample": sample,
            "cpu_model": _cpu_model(),
            "omp4": {"value": omp4["value"], "unit": "points/s", "cores": omp4["cores"], "seconds": omp4["seconds"]},
            "seconds": full["seconds"]}


# ------------------------------------------------------------------------------------------------ HBM-bound kernel classes
def _event_us(fn, reps=10, rounds=3):
    """best-of-`rounds` average GPU time of fn() in us, HIP events on torch's current stream = the stream every lt_* kernel
    is launched on (ops._stream)."""
    import torch
    fn()
    fn()
    best = float("inf")
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


def measured_peaks(dev):
    """The two roofs re-measured on this box: a 1 GiB device copy (read + write bytes / time) and a dense
    fp32 GEMM through torch (hipBLASLt), HIP events.  Reported beside the spec peaks the fractions are priced against."""
    import torch
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    us = _event_us(lambda: b.copy_(a), reps=5)
    copy = 2.0 * 4 * n / us / 1e3
    del a, b
    m = 8192
    x = torch.randn(m, m, device=dev)
    y = torch.randn(m, m, device=dev)
    prev = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    us = _event_us(lambda: torch.mm(x, y), reps=5)
    torch.backends.cuda.matmul.allow_tf32 = prev
    return {"copy_GBs": round(copy, 1), "copy_what": "torch copy_ of 1 GiB fp32 (read + write bytes)",
            "sgemm_TFLOPs": round(2.0 * m ** 3 / us / 1e6, 1), "sgemm_what": "torch.mm 8192^3 fp32 (hipBLASLt)"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def hbm_classes(batch, w, hr):
    """Achieved HBM GB/s per kernel class: ALGORITHMIC bytes (SURVEY 8d formulas, restated in DESIGN.md 4) / HIP-event time
    of the library call on this workload's own tensors.  N = points the backbone sees (unique input voxels), V = voxels
    of x0 (= N: the input is already unique), M = pairs of the stride-1 3^3 map."""
    import torch
    from latte_amd import functional as LF
    from latte_amd import ops
    from latte_amd import sparse_modules as spnn
    from latte_amd.point_voxel import initial_voxelize, v2p_index
    from latte_amd.tensor import PointTensor

    lidar = batch["lidar"]
    dev = lidar.C.device
    z = PointTensor(lidar.F, lidar.C.float())
    x0 = initial_voxelize(z, 0.05, 0.05)
    N, V = int(lidar.C.shape[0]), int(x0.C.shape[0])
    out = {}

    def add(name, us, nbytes, what):
        # the fraction is of the 8 TB/s HBM spec, but each class is timed in a loop over ONE working set of 2..700 MB that
        # stays resident in the 256 MB Infinity Cache between repetitions: a cache-resident rate, not a cold HBM rate
        out[name] = {"us": round(us, 1), "bytes": int(nbytes), "GBs": round(nbytes / us / 1e3, 1),
                     "frac_of_8TBs_cache_resident": round(nbytes / us / 1e3 / PEAK_HBM_GBS, 4), "what": what}

    # hash + unique (initial_voxelize: FNV hash of every point, sorted unique hashes, point -> voxel inverse)
    ci = lidar.C.int().contiguous()

    def hash_unique():
        h = ops.sphash(ci)
        ops.unique_sorted(h, key_bits=60, want_first=False)
    add("hash_unique", _event_us(hash_unique), 16 * N + 8 * N + 20 * V,
        "lt_sphash + lt_unique_sorted over the batch's points: 16N + 8N + 20V")
    # kernel-map query, 3^3 at stride 1
    off = LF.get_kernel_offsets(3, 1, 1, device=dev)
    res = ops.kmap_build(x0.C, x0.C, off)
    M = int((res >= 0).sum())
    add("kmap_query", _event_us(lambda: ops.kmap_build(x0.C, x0.C, off, submanifold=True)), 16 * V + 16 * V + 8 * M,
        "lt_kmap_build_sub 3^3 stride 1 (table insert + 14 probes per voxel, the other 13 offsets mirrored): "
        "16Vin + 16Vout + 8M")
    # voxelize (scatter-mean point -> voxel as an atomic-free CSR gather), C = 32
    idx = z.additional_features["idx_query"][1]
    csr = z.additional_features["p2v_csr"][x0.s]
    f32 = torch.randn(N, 32, device=dev)
    add("voxelize", _event_us(lambda: ops.voxelize_csr_fwd(f32, csr, V)), 4 * N * 32 + 4 * N + 4 * V * 32,
        "lt_voxelize_csr_fwd C=32 (scatter-mean as sorted segments): 4NC + 4N + 4VC")
    # devoxelize (8-corner trilinear gather), C = 96
    iq, wt = v2p_index(x0, z)
    f96 = torch.randn(V, 96, device=dev)
    add("devoxelize", _event_us(lambda: ops.devoxelize_fwd(f96, iq, wt)), 64 * N + 4 * V * 96 + 4 * N * 96,
        "lt_devoxelize_fwd C=96 (8 gathered rows per point): 64N + 4VC + 4NC")
    # BatchNorm: statistics, apply(+ReLU), backward (reduce + apply) at [V, 96]
    gamma, beta = torch.ones(96, device=dev), torch.zeros(96, device=dev)
    mean, var = ops.bn_stats(f96)
    add("bn_stats", _event_us(lambda: ops.bn_stats(f96)), 4 * V * 96, "lt_bn_stats [V,96]: 4VC")
    y = torch.empty_like(f96)
    add("bn_apply_relu", _event_us(lambda: ops.bn_apply(f96, mean, var, gamma, beta, 1e-5, relu=True, out=y)), 8 * V * 96,
        "lt_bn_apply [V,96] with ReLU: 8VC")
    dy = torch.randn(V, 96, device=dev)
    add("bn_backward", _event_us(lambda: ops.bn_bwd(dy, f96, None, mean, var, gamma, 1e-5, True, beta=beta)), 24 * V * 96,
        "lt_bn_bwd_reduce + lt_bn_bwd_apply [V,96], ReLU mask rebuilt from x: 24VC")
    # ST-voxel stage: segmented mean over sorted voxel keys, softmax+entropy, sorted-set join
    extra = {k: batch[k] for k in ("pose", "scene_name", "scan_ID", "img_indices", "ori_points")}
    geo = hr.geometry(extra, "3D", extra, "3D")
    C = w["C"]
    P = geo["Ph"]
    logits = 3 * torch.randn(P, C, device=dev)
    add("softmax_entropy", _event_us(lambda: ops.softmax_entropy(logits)), 4 * P * C * 2 + 4 * P,
        "lt_softmax_entropy [P,C_cls]: 8PC + 4P")
    probs, _ = ops.softmax_entropy(logits)
    E, Vt = int(geo["h_selem"].numel()), int(geo["h_seg"].numel()) - 1
    add("segment_reduce", _event_us(lambda: ops.segment_reduce(probs, geo["h_row"], geo["h_seg"], "mean")),
        4 * E * C + 4 * E + 4 * Vt * C + 4 * Vt,
        "lt_segment_reduce mean of hist probabilities over ST voxels: 4E*C (gathered rows) + 4E (row index) + 4Vt*C + 4Vt")
    a_keys, h_keys = None, None
    # the join's inputs are the sorted unique keys of both sides: rebuild them the way geometry() does
    pts = torch.cat([p.reshape(-1, 3) for p in batch["ori_points"]], 0).float()
    sizes = [int(p.shape[0]) for p in batch["ori_points"]]
    import numpy as np
    st = torch.as_tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int64, device=dev)
    anc = torch.arange(len(sizes), dtype=torch.int32, device=dev)
    cb = ops.st_coord_bits(hr.max_range, hr.voxel_size)
    kb = 3 * cb + max(1, (len(sizes) - 1).bit_length())
    keys, _ = ops.st_keys(pts, st, st[:-1].contiguous(), anc, None, int(pts.shape[0]), hr.max_range, hr.voxel_size,
                          hr.min_offset, cb)
    _, _, a_uk, _ = ops.sort_segments(keys, kb)
    add("sort_segments", _event_us(lambda: ops.sort_segments(keys, kb)), 8 * P + 8 * P + 12 * int(a_uk.numel()),
        "lt_sort_segments of the anchor keys (radix sort + segment heads): 8P keys + 8P (sorted elem, inverse) + 12V")
    add("st_join", _event_us(lambda: ops.st_join(a_uk, a_uk)), 8 * 2 * int(a_uk.numel()),
        "lt_st_join of two sorted voxel-key sets: 8(Va + Vt)")
    return out


# ------------------------------------------------------------------------------------------------ conv micro-benchmark
def run_conv_1m(args):
    """BASELINE configs[4]."""
    import torch
    from latte_amd import _lib, ops
    from latte_amd import sparse_modules as spnn
    from latte_amd.spvcnn import SPVCNN
    from latte_amd.tensor import SparseTensor
    from latte_amd.workload import build_conv_1m
    _lib.load()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    x0 = build_conv_1m(device=dev)
    kmap, _ = spnn.build_kernel_map(x0, 3, 1, 1)
    res, plan = kmap.results, kmap.plan()
    V = int(res.shape[1])
    M = int((res >= 0).sum())
    rows = []
    for C in (16, 32, 64, 96, 128, 256):
        feats = torch.randn(V, C, device=dev)
        pw = ops.pack_weights(torch.randn(27, C, C, device=dev) * 0.05, 0)
        us = _event_us(lambda: ops.spconv_plan(feats, pw, plan, 27, V, C), reps=max(args.steps, 5))
        flops = 2.0 * M * C * C
        comp = 4.0 * (2 * V * C + 27 * C * C) + 8.0 * M
        intensity = flops / comp
        bound = "mfma" if intensity > PEAK_F32_MFMA_TFLOPS * 1e3 / PEAK_HBM_GBS else "hbm"
        rows.append({"C": C, "us": round(us, 1), "tflops": round(flops / us / 1e6, 2),
                     "frac_mfma": round(flops / us / 1e6 / PEAK_F32_MFMA_TFLOPS, 4),
                     "compulsory_GBs": round(comp / us / 1e3, 1), "frac_hbm": round(comp / us / 1e3 / PEAK_HBM_GBS, 4),
                     "flop_per_byte": round(intensity, 1), "bound": bound})
    # one U-Net forward over the same voxels (features: 4 random channels, like the network's input)
    torch.manual_seed(1)
    net = SPVCNN(cr=args.cr).to(dev).train()
    lidar = SparseTensor(torch.randn(V, 4, device=dev), x0.C, 1)

    def fwd():
        lidar._lt_cache = None
        with torch.no_grad():
            net(lidar)
    for _ in range(max(args.warmup, 1)):
        fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    best = max(rows, key=lambda r: r["frac_mfma"])
    out = {"metric": "voxels/sec, SPVCNN 3D U-Net forward over 1M voxels (BASELINE configs[4])", "value": V / dt,
           "unit": "voxels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "conv_1m", "voxels": V, "pairs_3x3x3": M, "pairs_per_voxel": round(M / V, 2),
                      "spvcnn_cr": args.cr},
           "roofline": {"bound": "mfma", "achieved": best["tflops"], "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": best["frac_mfma"], "traffic": None,
                        "kernel": f"k_spconv_sj, 3^3 conv Cin = Cout = {best['C']} over {V} voxels (best of the table)"},
           "conv_table": rows, "cpu_baseline": None}
    print(json.dumps(out), flush=True)


# ------------------------------------------------------------------------------------------------ the step benchmark
def time_steps(step, batch, args, world, dev, backend, fwd_only=False, steps=None, warmup=None, kernel_events=True):
    """-> dict(dt, dt_host, trace, sampled, pairs).  W untimed warm-up steps, then exactly K steps between
    barrier + synchronize on both sides, max over ranks."""
    import torch
    import torch.distributed as dist
    from latte_amd import ops
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup

    def one_step():
        if fwd_only:
            with torch.no_grad():
                return step.ema_model(batch)
        # the synthetic workload feeds the same scans every step: `next_batch=batch` makes the step issue the NEXT
        # iteration's input-only structures (rebuilt from the coordinates, not reused) while its own backward runs
        return step.step(batch, next_batch=None if args.no_prefetch else batch)

    for _ in range(warmup):
        one_step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # conv launches are timed (HIP events attached to the dispatches) in ONE of the timed steps, the second
    # (all of them with --all-kernel-events) -- an event pair costs ~10 us of launch path, 125 conv launches per step
    sampled = [i for i in range(steps) if args.all_kernel_events or i == 1 or steps < 2]
    if kernel_events:
        ops.conv_timer_begin(1 << 16)
        ops.conv_timer_arm(False)
    wait0 = step.host_wait_s
    t0 = time.perf_counter()
    for i in range(steps):
        if kernel_events and i in sampled:
            ops.conv_timer_arm(True)
            one_step()
            ops.conv_timer_arm(False)
        else:
            one_step()
    dt_host = time.perf_counter() - t0  # the host has issued every step (the GPU may still be working)
    dt_wait = step.host_wait_s - wait0  # ... of which blocked on the run-ahead bound (2 steps in flight): GPU-bound time
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    trace = ops.conv_timer_end() if kernel_events else None
    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    pairs = None
    if kernel_events:
        # pair counts M of every conv launch of a step, from one more (untimed) step on every rank: the launch sequence
        # and the kernel maps are the same in every step (same scans), only the BN parameters move
        ops.CONV_PAIRS = []
        one_step()
        torch.cuda.synchronize()
        pairs = [int(m.item()) for m in ops.CONV_PAIRS]
        ops.CONV_PAIRS = None
    step.close()
    return dict(dt=dt, dt_host=dt_host, dt_wait=dt_wait, trace=trace, sampled=sampled, pairs=pairs, steps=steps)


def conv_roofline(r, args):
    ms, meta = r["trace"]
    sampled, pairs = r["sampled"], r["pairs"]
    n_s = len(sampled)
    per_step = len(ms) // n_s
    assert len(pairs) == per_step and per_step * n_s == len(ms), (len(pairs), per_step, len(ms))
    if args.dump_conv:
        us = 1e3 * ms.reshape(n_s, per_step).mean(0)
        table = [dict(i=i, Cin=int(c[0]), Cout=int(c[1]), K=int(c[2]), Vout=int(c[3]), M=m, us=round(float(u), 1),
                      tflops=round(2.0 * m * int(c[0]) * int(c[1]) / float(u) / 1e6, 2))
                 for i, (m, c, u) in enumerate(zip(pairs, meta[:per_step], us))]
        json.dump(table, open(args.dump_conv, "w"))
    # The roofline family is the SPARSE convs (K = 8 or 27 offsets).  The one-call backbone pass also issues the network's
    # dense layers (1x1x1 shortcut convs, point-branch Linears: K = 1 over an identity plan) through the same kernel; those
    # are HBM-bound dense products (hipBLASLt GEMMs on the module path) and are reported beside the family, not inside it.
    import numpy as np
    ks = np.array([int(c[2]) for c in meta[:per_step]] * n_s)
    sparse = ks > 1
    fl = np.array([2.0 * m * int(c[0]) * int(c[1]) for m, c in zip(pairs, meta[:per_step])] * n_s)
    msd = ms.astype("float64")
    ms_total = float(msd[sparse].sum())
    launches = int(sparse.sum())
    flops_total = float(fl[sparse].sum())
    achieved = flops_total / (ms_total * 1e-3) / 1e12
    out = {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
           "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
           "kernel": "sparse-conv family (k_spconv_sj<SR,NTW,W> split-j fp32 MFMA; k_spconv_ls for the transposed 2^3 convs; "
                     "4-channel stem: k_spconv_stem): every sparse-conv launch (K = 8 | 27) of the sampled steps of the "
                     "timed region",
           "launches": launches, "avg_launch_us": round(1e3 * ms_total / launches, 2),
           "flops_per_launch": round(flops_total / launches),
           "timed_steps": sampled,
           "share_of_step": round(ms_total * 1e-3 / n_s / (r["dt"] / r["steps"]), 4)}
    if (~sparse).any():
        md = float(msd[~sparse].sum())
        out["dense_layers_through_the_conv_kernel"] = {
            "launches": int((~sparse).sum()), "avg_launch_us": round(1e3 * md / int((~sparse).sum()), 2),
            "tflops": round(float(fl[~sparse].sum()) / (md * 1e-3) / 1e12, 3),
            "share_of_step": round(md * 1e-3 / n_s / (r["dt"] / r["steps"]), 4),
            "what": "1x1x1 shortcut convs and point-branch Linears (K = 1, identity plan, statistics in the epilogue): "
                    "HBM-bound dense products, not part of the family above"}
    return out


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
    from latte_amd import _lib, synth
    from latte_amd.spvcnn import Net3DSeg
    from latte_amd.tta_step import LatteStep, shard_scan_offset, shard_sequences
    from latte_amd.workload import build_batch, hr_cfg_of

    if os.environ.get("LATTE_BENCH_TRACE"):  # where a stuck multi-rank run stands: Python stacks of every rank after 60 s
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LATTE_BENCH_TRACE"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.workload == "conv_1m":
        if world > 1:
            raise SystemExit("--workload conv_1m is a one-GPU micro-benchmark")
        return run_conv_1m(args)
    _lib.load()
    # LATTE_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share GPUs)
    backend = os.environ.get("LATTE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # stream creation order of a rank, the same in every run: main (the null stream), the side stream, then whatever the
    # process group creates (main + side + collective stream = 3 <= 4 hardware queues)
    from latte_amd.tta_step import ensure_side_stream
    ensure_side_stream(dev)
    pg = None
    # LATTE_BENCH_FORCE_PG=1 (one rank): a one-rank nccl group whose all-reduce is issued every step -- the stream and thread
    # topology of an N-rank run (collective stream, watchdog thread) rehearsed on a one-GPU box
    force_pg = world == 1 and bool(os.environ.get("LATTE_BENCH_FORCE_PG"))
    if force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": dev} if backend == "nccl" else {}))
        pg = dist.group.WORLD
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        pg = dist.group.WORLD

    from latte_amd import ops as _ops
    _ops.set_conv_precision(args.conv_precision)
    name = args.workload or "a2d2_kitti_128k"
    w = synth.workload(name)
    scaling = "weak"
    total_scans = w["B"] * world
    if name == "multiseq_64":
        # BASELINE configs[3]: the 64 scans of ONE step are dealt to the ranks by whole sequences (strong scaling)
        n_seq = w["B"] // w["seq_len"]
        if world > n_seq:
            raise SystemExit(f"multiseq_64 has {n_seq} sequences: at most {n_seq} ranks")
        s0, ns = shard_sequences(n_seq, world, rank)
        total_scans = w["B"]
        if args.emulate_shard > 1 and world == 1:
            s0, ns = shard_sequences(n_seq, args.emulate_shard, 0)
            total_scans = ns * w["seq_len"]
        w_rank = dict(w, B=ns * w["seq_len"])
        scan_offset = s0 * w["seq_len"]
        scaling = "strong"
    else:
        w_rank = w
        scan_offset = shard_scan_offset(rank, w["B"])     # weak scaling: every rank owns its own B scans
    torch.manual_seed(1)  # cfg.RNG_SEED (latte/common/config/base.py:173); same weights on every rank
    model = Net3DSeg(w["C"], True, "SPVCNN", {"cr": args.cr}).to(dev)
    step = LatteStep(model, w["C"], hr_cfg_of(w), process_group=pg, side_stream=not args.no_side_stream,
                     sync_bn=args.sync_bn, force_collective=force_pg)
    batch = build_batch(w_rank, device=dev, scan_offset=scan_offset, seed=rank)
    n_points_total = total_scans * w["n"]

    r = time_steps(step, batch, args, world, dev, backend, fwd_only=args.fwd_only, kernel_events=not args.no_kernel_events)
    dt = r["dt"]

    # N > 1, default workload: after the weak-scaling headline every rank takes its share of BASELINE configs[3] -- ONE step of
    # 64 scans = 8 sequences x 8 frames, whole sequences dealt to the ranks (strong scaling: total work fixed)
    strong = None
    sw = None
    if world > 1 and args.workload is None and not args.no_strong_line and not args.fwd_only:
        sw = synth.workload("multiseq_64")
        n_seq = sw["B"] // sw["seq_len"]
        if world <= n_seq:
            s0, ns = shard_sequences(n_seq, world, rank)
            torch.manual_seed(1)
            sm = Net3DSeg(sw["C"], True, "SPVCNN", {"cr": args.cr}).to(dev)
            sstep = LatteStep(sm, sw["C"], hr_cfg_of(sw), process_group=pg, side_stream=not args.no_side_stream)
            sb = build_batch(dict(sw, B=ns * sw["seq_len"]), device=dev, scan_offset=s0 * sw["seq_len"], seed=rank)
            sr = time_steps(sstep, sb, args, world, dev, backend, steps=max(5, args.steps // 2), warmup=2, kernel_events=False)
            strong = {"workload": "multiseq_64", "scaling": "strong", "scans_per_step": sw["B"], "scans_per_gpu": ns * sw["seq_len"],
                      "n_gpus": world, "value": sw["B"] * sw["n"] * sr["steps"] / sr["dt"], "unit": "points/s",
                      "ms_per_step": round(1e3 * sr["dt"] / sr["steps"], 3), "steps": sr["steps"]}
            del sm, sstep, sb

    if rank == 0:
        roof = None
        if not args.no_kernel_events:
            roof = conv_roofline(r, args)
            pmc = os.path.join(ROOT, "profiles", "conv_traffic.json")
            if os.path.exists(pmc) and args.cr == 1.0 and not args.fwd_only:
                try:
                    t = json.load(open(pmc))
                    if t.get("workload") == name:
                        roof["traffic"] = t.get("hbm_bytes_per_launch")
                        roof["traffic_source"] = "profiles/conv_traffic.json (rocprofv3 --pmc passes of this command on " \
                                                 "this workload; not re-measured in this run)"
                except Exception:
                    pass
            if world == 1 and not args.no_hbm_classes and not args.fwd_only:
                roof["hbm_classes"] = hbm_classes(batch, w, step.hr)
                roof["peaks_measured_on_this_box"] = measured_peaks(dev)
        others = None
        if world == 1 and args.workload is None and not args.no_other_workloads and not args.fwd_only:
            others = {}
            for oname in ("nuscenes_34k_w5", "synthia_kitti_32k", "synthia_kitti_32k_b1"):
                ow = synth.workload(oname)
                torch.manual_seed(1)
                om = Net3DSeg(ow["C"], True, "SPVCNN", {"cr": args.cr}).to(dev)
                ostep = LatteStep(om, ow["C"], hr_cfg_of(ow), side_stream=not args.no_side_stream)
                ob = build_batch(ow, device=dev)
                orr = time_steps(ostep, ob, args, 1, dev, backend, steps=max(5, args.steps // 2), warmup=2)
                oroof = conv_roofline(orr, argparse.Namespace(dump_conv=None))
                others[oname] = {"value": ob["n_points"] * orr["steps"] / orr["dt"], "unit": "points/s",
                                 "ms_per_step": round(1e3 * orr["dt"] / orr["steps"], 3), "steps": orr["steps"],
                                 "points_per_step": ob["n_points"], "voxels": ob["n_voxels"],
                                 "conv_tflops": oroof["achieved"], "conv_frac_mfma": oroof["frac"],
                                 "host_busy_ms_per_step": round(1e3 * (orr["dt_host"] - orr["dt_wait"]) / orr["steps"], 3)}
                if oname == "synthia_kitti_32k_b1" and not args.no_cpu_baseline:
                    # BASELINE configs[0] is THE CPU-reference configuration (one 32 768-point scan): the oracle's step on
                    # exactly that input, all threads and the reference's OMP_NUM_THREADS = 4
                    others[oname]["config"] = "BASELINE configs[0]: one synthetic 32k-point scan, B = 1 (the CPU-reference case)"
                    others[oname]["cpu_baseline"] = cpu_baseline(ow, args.cr, 1)
                del om, ostep, ob
        bf = None
        if (world == 1 and args.workload is None and args.conv_precision == "f32" and not args.no_bf16x3_line
                and not args.fwd_only):
            # the same step with the opt-in bf16x3 split-precision conv kernel (fp32-equivalent to ~1e-5; NOT the headline)
            _ops.set_conv_precision("bf16x3")
            torch.manual_seed(1)
            bm = Net3DSeg(w["C"], True, "SPVCNN", {"cr": args.cr}).to(dev)
            bstep = LatteStep(bm, w["C"], hr_cfg_of(w), side_stream=not args.no_side_stream)
            br = time_steps(bstep, batch, args, 1, dev, backend, steps=args.steps, warmup=2)
            broof = conv_roofline(br, argparse.Namespace(dump_conv=None))
            bf = {"value": n_points_total * br["steps"] / br["dt"], "unit": "points/s",
                  "ms_per_step": round(1e3 * br["dt"] / br["steps"], 3), "steps": br["steps"],
                  "conv_tflops_fp32_equivalent": broof["achieved"], "conv_avg_launch_us": broof["avg_launch_us"],
                  "note": "convs with >= 64 input channels on v_mfma_f32_16x16x32_bf16, operands split hi + lo in bf16, products "
                          "hi*hi + hi*lo + lo*hi accumulated in fp32 (~1e-5 relative to the fp32 kernel); opt-in, never the headline"}
            _ops.set_conv_precision("f32")
            del bm, bstep
        from_raw = None
        if world == 1 and not args.no_from_raw and not args.fwd_only and not args.no_side_stream and not args.no_prefetch:
            # the same step with the timed region starting at RAW host points: per iteration 6 x 131 072 x 16 B over PCIe from
            # pinned memory + round(p*20)-min + sparse_quantize + collate (semantic_kitti_dataloader.py:542-577,
            # tools/latte.py:139-148), done by the helper thread on the side stream one iteration ahead of its step
            from latte_amd.workload import batch_from_raw, make_raw
            raw = make_raw(w_rank, scan_offset)
            l2 = (batch["logit_2d_student"], batch["logit_2d_teacher"])
            factory = lambda: batch_from_raw(raw, w_rank, device=dev, logits_2d=l2)   # noqa: E731
            step.prefetch(factory)
            for _ in range(2):
                step.step(None, next_batch=factory)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step.step(None, next_batch=factory)
            torch.cuda.synchronize()
            dt_raw = time.perf_counter() - t0
            step.close()
            from_raw = {"value": n_points_total * args.steps / dt_raw, "unit": "points/s",
                        "ms_per_step": round(1e3 * dt_raw / args.steps, 3), "steps": args.steps,
                        "h2d_bytes_per_step": int(n_points_total * 16),
                        "what": "timed region starts at raw host points (pinned): H2D + input voxelization "
                                "(quantize_scan_gpu) + collate run on the side stream one iteration ahead, then the same step; "
                                "never the headline `value` (that one starts with the inputs resident in HBM)"}
        if strong is not None and not args.no_scaling_base:
            # the strong-scaling base in the SAME run: rank 0 alone takes the whole 64-scan step (the other ranks wait at
            # the barrier below), so the speed-up over one GPU on identical work is in the same JSON line
            torch.manual_seed(1)
            m1 = Net3DSeg(sw["C"], True, "SPVCNN", {"cr": args.cr}).to(dev)
            s1 = LatteStep(m1, sw["C"], hr_cfg_of(sw), side_stream=not args.no_side_stream)
            b1 = build_batch(sw, device=dev, scan_offset=0, seed=0)
            r1 = time_steps(s1, b1, args, 1, dev, backend, steps=max(4, args.steps // 2), warmup=2, kernel_events=False)
            strong["one_gpu_base"] = {"ms_per_step": round(1e3 * r1["dt"] / r1["steps"], 3),
                                      "value": sw["B"] * sw["n"] * r1["steps"] / r1["dt"], "unit": "points/s", "steps": r1["steps"]}
            strong["speedup_over_one_gpu"] = round(strong["one_gpu_base"]["ms_per_step"] / strong["ms_per_step"], 3)
            del m1, s1, b1
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(w, args.cr, args.cpu_scans)
        out = {
            "metric": METRIC if not args.fwd_only else "points/sec, 3D fwd only",
            "value": n_points_total * args.steps / dt, "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "scans_per_step": total_scans, "scans_per_gpu": w_rank["B"],
                       "points_per_scan": w["n"], "classes": w["C"], "st_window": w["W"], "st_voxel_m": w["vs"],
                       "max_range_m": w["R"], "spvcnn_cr": args.cr, "voxels_per_gpu": batch["n_voxels"],
                       "batchnorm_statistics": "cross-rank (exact)" if (args.sync_bn and world > 1) else "per GPU",
                       "parallelism": f"dp{world} (whole sequences per rank, no data-path collective"
                                      + (", one flat RCCL all-reduce of the BatchNorm grads)" if world > 1 else ")"),
                       },
            "host_issue_ms_per_step": round(1e3 * r["dt_host"] / args.steps, 3),
            # the main thread's own time: issue time minus the time step() sat blocked on the run-ahead bound (waiting for
            # the GPU to finish the iteration before the previous one)
            "host_busy_ms_per_step": round(1e3 * (r["dt_host"] - r["dt_wait"]) / args.steps, 3),
            "roofline": roof, "cpu_baseline": cpu,
        }
        if others is not None:
            out["other_workloads"] = others
        if bf is not None:
            out["bf16x3_mode"] = bf
        if from_raw is not None:
            out["from_raw"] = from_raw
        if strong is not None:
            out["strong_scaling_multiseq_64"] = strong
        out["config"]["conv_precision"] = args.conv_precision
        if force_pg:
            out["config"]["rehearsal"] = f"one-rank {backend} process group, gradient all-reduce issued every step"
        if args.emulate_shard > 1 and world == 1:
            out["metric"] = f"DIAGNOSTIC: one rank's shard of multiseq_64 over {args.emulate_shard} ranks, timed alone on one GPU"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()      # ranks > 0 wait here while rank 0 measures the one-GPU base
    if world > 1 or force_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
