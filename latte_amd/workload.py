"""Synthetic TTA batches for the BASELINE.json configs: raw scans from latte_amd.synth are pushed through
the dataloader's voxelization contract ON THE GPU (quantize_scan_gpu: round(p*20)-min, unique 5 cm voxels, first-point
features), collated like latte/data/collate.py:403-421, and completed with the per-scan metadata HR_Block needs."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import synth
from .quantize import quantize_scan_gpu
from .tensor import SparseTensor

__all__ = ["build_batch", "make_raw", "batch_from_raw", "hr_cfg_of", "build_conv_1m"]


def hr_cfg_of(w: Dict) -> Dict:
    """HR_BLOCK knobs of a workload (latte/config/xmuda.py:293-298); conf_perc is the dataset's own value"""
    return dict(temp_wd_size=w["W"], max_range=w["R"], voxel_size=w["vs"], conf_perc=w.get("conf", 0.9))


def make_raw(w: Dict, scan_offset: int = 0):
    """The step's raw input as a loader would hold it: per scan (points [n,3] fp32, intensity [n] fp32) in PINNED host
    memory (16 B per point), plus the per-scan metadata HR_Block needs."""
    B, n, seq_len = w["B"], w["n"], w["seq_len"]
    scans, poses, scenes, ids = [], [], [], []
    for b in range(B):
        gid = scan_offset + b
        pts_np, inten_np = synth.make_scan(n, gid, beams=w["beams"], full_360=w["full_360"])
        pts, inten = torch.from_numpy(pts_np), torch.from_numpy(inten_np)
        if torch.cuda.is_available():
            pts, inten = pts.pin_memory(), inten.pin_memory()
        scans.append((pts, inten))
        poses.append(torch.from_numpy(synth.make_pose(gid % seq_len)))
        scenes.append(f"scene-{1000 + gid // seq_len:04d}")
        ids.append(gid % seq_len)
    return {"scans": scans, "pose": poses, "scene_name": scenes, "scan_ID": ids}


def batch_from_raw(raw: Dict, w: Dict, device="cuda", logits_2d=None, seed: int = 0) -> Dict:
    """raw host scans -> the batch dict of one TTA iteration, on the CURRENT stream: H2D of the points (non-blocking from
    pinned memory), the dataloader's voxelization contract on the GPU (quantize_scan_gpu: round(p*20)-min, unique 5 cm
    voxels, first-point features -- latte/data/semantic_kitti/semantic_kitti_dataloader.py:542-577), sparse_collate
    (latte/data/collate.py:403-421), camera-FOV maps.  logits_2d: (student, teacher) 2-D logits already on the device
    (the 2-D network's output in the reference; synthetic here), made from `seed` when None."""
    C = w["C"]
    coords, feats, indices, inverse, ori, fov = [], [], [], [], [], []
    for b, (pts_h, inten_h) in enumerate(raw["scans"]):
        pts = pts_h.to(device, non_blocking=True)
        inten = inten_h.to(device, non_blocking=True)
        c, f, inds, inv = quantize_scan_gpu(pts, inten, scale=20)
        coords.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int32, device=device)], 1))
        feats.append(f)
        indices.append(inds)
        inverse.append(inv)
        ori.append(pts)
        fov.append((pts[:, 0] > 0) & (pts[:, 1].abs() < 0.8 * pts[:, 0]))  # camera frustum stand-in (bool map)
    lidar = SparseTensor(torch.cat(feats, 0), torch.cat(coords, 0), 1)
    n_fov = [int(m.sum()) for m in fov]
    if logits_2d is None:
        g = torch.Generator(device="cpu").manual_seed(seed)
        l2s = (3 * torch.randn(sum(n_fov), C, generator=g)).to(device).requires_grad_(True)
        l2t = (3 * torch.randn(sum(n_fov), C, generator=g)).to(device)
    else:
        l2s, l2t = logits_2d
        assert l2s.shape[0] == sum(n_fov)
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream())   # every input above was produced on this stream (LatteStep._ready_event)
    return {
        "ready_event": ready,
        "lidar": lidar, "indices": indices, "inverse_map": inverse, "ori_points": ori, "pc2img_idx": fov,
        "img_indices": [torch.empty(k, 2) for k in n_fov], "pose": raw["pose"], "scene_name": raw["scene_name"],
        "scan_ID": raw["scan_ID"], "logit_2d_student": l2s, "logit_2d_teacher": l2t,
        "n_points": sum(int(p.shape[0]) for p, _ in raw["scans"]), "n_voxels": int(lidar.C.shape[0]),
    }


def build_batch(w: Dict, device="cuda", scan_offset: int = 0, seed: int = 0) -> Dict:
    """w: a latte_amd.synth.WORKLOADS entry.  scan_offset shifts the scan/sequence ids (multi-GPU sharding: every rank
    builds its own sequences)."""
    return batch_from_raw(make_raw(w, scan_offset), w, device=device, seed=seed)


def build_conv_1m(device="cuda", min_voxels: int = 1_000_000, points_per_scan: int = 131072):
    """BASELINE configs[4]: >= 1 M voxels from tiling config-3 scans: as many 131 072-point KITTI-like
    scans as it takes, one batch index each, voxelized like the network's x0 (hash-ordered rows).  -> SparseTensor x0
    (zero features; the micro-bench and its test bring their own)."""
    from .point_voxel import initial_voxelize
    from .tensor import PointTensor
    coords, total, b = [], 0, 0
    while total < min_voxels:
        p, i = synth.make_scan(points_per_scan, b, beams=64, full_360=False)
        c, _, _, _ = quantize_scan_gpu(torch.from_numpy(p).to(device), torch.from_numpy(i).to(device))
        coords.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int32, device=device)], 1))
        total += c.shape[0]
        b += 1
    C4 = torch.cat(coords, 0)
    z = PointTensor(torch.zeros(C4.shape[0], 4, device=device), C4.float())
    return initial_voxelize(z, 0.05, 0.05)
