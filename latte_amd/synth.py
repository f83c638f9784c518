"""Deterministic LiDAR-like synthetic scans (numpy only; no GPU, no reference import).

Restates the generator SURVEY.md §8d prescribes for every BASELINE.json config:
64 (or 32) beams, elevation linspace(-24.8deg, +2deg), N/beams azimuth columns over
+-45deg (front-only, KITTI-like) or the full 360deg (nuScenes-like); range = min(ground
plane z=-1.73 m, side walls at y=+-(6+2u) m, 80 m) + N(0, 0.02 m); intensity U(0,1);
pose_j = yaw 0.01*j rad, x translation 0.9*j m (float64 4x4).

The dataloader contract the scans are pushed through (reference
latte/data/utils/augmentation_3d.py:48-56 and
latte/data/semantic_kitti/semantic_kitti_dataloader.py:542-577) is restated in
`quantize_scan`: coords = round(points*scale) - min, then sparse_quantize.
"""
from __future__ import annotations

import numpy as np

__all__ = ["make_scan", "make_pose", "WORKLOADS", "workload"]


def make_scan(n_points: int, scan_idx: int, beams: int = 64, full_360: bool = False,
              seed_base: int = 1000):
    """Return (points [N,3] float32 metres, intensity [N] float32)."""
    rng = np.random.default_rng(seed_base + scan_idx)
    cols = max(n_points // beams, 1)
    n = cols * beams
    elev = np.deg2rad(np.linspace(-24.8, 2.0, beams))
    if full_360:
        az = np.deg2rad(np.linspace(-180.0, 180.0, cols, endpoint=False))
    else:
        az = np.deg2rad(np.linspace(-45.0, 45.0, cols))
    el, a = np.meshgrid(elev, az, indexing="ij")
    el = el.reshape(-1)
    a = a.reshape(-1)
    dx = np.cos(el) * np.cos(a)
    dy = np.cos(el) * np.sin(a)
    dz = np.sin(el)
    u = rng.random()
    wall = 6.0 + 2.0 * u
    with np.errstate(divide="ignore", invalid="ignore"):
        t_ground = np.where(dz < -1e-6, -1.73 / dz, np.inf)
        t_wall = np.where(np.abs(dy) > 1e-6, wall / np.abs(dy), np.inf)
    r = np.minimum(np.minimum(t_ground, t_wall), 80.0)
    r = r + rng.normal(0.0, 0.02, size=n)
    pts = np.stack([r * dx, r * dy, r * dz], axis=1).astype(np.float32)
    inten = rng.random(n).astype(np.float32)
    if n < n_points:  # pad by repeating jittered points so the count is exact
        extra = n_points - n
        sel = rng.integers(0, n, size=extra)
        pts = np.concatenate([pts, pts[sel] + rng.normal(0, 0.01, (extra, 3)).astype(np.float32)], 0)
        inten = np.concatenate([inten, inten[sel]], 0)
    return np.ascontiguousarray(pts), np.ascontiguousarray(inten)


def make_pose(j: int) -> np.ndarray:
    """pose_j: yaw 0.01*j rad about z, translation 0.9*j m along x (float64 4x4)."""
    th = 0.01 * j
    T = np.eye(4, dtype=np.float64)
    T[0, 0], T[0, 1] = np.cos(th), -np.sin(th)
    T[1, 0], T[1, 1] = np.sin(th), np.cos(th)
    T[0, 3] = 0.9 * j
    return T


# BASELINE.json configs restated as synthetic workloads.
WORKLOADS = {
    # name: n_points, n_scans(B), classes, temp_wd_size, voxel_size, max_range, full_360, beams, frames/sequence,
    # conf = HR_BLOCK.conf_perc of the matching configs/*/latte.yaml:29 (0.9 for the two KITTI targets, 0.5 for nuScenes)
    "synthia_kitti_32k_b1": dict(n=32768, B=1, C=9, W=3, vs=0.2, R=40, full_360=False, beams=64, seq_len=1, conf=0.9),
    "synthia_kitti_32k": dict(n=32768, B=6, C=9, W=3, vs=0.2, R=40, full_360=False, beams=64, seq_len=6, conf=0.9),
    "nuscenes_34k_w5": dict(n=34816, B=6, C=16, W=5, vs=0.2, R=50, full_360=True, beams=32, seq_len=6, conf=0.5),
    "a2d2_kitti_128k": dict(n=131072, B=6, C=10, W=3, vs=0.1, R=40, full_360=False, beams=64, seq_len=6, conf=0.9),
    "multiseq_64": dict(n=34816, B=64, C=16, W=5, vs=0.2, R=50, full_360=True, beams=32, seq_len=8, conf=0.5),
    "tiny": dict(n=2048, B=3, C=5, W=3, vs=0.4, R=30, full_360=False, beams=16, seq_len=3, conf=0.9),
}


def workload(name: str) -> dict:
    return dict(WORKLOADS[name])
