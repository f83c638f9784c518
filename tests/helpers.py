"""Shared loaders for the golden fixtures under tests/golden (see make_hr_golden.py for how they were made)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HR_CASES = ["tiny", "twoseq", "w5_360", "a2d2", "b1_32k", "nusc_cp05", "interleaved"]


def load_hr_case(name):
    z = np.load(os.path.join(GOLDEN, f"hr_{name}.npz"))
    n, B, C, W, vs, R, conf = z["cfg"]
    B, C, W = int(B), int(C), int(W)
    pts = [torch.from_numpy(z[f"in_points_{b}"]) for b in range(B)]
    fov = [torch.from_numpy(z[f"in_fov_{b}"]) for b in range(B)]
    poses = [torch.from_numpy(z[f"in_pose_{b}"]) for b in range(B)]
    common = dict(pose=poses, scene_name=[str(s) for s in z["in_scene"]], scan_ID=[int(i) for i in z["in_scan_id"]],
                  img_indices=[torch.zeros(f.shape[0], 2) for f in fov])
    pts_2d = [p[f] for p, f in zip(pts, fov)]

    def batch(kind, logit):
        return dict(seg_logit=logit, ori_points=pts if kind == "3D" else pts_2d, **common)

    cfg = dict(temp_wd_size=W, max_range=float(R) if R != int(R) else int(R), voxel_size=float(vs),
               conf_perc=float(conf), B=B, C=C)
    return z, cfg, batch


def hr_roles(z, batch, tag, requires_grad=False):
    """tag '23': anchor = student 2D, hist = teacher 3D; tag '32': anchor = student 3D, hist = teacher 2D
    (latte/tta/tools/latte.py:199-202)."""
    if tag == "23":
        a = torch.from_numpy(z["in_logit_2d_student"]).clone().requires_grad_(requires_grad)
        h = torch.from_numpy(z["in_logit_3d_teacher"])
        return batch("2D", a), "2D", batch("3D", h), "3D", a
    a = torch.from_numpy(z["in_logit_3d_student"]).clone().requires_grad_(requires_grad)
    h = torch.from_numpy(z["in_logit_2d_teacher"])
    return batch("3D", a), "3D", batch("2D", h), "2D", a
