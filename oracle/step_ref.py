"""ORACLE (test infrastructure, never the product path) -- CPU restatement of ONE Latte TTA iteration, 3-D side.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows latte/tta/tools/latte.py:156-313 with the 2-D SegFormer replaced by given 2-D logits (the 2-D branch is out
of scope): teacher 3-D forward (:159), inverse_to_all (latte/data/collate.py:34-55), camera-FOV
subset (latte/data/utils/all_to_partial.py:4-37), student forward (:176), HR_Block x2 (:199-202), pseudo-label
fusion (:207-238), CR_Block x2 (:276-291), cross-entropy (:295-296), backward (:313).  The backbone is
oracle/spvcnn_ref.py (PARITY UNPINNED w.r.t. torchsparse, see there), the ST-voxel part is oracle/hr_ref.py (pinned by
tests/golden/hr_*.npz).  Plain numpy / torch-CPU fp32.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from . import hr_ref, ts_ref
from .spvcnn_ref import SPVCNNRef


def quantize_batch(scans: List, scale: int = 20):
    """scans: list of (points [N,3] f32, intensity [N] f32) -> (coords [V,4] int32, feats [V,4] f32, inverse maps,
    voxels per scan) through the dataloader contract (oracle/ts_ref.quantize_scan + sparse_collate)."""
    cs, fs, invs = [], [], []
    for pts, inten in scans:
        c, f, _inds, inv = ts_ref.quantize_scan(pts, inten, scale=scale)
        cs.append(c)
        fs.append(f)
        invs.append(inv)
    coords, feats = ts_ref.sparse_collate(cs, fs)
    return coords, feats, invs, [c.shape[0] for c in cs]


def inverse_to_all(logit: torch.Tensor, inverse_maps, sizes) -> torch.Tensor:
    out, off = [], 0
    for inv, n in zip(inverse_maps, sizes):
        out.append(logit[off:off + n][torch.from_numpy(np.asarray(inv)).long()])
        off += n
    return torch.cat(out, 0)


def pseudo_label_fuse_ref(logit_2d_teacher: torch.Tensor, logit_3d_teacher: torch.Tensor, mask: torch.Tensor,
                          st2: torch.Tensor, st3: torch.Tensor, cp):
    """latte/tta/tools/latte.py:207-233: point entropies (ST-voxel entropy substituted where valid) and the fused
    pseudo-label probabilities.  mask: bool [N] (in-FOV points); -> (pc2 [Nf], pc3 [N], ps_prob [N, C])."""
    probs_2d = F.softmax(logit_2d_teacher, 1)
    all_probs_3d = F.softmax(logit_3d_teacher, 1)
    probs_3d = F.softmax(logit_3d_teacher[mask], 1)
    v3, v2 = st3 != 0, st2 != 0
    if cp is not None and cp >= 0:
        v3 = torch.logical_and(st3 <= torch.quantile(st3, cp), v3)
        v2 = torch.logical_and(st2 <= torch.quantile(st2, cp), v2)
    pc2 = hr_ref.prob_2_entropy(probs_2d).sum(1)
    pc3 = hr_ref.prob_2_entropy(all_probs_3d).sum(1)
    pc2[v2] = st2[v2]
    pc3[v3] = st3[v3]
    rv2, rv3 = torch.exp(-pc2), torch.exp(-pc3)
    w2 = rv2 / (rv2 + rv3[mask])
    w3 = rv3[mask] / (rv2 + rv3[mask])
    ps_prob = all_probs_3d
    ps_prob[mask, :] = w2.unsqueeze(1) * probs_2d + w3.unsqueeze(1) * probs_3d
    return pc2, pc3, ps_prob


def tta_step_ref(student_sd: Dict, teacher_sd: Dict, coords: np.ndarray, feats: torch.Tensor, inverse_maps, sizes,
                 ori_points: List[torch.Tensor], fov_masks: List[torch.Tensor], poses, scene_names, scan_ids,
                 logit_2d_student: torch.Tensor, logit_2d_teacher: torch.Tensor, hr_cfg: Dict,
                 lambda_hr_23: float = 0.3, lambda_hr_32: float = 0.3, grad_names=None, backward: bool = True,
                 student_relu_masks=None, ps_label_override=None, conf_mask_override=None):
    """-> (losses dict, {param name: grad} for `grad_names`, grad of logit_2d_student, aux).

    The three *_override arguments fix the step's DISCRETE decisions from outside (tests/test_gpu_step.py feeds the GPU
    run's): the student's ReLU activity pattern (see SPVCNNRef.relu_masks), the pseudo labels after refine, and the two
    tr_conf_masks (hr_23, hr_32).  An activation / label / entropy that is tied to fp32 rounding may fall on either side
    in two correct implementations; with the decisions shared, what is left to compare is arithmetic."""
    teacher = SPVCNNRef(teacher_sd)
    student = SPVCNNRef(student_sd)
    student.relu_masks = student_relu_masks
    params = student.parameters_requiring_grad(grad_names) if grad_names else []
    l2s = logit_2d_student.detach().clone().requires_grad_(True)
    extra = dict(pose=poses, scene_name=scene_names, scan_ID=scan_ids,
                 img_indices=[torch.empty(int(m.sum()), 2) for m in fov_masks])
    with torch.no_grad():
        ema3d = inverse_to_all(teacher(coords, feats)["seg_logit"], inverse_maps, sizes)
        mask = torch.cat(list(fov_masks), 0)
    s3d = inverse_to_all(student(coords, feats)["seg_logit"], inverse_maps, sizes)
    ori_2d = [p[m] for p, m in zip(ori_points, fov_masks)]
    a2 = dict(seg_logit=l2s, ori_points=ori_2d, **extra)
    h2 = dict(seg_logit=logit_2d_teacher, ori_points=ori_2d, **extra)
    a3 = dict(seg_logit=s3d, ori_points=ori_points, **extra)
    h3 = dict(seg_logit=ema3d, ori_points=ori_points, **extra)
    kw = dict(temp_wd_size=hr_cfg["temp_wd_size"], max_range=hr_cfg["max_range"], voxel_size=hr_cfg["voxel_size"],
              conf_perc=hr_cfg["conf_perc"])
    hr23 = hr_ref.hr_forward(a2, "2D", h3, "3D", **kw)
    hr32 = hr_ref.hr_forward(a3, "3D", h2, "2D", **kw)
    with torch.no_grad():
        pc2, pc3, ps_prob = pseudo_label_fuse_ref(logit_2d_teacher, ema3d, mask, hr32["hs_pc_ety"], hr23["hs_pc_ety"],
                                                  hr_cfg["conf_perc"])
        ps_label = hr_ref.refine_pseudo_labels(ps_prob.amax(1), ps_prob.argmax(1))
    ps_label_own = ps_label
    if ps_label_override is not None:
        ps_label = ps_label_override.to(ps_label.dtype)
    own_masks = (hr23["tr_conf_mask"], hr32["tr_conf_mask"])
    if conf_mask_override is not None:
        hr23["tr_conf_mask"], hr32["tr_conf_mask"] = conf_mask_override
    hr23.update({"ac_pc_ety": pc2, "tr_pc_ety": pc3})
    l23 = hr_ref.cr_forward(hr23)
    hr32.update({"ac_pc_ety": pc3, "tr_pc_ety": pc2})
    l32 = hr_ref.cr_forward(hr32)
    pl2 = F.cross_entropy(l2s, ps_label[mask].long())
    pl3 = F.cross_entropy(s3d, ps_label.long())
    total = lambda_hr_32 * l32 + pl3 + lambda_hr_23 * l23 + pl2
    grads = {}
    if backward:
        total.backward()
        grads = {n: p.grad for n, p in zip(grad_names or [], params)}
    losses = {"hr_23": float(l23.detach()), "hr_32": float(l32.detach()), "pl_2d": float(pl2.detach()), "pl_3d": float(pl3.detach())}
    return losses, grads, (l2s.grad if backward else None), {"ps_label": ps_label_own, "hr23": hr23, "hr32": hr32,
                                                                   "conf_masks": own_masks, "student": student}
