"""ORACLE (test infrastructure, never the product path) -- CPU restatement of Latte's ST-voxel aggregation.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Restates, in plain numpy / torch-CPU:
  * HR_Block.forward            reference latte/models/hr_module.py:135-331
  * HR_Block.sparse_quantize    reference latte/models/hr_module.py:58-97   (incl. quirk Q1: returned coords are shifted)
  * CR_Block.forward            reference latte/models/hr_module.py:344-420 (quirks Q2, Q3, Q6)
  * prob_2_entropy              reference latte/models/losses.py:10-19
  * refine_ety                  reference latte/data/utils/refine_pseudo_labels.py:33-53
  * refine_pseudo_labels        reference latte/data/utils/refine_pseudo_labels.py:10-30
  * torch_scatter.scatter(mean) reference call sites hr_module.py:236,237,252 (published semantics; wheel absent)

PINNED: tests/test_oracle_golden.py checks every function here against tests/golden/hr_*.npz, which were
produced by running the reference's own hr_module.py in the build container (tests/golden/make_hr_golden.py).
The dense [Va,Vt] voxel comparison of the reference (:271) is replaced by a set intersection on ravel keys --
same result, O(V log V)."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def prob_2_entropy(prob: torch.Tensor) -> torch.Tensor:
    c = prob.shape[1]
    return -torch.mul(prob, torch.log2(prob + 1e-30)) / np.log2(c)


def scatter_mean(src: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    n = int(index.max()) + 1
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).index_add(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add(0, index, torch.ones_like(index, dtype=src.dtype)).clamp(min=1)
    return out / cnt.view((-1,) + (1,) * (src.dim() - 1))


def refine_ety(ety: torch.Tensor, pseudo_label: torch.Tensor, conf_perc: float) -> torch.Tensor:
    mask = torch.ones_like(ety, dtype=torch.bool)
    for cls in pseudo_label.unique():
        idx = torch.nonzero(pseudo_label == cls).squeeze(1)
        thresh = torch.quantile(ety[idx], conf_perc)
        mask[idx[ety[idx] > thresh]] = False
    return mask


def refine_pseudo_labels(probs: torch.Tensor, pseudo_label: torch.Tensor, ignore_label: int = -100) -> torch.Tensor:
    pseudo_label = pseudo_label.clone()
    for cls in pseudo_label.unique():
        idx = torch.nonzero(pseudo_label == cls).squeeze(1)
        thresh = min(probs[idx].median(), 0.9)
        pseudo_label[idx[probs[idx] < thresh]] = ignore_label
    return pseudo_label


def st_sparse_quantize(points: np.ndarray, voxel_size: float, min_offset: int):
    """-> (voxels [V,3] int32 SHIFTED by -min_offset (Q1), indices [V], inverse [N])."""
    c = np.floor(points / np.array([voxel_size] * 3)).astype(np.int32)
    c = c - np.int32(min_offset)  # the reference's ravel_hash does `x -= min_offset` in place on the caller's array
    x = c.astype(np.uint64)
    xmax = np.max(x, axis=0).astype(np.uint64) + np.uint64(1)
    h = np.zeros(x.shape[0], dtype=np.uint64)
    for k in range(2):
        h += x[:, k]
        h *= xmax[k + 1]
    h += x[:, 2]
    _, indices, inverse = np.unique(h, return_index=True, return_inverse=True)
    return c[indices], indices.astype(np.int64), inverse.reshape(-1).astype(np.int64)


def _scene_id(s: str) -> int:
    return int(s.split("-")[-1]) if "-" in s else int(s)


def _key3(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.int64)
    return (v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2]


def hr_forward(anchor_batch, anchor_type, hist_batch, hist_type, temp_wd_size, max_range, voxel_size, conf_perc):
    """HR_Block.forward restated.  Batches are dicts with seg_logit [sum N, C] (torch), pose (list of 4x4 float64),
    scene_name, scan_ID, ori_points (list of [N_b,3] float32 torch), img_indices (list; only .shape[0] is used)."""
    temp_bond = math.floor(temp_wd_size / 2)
    min_offset = -math.ceil(max_range / voxel_size) - 1
    ac_ref = "img_indices" if anchor_type.upper() == "2D" else "ori_points"
    hs_ref = "img_indices" if hist_type.upper() == "2D" else "ori_points"
    a_scene = [_scene_id(s) for s in anchor_batch["scene_name"]]
    h_scene = [_scene_id(s) for s in hist_batch["scene_name"]]
    a_id, h_id = list(anchor_batch["scan_ID"]), list(hist_batch["scan_ID"])
    n_hist = hist_batch["seg_logit"].shape[0]
    hs_pc_ety = torch.zeros(n_hist)
    hs_cnt = torch.zeros(n_hist, dtype=torch.int32)
    out = {k: [] for k in ("ac_logits", "ac_voxels", "tr_logits", "tr_voxels", "ac_p2v_indices", "tr_p2v_indices")}
    voxel_ety_all = []
    ac_left = 0
    for i, pose_i in enumerate(anchor_batch["pose"]):
        corr = [j for j in range(len(h_id)) if h_scene[j] == a_scene[i] and abs(h_id[j] - a_id[i]) <= temp_bond]
        ac_right = ac_left + anchor_batch[ac_ref][i].shape[0]
        init_left = sum(hist_batch[hs_ref][j].shape[0] for j in range(min(corr)))
        left = init_left
        tr_pts, tr_lg = [], []
        for j in corr:
            right = left + hist_batch[hs_ref][j].shape[0]
            p = hist_batch["ori_points"][j]
            ph = torch.cat([p, torch.ones((p.shape[0], 1))], 1)
            T = torch.linalg.inv(torch.as_tensor(pose_i)) @ torch.as_tensor(hist_batch["pose"][j])
            tr_pts.append((T @ ph.double().T).T[:, :3])
            tr_lg.append(hist_batch["seg_logit"][left:right])
            left = right
        right_end = left
        tr_pts, tr_lg = torch.cat(tr_pts, 0), torch.cat(tr_lg, 0)
        tr_mask = torch.linalg.norm(tr_pts, ord=2, dim=1) <= max_range
        tr_pts, tr_lg = tr_pts[tr_mask], tr_lg[tr_mask]
        ac_all = anchor_batch["ori_points"][i]
        ac_mask = torch.linalg.norm(ac_all, ord=2, dim=1) <= max_range
        ac_pts = ac_all[ac_mask]
        ac_lg = anchor_batch["seg_logit"][ac_left:ac_right][ac_mask]

        tr_vox, tr_ind, tr_inv = st_sparse_quantize(tr_pts.numpy(), voxel_size, min_offset)
        ac_vox, ac_ind, ac_inv = st_sparse_quantize(ac_pts.numpy(), voxel_size, min_offset)
        tr_inv_t, ac_inv_t = torch.from_numpy(tr_inv), torch.from_numpy(ac_inv)
        tr_prob = F.softmax(tr_lg, 1)
        tr_vlog = scatter_mean(tr_prob, tr_inv_t)
        ac_vlog = scatter_mean(F.softmax(ac_lg, 1), ac_inv_t)
        tr_vety = scatter_mean(prob_2_entropy(tr_prob).sum(1), tr_inv_t)
        sl = slice(init_left, right_end)
        seg = hs_pc_ety[sl]
        seg[tr_mask] = (seg[tr_mask] + tr_vety[tr_inv_t]).detach()
        hs_pc_ety[sl] = seg
        segc = hs_cnt[sl]
        segc[tr_mask] = segc[tr_mask] + 1
        hs_cnt[sl] = segc
        tr_p2v = tr_mask.nonzero().reshape(-1)[torch.from_numpy(tr_ind)] + init_left
        ac_p2v = ac_mask.nonzero().reshape(-1)[torch.from_numpy(ac_ind)] + ac_left
        ka, kt = _key3(ac_vox), _key3(tr_vox)
        ac_corr = np.isin(ka, kt)
        tr_corr = np.isin(kt, ka)
        assert ac_corr.any() and (tr_vox >= 0).all() and (ac_vox >= 0).all(), "No correspondence found!"
        out["ac_logits"].append(ac_vlog[torch.from_numpy(ac_corr)])
        out["ac_voxels"].append(torch.from_numpy(ac_vox[ac_corr]))
        out["tr_logits"].append(tr_vlog[torch.from_numpy(tr_corr)].detach())
        out["tr_voxels"].append(torch.from_numpy(tr_vox[tr_corr]))
        voxel_ety_all.append(tr_vety)  # ALL hist voxels, not only matched ones (Q2)
        out["ac_p2v_indices"].append(ac_p2v[torch.from_numpy(np.nonzero(ac_corr)[0])])
        out["tr_p2v_indices"].append(tr_p2v[torch.from_numpy(np.nonzero(tr_corr)[0])])
        ac_left = ac_right
    valid = hs_cnt > 0
    hs_pc_ety[valid] = hs_pc_ety[valid] / hs_cnt[valid]
    if conf_perc is not None and conf_perc > 0.0:
        mask = refine_ety(torch.cat(voxel_ety_all, 0).detach(), torch.cat(out["tr_logits"], 0).argmax(1), conf_perc)
    else:
        mask = torch.ones(torch.cat(out["tr_voxels"], 0).shape[0], dtype=torch.bool)
    out["tr_conf_mask"] = mask
    out["hs_pc_ety"] = hs_pc_ety
    out["_voxel_ety_all"] = torch.cat(voxel_ety_all, 0).detach()   # not a key of the reference: what refine_ety saw (tests)
    return out


def cr_forward(d, conf_mask=True, ety_weight=True, self_corr=False):
    """CR_Block.forward restated (hr_module.py:344-420); matched anchor/hist voxel lists are identical and sorted, so both
    argsorts of the reference (:359-360) are the identity (Q3).  self_corr: the reference's flag that switches the entropy
    weights off (:347,373,379)."""
    ety_weight = ety_weight and not self_corr
    losses, left = [], 0
    for i, (ac, tr) in enumerate(zip(d["ac_logits"], d["tr_logits"])):
        right = left + ac.shape[0]
        if conf_mask:
            m = d["tr_conf_mask"][left:right]
            ac, tr = ac[m], tr[m]
        if ety_weight:
            a_e = d["ac_pc_ety"][d["ac_p2v_indices"][i]]
            t_e = d["tr_pc_ety"][d["tr_p2v_indices"][i]]
            w = torch.exp(-t_e) / (torch.exp(-t_e) + torch.exp(-a_e))
            w = w[m] if conf_mask else w
        else:
            w = torch.ones(ac.shape[0])
        left = right
        tr = torch.where(tr == 0.0, torch.full_like(tr, 1e-8), tr)  # Q6: clamp applied to probabilities
        ac = torch.where(ac == 0.0, torch.full_like(ac, 1e-8), ac)
        losses.append((w * F.kl_div(torch.log(ac), tr.detach(), reduction="none").sum(1)).mean())
    return sum(losses) / len(losses)
