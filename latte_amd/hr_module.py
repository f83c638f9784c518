"""ST-voxel aggregation modules with the reference's API (latte/models/hr_module.py:31-331 HR_Block,
:334-420 CR_Block; built by build_model_hrcr, latte/models/build.py:43-53).

Same constructor arguments, same forward signature, same 11-key output dict -- but ONE batched GPU pipeline over
all anchors of the batch instead of a Python loop with host numpy in the middle:

  pose transform (fp64) + range mask + voxel key      lt_st_keys            (anchor index in the key's top bits)
  unique voxels / first index / inverse               lt_sort_segments      (one radix sort for all anchors)
  softmax, point entropy                              lt_softmax_entropy
  scatter-mean of probs and entropies per voxel       lt_segment_reduce     (segments of the sorted keys)
  ST entropy back to the points                       lt_st_gather_back
  anchor/hist voxel correspondence                    lt_st_join            (sorted-set join, not a [Va,Vt] matrix)
  class-wise entropy quantile mask (refine_ety)       lt_class_quantile_mask

Reference quirks Q1 (shifted voxel coords), Q2 (mask over all hist voxels, indexed by matched ones), Q4, Q6 are
reproduced; see SURVEY.md §8a."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import ops
from .functional import gather_rows

__all__ = ["HR_Block", "CR_Block", "prob_2_entropy", "refine_ety"]


def prob_2_entropy(prob: torch.Tensor) -> torch.Tensor:
    """latte/models/losses.py:10-19 (elementwise; the fused sum over classes lives in lt_softmax_entropy)."""
    c = prob.shape[1]
    return -torch.mul(prob, torch.log2(prob + 1e-30)) / np.log2(c)


def refine_ety(ety: torch.Tensor, pseudo_label: torch.Tensor, conf_perc: float, num_classes: int | None = None):
    """latte/data/utils/refine_pseudo_labels.py:33-53: per predicted class, mask entropies above the class quantile.
    `ety` may be longer than `pseudo_label` (quirk Q2): only the first len(pseudo_label) entries can be masked."""
    n = pseudo_label.numel()
    mask = torch.ones(ety.numel(), dtype=torch.bool, device=ety.device)
    if n == 0:
        return mask
    if num_classes is None:
        num_classes = int(pseudo_label.max().item()) + 1
    mask[:n] = ops.class_quantile_mask(ety[:n].detach().float().contiguous(), pseudo_label.to(torch.int32).contiguous(),
                                       num_classes, float(conf_perc))
    return mask


class _VoxelMeanSoftmax(Function):
    """voxel_probs = scatter_mean(softmax(logits)[in-range points], voxel id) with gradients to the logits."""

    @staticmethod
    def forward(ctx, logits, selem, inverse, seg_start):
        probs, _ = ops.softmax_entropy(logits, want_ety=False)
        out = ops.segment_reduce(probs, selem, seg_start, "mean")
        ctx.save_for_backward(probs, inverse, seg_start)
        return out

    @staticmethod
    def backward(ctx, g):
        probs, inverse, seg_start = ctx.saved_tensors
        ge = ops.segment_reduce_bwd(g.contiguous(), inverse, seg_start, True)
        return ops.softmax_bwd(probs, ge), None, None, None


def _scene_id(s) -> int:
    s = str(s)
    return int(s.split("-")[-1]) if "-" in s else int(s)


def _decode_xyz(keys: torch.Tensor, cb: int) -> torch.Tensor:
    """voxel coordinates out of dense ST keys (anchor << 3cb | x << 2cb | y << cb | z)"""
    m = (1 << cb) - 1
    return torch.stack([(keys >> (2 * cb)) & m, (keys >> cb) & m, keys & m], 1).to(torch.int32)


class _HRResult(dict):
    """HR_Block's output dict (the reference's 11 keys) plus, as attributes, what CR_Block needs to treat all anchors as
    rows of ONE matrix: the anchor index of every matched row, the row offsets of the anchors (device), and the unsplit
    tensors the per-anchor lists are views of"""
    row_anchor = None      # int64 [M] device
    anchor_start = None    # int32 [B+1] device
    flat = None            # dict(ac_logits, tr_logits, ac_p2v_indices, tr_p2v_indices): the tensors before the split


class _CRLoss(Function):
    """CR_Block's loss over all anchors at once (lt_cr_loss_fwd / lt_cr_loss_bwd); gradient to the anchor probabilities."""

    @staticmethod
    def forward(ctx, ac, tr, anchor_start, row_anchor, B, keep, ac_pc_ety, ac_p2v, tr_pc_ety, tr_p2v):
        loss, row_w, stats, keep8 = ops.cr_loss_fwd(ac, tr, anchor_start, B, keep, ac_pc_ety, ac_p2v, tr_pc_ety, tr_p2v)
        ctx.save_for_backward(ac, tr, row_w, stats, row_anchor, keep8)
        ctx.B = B
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        ac, tr, row_w, stats, row_anchor, keep8 = ctx.saved_tensors
        g_ac = ops.cr_loss_bwd(ac, tr, row_w, keep8, row_anchor, stats, ctx.B, g.float().contiguous())
        return g_ac, None, None, None, None, None, None, None, None, None


class HR_Block(nn.Module):
    def __init__(self, temp_wd_size: int = 50, max_range: int = 50, voxel_size: float = 0.2, conf_perc: float = 0.75,
                 exclude_labels: Tuple[int] = None, resize: Tuple[int, int] = None):
        super().__init__()
        self.temp_wd_size = temp_wd_size
        self.temp_bond = math.floor(temp_wd_size / 2)
        self.max_range = max_range
        self.voxel_size = voxel_size
        self.min_offset = -math.ceil(max_range / voxel_size) - 1  # fixed origin (hr_module.py:53)
        self.conf_perc = conf_perc
        self.exclude_labels = torch.Tensor(exclude_labels) if exclude_labels else None
        self.resize = resize

    # ------------------------------------------------------------------ host-side window bookkeeping
    def _plan(self, anchor_batch, ac_ref, hist_batch, hs_ref):
        a_scene = [_scene_id(s) for s in anchor_batch["scene_name"]]
        h_scene = [_scene_id(s) for s in hist_batch["scene_name"]]
        a_id, h_id = list(anchor_batch["scan_ID"]), list(hist_batch["scan_ID"])
        h_sizes = [int(hist_batch[hs_ref][j].shape[0]) for j in range(len(h_id))]
        h_start = np.concatenate([[0], np.cumsum(h_sizes)]).astype(np.int64)
        segs = []  # (anchor, frame, src_start, n, row_start)
        for i in range(len(a_id)):
            corr = [j for j in range(len(h_id))
                    if h_scene[j] == a_scene[i] and abs(float(h_id[j]) - float(a_id[i])) <= self.temp_bond]
            if not corr:
                raise ValueError(f"HR_Block: anchor {i} has no frame of its scene in the hist batch")
            # the reference takes the hist LOGITS (and writes the entropies back) sequentially from the first matching
            # frame on, one slice per frame of that frame's size (hr_module.py:203-220,256-263: quirk Q5); the points
            # come from the frame itself.  For a contiguous window (its sequential loader) row_start == src_start; the
            # slices never run past the hist rows (they are no longer than the frames from corr[0] on).
            row = int(h_start[corr[0]])
            for j in corr:
                segs.append((i, j, int(h_start[j]), h_sizes[j], row))
                row += h_sizes[j]
        return segs, h_start, h_sizes

    def geometry(self, anchor_batch: Dict, anchor_type: str, hist_batch: Dict, hist_type: str, device=None) -> Dict:
        """Everything of the block that depends on points and poses only (no logits): ST-voxel keys, the sort, voxel
        segments and the anchor/hist correspondence.  All host synchronisations of the block are in here, so a caller
        can run it on a side stream while the networks that produce the logits are still executing."""
        required = ["pose", "scene_name", "scan_ID", "ori_points"]
        assert set(required) <= set(anchor_batch.keys())
        assert set(required) <= set(hist_batch.keys())
        assert anchor_type.upper() in ["2D", "3D"] and hist_type.upper() in ["2D", "3D"]
        ac_ref = "img_indices" if anchor_type.upper() == "2D" else "ori_points"
        hs_ref = "img_indices" if hist_type.upper() == "2D" else "ori_points"
        dev = device if device is not None else anchor_batch["ori_points"][0].device
        B = len(anchor_batch["pose"])
        segs, h_start, h_sizes = self._plan(anchor_batch, ac_ref, hist_batch, hs_ref)
        a_sizes = [int(anchor_batch[ac_ref][i].shape[0]) for i in range(B)]
        for i in range(B):
            assert anchor_batch["ori_points"][i].shape[0] == a_sizes[i]
        for j in range(len(h_sizes)):
            assert hist_batch["ori_points"][j].shape[0] == h_sizes[j]

        # ---- device inputs
        h_pts = torch.cat([p.reshape(-1, 3) for p in hist_batch["ori_points"]], 0).to(dev, torch.float32)
        a_pts = torch.cat([p.reshape(-1, 3) for p in anchor_batch["ori_points"]], 0).to(dev, torch.float32)
        Ph, Pa = h_pts.shape[0], a_pts.shape[0]
        seg_n = np.array([s[3] for s in segs], dtype=np.int64)
        seg_out = np.concatenate([[0], np.cumsum(seg_n)]).astype(np.int64)
        E = int(seg_out[-1])
        poses_a = [torch.as_tensor(p, dtype=torch.float64).cpu() for p in anchor_batch["pose"]]
        poses_h = [torch.as_tensor(p, dtype=torch.float64).cpu() for p in hist_batch["pose"]]
        inv_a = [torch.linalg.inv(p) for p in poses_a]
        seg_T = torch.stack([(inv_a[i] @ poses_h[j])[:3].reshape(12) for (i, j, _, _, _) in segs], 0)
        Wmax = max(sum(1 for s in segs if s[1] == j) for j in range(len(h_sizes)))
        frame_seg = np.full((len(h_sizes), max(Wmax, 1)), -1, dtype=np.int32)
        fill = [0] * len(h_sizes)
        for si, (i, j, _, _, _) in enumerate(segs):  # segs are anchor-major => ascending anchor per frame
            frame_seg[j, fill[j]] = si
            fill[j] += 1
        a_start = np.concatenate([[0], np.cumsum(a_sizes)]).astype(np.int64)
        # one small H2D per table
        d = lambda a, dt: torch.as_tensor(a, dtype=dt).to(dev)
        seg_out_d = d(seg_out, torch.int64)
        seg_src_d = d(np.array([s[2] for s in segs], dtype=np.int64), torch.int64)
        seg_anchor_d = d(np.array([s[0] for s in segs], dtype=np.int32), torch.int32)
        seg_T_d = seg_T.to(dev)
        a_seg_out_d = d(a_start, torch.int64)
        a_seg_anchor_d = torch.arange(B, dtype=torch.int32, device=dev)

        # ---- keys, sort, segments
        # dense keys: anchor bits on top of 3 x cb coordinate bits -> the radix sorts run over key_bits + 1 bits
        # (a2d2: 3 + 30, nuScenes: 3 + 27) instead of 63
        cb = ops.st_coord_bits(self.max_range, self.voxel_size)
        key_bits = 3 * cb + max(1, (B - 1).bit_length())
        h_keys, h_src = ops.st_keys(h_pts, seg_out_d, seg_src_d, seg_anchor_d, seg_T_d, E, self.max_range,
                                    self.voxel_size, self.min_offset, cb)
        a_keys, _ = ops.st_keys(a_pts, a_seg_out_d, a_seg_out_d[:-1].contiguous(), a_seg_anchor_d, None, Pa,
                                self.max_range, self.voxel_size, self.min_offset, cb)
        contiguous = all(s[2] == s[4] for s in segs)
        seg_row_d = None
        if not contiguous:
            # quirk Q5 (see _plan): element e of segment s stands for hist ROW row_start[s] + (e - seg_out[s]) -- its logits,
            # its tr_p2v index and the row its entropy is written back to -- not for the point it was made from
            seg_row_d = d(np.array([s[4] for s in segs], dtype=np.int64), torch.int64)
            elem = torch.arange(E, device=dev, dtype=torch.int64)
            seg_of = torch.bucketize(elem, seg_out_d[1:], right=True)
            h_src = (seg_row_d[seg_of] + (elem - seg_out_d[seg_of])).to(torch.int32)
        h_selem, h_inv, h_uk, h_seg = ops.sort_segments(h_keys, key_bits)
        a_selem, a_inv, a_uk, a_seg = ops.sort_segments(a_keys, key_bits)

        # ---- correspondence
        a_sel, t_sel = ops.st_join(a_uk, h_uk)
        a_sel_l, t_sel_l = a_sel.long(), t_sel.long()
        matched_keys = a_uk[a_sel_l]
        per_anchor = torch.bincount((matched_keys >> (3 * cb)), minlength=B).tolist()
        if min(per_anchor) == 0:
            print("No correspondence found!")
            raise AssertionError
        ac_first = a_selem[a_seg[:-1].long()].long()            # first in-range point of each anchor voxel
        tr_first = h_src[h_selem[h_seg[:-1].long()].long()].long()  # ... of each hist voxel, as hist point index
        # source row of every sorted hist element, composed once: the two segmented means of forward() then follow
        # seg_start -> row -> probs instead of seg_start -> sorted_elem -> h_src -> probs
        h_row = h_src[h_selem.long()]
        return dict(Ph=Ph, Pa=Pa, B=B, h_src=h_src, h_selem=h_selem, h_row=h_row, h_inv=h_inv, h_seg=h_seg, a_selem=a_selem,
                    a_inv=a_inv, a_seg=a_seg, a_sel_l=a_sel_l, t_sel_l=t_sel_l, per_anchor=per_anchor,
                    vox=_decode_xyz(matched_keys, cb), ac_p2v=ac_first[a_sel_l], tr_p2v=tr_first[t_sel_l],
                    row_anchor=(matched_keys >> (3 * cb)),
                    anchor_start=d(np.concatenate([[0], np.cumsum(per_anchor)]), torch.int32),
                    h_start_d=d(h_start, torch.int64), frame_seg_d=d(frame_seg, torch.int32), seg_out_d=seg_out_d,
                    seg_row_d=seg_row_d)

    def forward(self, anchor_batch: Dict, anchor_type: str, hist_batch: Dict, hist_type: str, geo: Dict = None) -> Dict:
        assert "seg_logit" in anchor_batch and "seg_logit" in hist_batch
        a_logits, h_logits = anchor_batch["seg_logit"], hist_batch["seg_logit"]
        dev = a_logits.device
        C = a_logits.shape[1]
        if geo is None:
            geo = self.geometry(anchor_batch, anchor_type, hist_batch, hist_type, device=dev)
        Ph, Pa = geo["Ph"], geo["Pa"]
        assert h_logits.shape[0] == Ph and a_logits.shape[0] == Pa
        h_src, h_selem, h_seg = geo["h_src"], geo["h_selem"], geo["h_seg"]
        a_sel_l, t_sel_l, per_anchor = geo["a_sel_l"], geo["t_sel_l"], geo["per_anchor"]

        # ---- per-voxel means
        h_probs, h_ety = ops.softmax_entropy(h_logits.detach().float().contiguous())
        tr_voxel_probs = ops.segment_reduce(h_probs, geo["h_row"], h_seg, "mean")
        tr_voxel_ety = ops.segment_reduce(h_ety, geo["h_row"], h_seg, "mean")
        a_in = a_logits if a_logits.dtype == torch.float32 else a_logits.float()
        ac_voxel_probs = _VoxelMeanSoftmax.apply(a_in.contiguous(), geo["a_selem"], geo["a_inv"], geo["a_seg"])

        # ---- ST-voxel entropy written back to the hist points (mean over the anchors that see the point)
        if geo.get("seg_row_d") is None:
            hs_pc_ety = ops.st_gather_back(tr_voxel_ety, geo["h_inv"], geo["h_start_d"], geo["frame_seg_d"], geo["seg_out_d"], Ph)
        else:   # non-contiguous windows: the write-back follows the reference's row slices
            hs_pc_ety = ops.st_gather_back_rows(tr_voxel_ety, geo["h_inv"], geo["seg_row_d"], geo["seg_out_d"], Ph)

        vox = geo["vox"]
        split = lambda t: list(torch.split(t, per_anchor, 0))
        tr_logits_all = tr_voxel_probs[t_sel_l]
        ac_logits_all = gather_rows(ac_voxel_probs, a_sel_l)
        out = _HRResult({
            "ac_logits": split(ac_logits_all),
            "ac_voxels": split(vox),
            "tr_logits": split(tr_logits_all),
            "tr_voxels": split(vox.clone()),
            "tr_ety_loss": [],
            "tr_pc2img_mask": [],
            "tr_img_points": [],
            "ac_p2v_indices": split(geo["ac_p2v"]),
            "tr_p2v_indices": split(geo["tr_p2v"]),
            "hs_pc_ety": hs_pc_ety,
        })
        out.row_anchor = geo["row_anchor"]   # anchor of every matched row (device), for CR_Block's per-anchor means
        out.anchor_start = geo["anchor_start"]
        out.flat = {"ac_logits": ac_logits_all, "tr_logits": tr_logits_all, "ac_p2v_indices": geo["ac_p2v"],
                    "tr_p2v_indices": geo["tr_p2v"]}
        if self.conf_perc is not None and self.conf_perc > 0.0:
            out["tr_conf_mask"] = refine_ety(tr_voxel_ety, tr_logits_all.argmax(1), self.conf_perc, num_classes=C)
        else:
            out["tr_conf_mask"] = torch.ones(t_sel_l.numel(), dtype=torch.bool, device=dev)
        return out


class CR_Block(nn.Module):
    """Entropy-weighted KL between matched anchor / hist voxel distributions (hr_module.py:344-420)."""

    def __init__(self, reduction="mean", conf_mask=False, ety_weight=False):
        super().__init__()
        self.reduction = reduction
        self.conf_mask = conf_mask
        self.ety_weight = ety_weight

    @staticmethod
    def _flat(logit_dict, key: str, sizes):
        """all anchors' rows of logit_dict[key] as ONE tensor: the tensor HR_Block split the list from when the list still is
        that split (same storage, same offsets -- the reference's callers pass HR_Block's dict on unchanged,
        latte/tta/tools/latte.py:276-291), a concatenation otherwise"""
        lst = logit_dict[key]
        flat = getattr(logit_dict, "flat", None)
        t = flat.get(key) if flat else None
        if t is not None and t.shape[0] == sum(sizes) and len(lst) == len(sizes):
            row_bytes = t.stride(0) * t.element_size() if t.dim() > 0 else 0
            off, ok = 0, True
            for part, n in zip(lst, sizes):
                if part.shape[0] != n or (n and part.data_ptr() != t.data_ptr() + off * row_bytes):
                    ok = False
                    break
                off += n
            if ok:
                return t
        return torch.cat(list(lst), 0)

    def forward(self, logit_dict: Dict[str, torch.Tensor], self_corr: bool = False) -> torch.Tensor:
        """-> mean over the anchors of the masked mean of  w * KL(tr || ac)  over the anchor's matched voxels.  Two HIP
        launches forward, one backward (lt_cr_loss_fwd / _bwd) for ALL anchors; matched anchor / hist voxel lists are identical
        and sorted, so the reference's two argsorts (:359-360) are the identity (Q3)."""
        ac_list = logit_dict["ac_logits"]
        sizes = [int(t.shape[0]) for t in ac_list]
        B, M = len(sizes), sum(sizes)
        ac = self._flat(logit_dict, "ac_logits", sizes)
        tr = self._flat(logit_dict, "tr_logits", sizes).detach()
        dev = ac.device
        anchor_start = getattr(logit_dict, "anchor_start", None)
        row_anchor = getattr(logit_dict, "row_anchor", None)
        if anchor_start is None or row_anchor is None or row_anchor.numel() != M or anchor_start.numel() != B + 1:
            anchor_start = torch.as_tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32).to(dev)
            row_anchor = torch.repeat_interleave(torch.arange(B, device=dev), torch.as_tensor(sizes, device=dev), output_size=M)
        ety = {}
        if self.ety_weight and not self_corr:
            ety = dict(ac_pc_ety=logit_dict["ac_pc_ety"].detach().float().contiguous(),
                       ac_p2v=self._flat(logit_dict, "ac_p2v_indices", sizes).long(),
                       tr_pc_ety=logit_dict["tr_pc_ety"].detach().float().contiguous(),
                       tr_p2v=self._flat(logit_dict, "tr_p2v_indices", sizes).long())
        keep = logit_dict["tr_conf_mask"][:M].contiguous() if self.conf_mask else None
        return _CRLoss.apply(ac.float().contiguous(), tr.float().contiguous(), anchor_start, row_anchor, B, keep,
                             ety.get("ac_pc_ety"), ety.get("ac_p2v"), ety.get("tr_pc_ety"), ety.get("tr_p2v"))
