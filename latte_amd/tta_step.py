"""One Latte TTA iteration, 3-D side (the unit BASELINE.json's metric counts).

Host-side restatement of latte/tta/tools/latte.py:133-323 with the 2-D SegFormer replaced by given 2-D logits
(the 2-D branch is out of scope; the synthetic bench feeds random camera-FOV logits):
  teacher 3-D forward (no grad) -> inverse_to_all -> camera-FOV subset                    latte.py:156-167
  student 3-D forward -> inverse_to_all                                                   :176-178
  HR_Block(student 2D | teacher 3D), HR_Block(student 3D | teacher 2D)                    :199-202
  entropy-weighted pseudo-label fusion + refine_pseudo_labels                             :207-238
  CR_Block x2, cross-entropy on the pseudo labels                                         :276-303
  backward to the norm parameters, (multi-GPU: one flat RCCL all-reduce), Adam, EMA       :313-323
Model configuration helpers mirror latte/common/utils/ema_util.py:6-41,74-118."""
from __future__ import annotations

import time
from copy import deepcopy
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .functional import gather_rows
from .hr_module import CR_Block, HR_Block
from .spvcnn import inverse_to_all

__all__ = ["ensure_side_stream", "MAX_STEPS_IN_FLIGHT", "configure_model_norm", "configure_model_all", "create_ema_model", "update_ema_norm", "update_ema_all", "refine_pseudo_labels", "pc2img_logit",
           "LatteStep", "FlatAdamEMA", "shard_scan_offset", "shard_sequences", "allreduce_mean_grads"]


def configure_model_norm(model: nn.Module, reset_dropout: bool = True) -> list:
    """ema_util.configure_model_norm (:6-41): train mode, only *norm* modules trainable, BN on batch statistics.
    -> [model, [norm module names, [dropout module names]]] -- the reference's return value, which its callers unpack as
    `model, _ = configure_model_norm(...)` (latte/tta/tta.py:130-136); the inner list nesting is the reference's own."""
    model.train()
    model.requires_grad_(False)
    norm_list, dropout_modules = [], []
    for mod_name, m in model.named_modules():
        name = type(m).__name__.lower()
        if "norm" in name:
            m.requires_grad_(True)
            norm_list.append(mod_name)
            if "batchnorm" in name:
                m.track_running_stats = False
                m.running_mean = None
                m.running_var = None
        elif reset_dropout and "dropout" in name:
            m.p = 0.0
            dropout_modules.append(mod_name)
    out = [model, [norm_list]]
    if reset_dropout:
        out[1].append([dropout_modules])
    return out


def configure_model_all(model: nn.Module, reset_dropout: bool = True) -> list:
    """ema_util.configure_model_all (:44-71, TRAIN.UPDATE_PARAMS == "all"): train mode, EVERY parameter trainable (the
    reference walks named_modules and enables grad on every module that is neither norm nor dropout -- the root module
    included), BatchNorm on batch statistics, dropout reset on request.  -> [model, [dropout module names]]"""
    model.train()
    model.requires_grad_(False)
    dropout_modules = []
    for mod_name, m in model.named_modules():
        name = type(m).__name__.lower()
        if "norm" in name:
            m.requires_grad_(True)
            m.track_running_stats = False
            m.running_mean = None
            m.running_var = None
        elif reset_dropout and "dropout" in name:
            m.p = 0.0
            dropout_modules.append(mod_name)
        else:
            m.requires_grad_(True)
    return [model, [dropout_modules]]


def create_ema_model(model: nn.Module) -> list:
    """ema_util.create_ema_model (:74-96): a detached deep copy with dropout off.  -> [ema_model, [dropout module names]]"""
    ema = deepcopy(model)
    for p in ema.parameters():
        p.detach_()
    dropout_modules = []
    for mod_name, m in ema.named_modules():
        if "dropout" in type(m).__name__.lower():
            m.p = 0.0
            dropout_modules.append(mod_name)
    return [ema, [dropout_modules]]


def _norm_params(model: nn.Module) -> List[torch.Tensor]:
    out = []
    for m in model.modules():
        if "norm" in type(m).__name__.lower():
            out.extend(m.parameters(recurse=False))
    return out


@torch.no_grad()
def update_ema_norm(ema_model: nn.Module, model: nn.Module, alpha_teacher: float = 0.99, iteration=None, params=None) -> nn.Module:
    """ema = alpha*ema + (1-alpha)*student on the norm parameters (ema_util.update_ema_norm :99-118, same argument names
    and the same warm-up rule for `iteration`), as two multi-tensor ops.
    params: (ema list, student list) collected once by the caller (walking both module trees costs 1.2 ms per step)."""
    if iteration:
        alpha_teacher = min(1 - 1 / (iteration + 1), alpha_teacher)
    e, s = params if params is not None else (_norm_params(ema_model), _norm_params(model))
    torch._foreach_mul_(e, alpha_teacher)
    torch._foreach_add_(e, s, alpha=1.0 - alpha_teacher)
    return ema_model


@torch.no_grad()
def update_ema_all(ema_model: nn.Module, model: nn.Module, alpha_teacher: float = 0.99, iteration=None, params=None) -> nn.Module:
    """ema = alpha*ema + (1-alpha)*student on EVERY parameter (ema_util.update_ema_all :137-149, TRAIN.UPDATE_PARAMS == "all")."""
    if iteration:
        alpha_teacher = min(1 - 1 / (iteration + 1), alpha_teacher)
    e, s = params if params is not None else (list(ema_model.parameters()), list(model.parameters()))
    torch._foreach_mul_(e, alpha_teacher)
    torch._foreach_add_(e, s, alpha=1.0 - alpha_teacher)
    from . import functional as LF
    LF.bump_pack_epoch()   # the teacher's conv kernels changed: their packed MFMA images are stale
    return ema_model


def refine_pseudo_labels(probs: torch.Tensor, pseudo_label: torch.Tensor, ignore_label: int = -100,
                         num_classes: int | None = None) -> torch.Tensor:
    """latte/data/utils/refine_pseudo_labels.py:10-30: per class, labels below min(class median, 0.9) are ignored.
    One sort + one threshold kernel + one mask kernel (lt_class_median_mask) instead of a per-class Python loop."""
    if num_classes is None:
        num_classes = int(pseudo_label.max().item()) + 1
    keep = ops.class_median_mask(probs.float().contiguous(), pseudo_label.to(torch.int32).contiguous(), num_classes, 0.9)
    return torch.where(keep, pseudo_label, torch.full_like(pseudo_label, ignore_label))


def pc2img_logit(pc_logits: torch.Tensor, pc2img_idx_ls: List[torch.Tensor]):
    """latte/data/utils/all_to_partial.py:4-37 with return_mask=True: per-scan bool maps -> (mask over all points,
    logits of the in-FOV points)."""
    mask = torch.cat(list(pc2img_idx_ls), 0)
    return mask, pc_logits[mask]


def cross_entropy(logits: torch.Tensor, target: torch.Tensor, weight=None, ignore_index: int = -100) -> torch.Tensor:
    """F.cross_entropy(logits, target, weight, ignore_index=-100, reduction='mean') (latte.py:288-299) out of gather / sum
    kernels: torch's nll_loss forward and backward reduce in ONE workgroup (92 us + 67 us per call at 209 k points)."""
    logp = F.log_softmax(logits, 1)
    valid = target != ignore_index
    idx = torch.where(valid, target, torch.zeros_like(target))
    picked = logp.gather(1, idx.unsqueeze(1)).squeeze(1)
    w = valid.to(logp.dtype) if weight is None else weight[idx] * valid
    return -(picked * w).sum() / w.sum()


def allreduce_mean_grads(params: List[torch.Tensor], world: int, group=None) -> None:
    """The only collective of the path: ONE flat all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests) of the
    gradients of `params` (the norm parameters: 13 248 floats at cr=1), averaged over the ranks, written back as views
    of the flat buffer.  Parameters without a gradient contribute zeros."""
    if world == 1:
        return
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    if flat.is_cuda and torch.distributed.get_backend(group) == "gloo":
        # rehearsals of the multi-rank path without RCCL (ranks sharing a GPU): 53 KB through the host
        host = flat.cpu()
        torch.distributed.all_reduce(host, group=group)
        flat.copy_(host)
    else:
        torch.distributed.all_reduce(flat, group=group)
    flat.div_(world)
    off = 0
    for p, g in zip(params, grads):
        n = g.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n


class FlatAdamEMA:
    """torch.optim.Adam over the norm parameters + update_ema_norm as ONE launch over flat buffers
    (latte/tta/tools/latte.py:317-323, latte/common/utils/ema_util.py:99-118).

    The student's trainable parameters and the teacher's counterparts are re-pointed at views of two flat fp32 buffers
    (13 248 scalars at cr = 1), gradients are gathered into a third with one multi-tensor copy -- which is also the
    buffer the RCCL all-reduce runs on -- and lt_adam_ema_step does Adam + EMA in one kernel.  Same hyper-parameter
    names and defaults as torch.optim.Adam (amsgrad / maximize are not supported)."""

    def __init__(self, params, ema_params=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_alpha=0.99,
                 allow_missing_grads=False):
        # allow_missing_grads: a parameter without a gradient counts as a zero gradient (update_params="all": the second
        # head is outside the Latte losses).  torch.optim.Adam skips such a parameter instead; the result is the same
        # while weight_decay == 0 and the parameter never gets a gradient later (its moments stay zero).
        self.allow_missing = bool(allow_missing_grads)
        if self.allow_missing and weight_decay != 0.0:
            # torch.optim.Adam SKIPS a parameter without a gradient; a zero gradient is the same only without weight decay
            raise ValueError("FlatAdamEMA: allow_missing_grads counts a missing gradient as zero, which differs from "
                             "torch.optim.Adam (it skips the parameter) once weight_decay != 0")
        self.params = list(params)
        if not self.params:
            raise ValueError("FlatAdamEMA: no parameters")
        self.lr, self.betas, self.eps, self.weight_decay, self.ema_alpha = lr, tuple(betas), eps, weight_decay, ema_alpha
        self.flat_p, _ = self._flatten(self.params)
        self.flat_e = None
        if ema_params is not None:
            ema_params = list(ema_params)
            if [tuple(p.shape) for p in ema_params] != [tuple(p.shape) for p in self.params]:
                raise ValueError("FlatAdamEMA: teacher and student parameter lists differ")
            self.flat_e, _ = self._flatten(ema_params)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.g_views, off = [], 0
        for p in self.params:
            self.g_views.append(self.flat_g[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.steps = 0
        self._ema_params = ema_params

    def _check_aliasing(self) -> None:
        """The parameters must still be views of the flat buffers the kernel updates.  Module._apply (model.cuda() /
        .float() / .to()) or another `.data` assignment re-points them silently -- the kernel would go on updating
        buffers the model no longer sees.  First and last parameter of both lists: O(1) per step."""
        for flat, plist, who in ((self.flat_p, self.params, "student"), (self.flat_e, self._ema_params, "teacher")):
            if flat is None:
                continue
            last = plist[-1]
            ok = (plist[0].data_ptr() == flat.data_ptr()
                  and last.data_ptr() == flat.data_ptr() + 4 * (flat.numel() - last.numel()))
            if not ok:
                raise RuntimeError(f"FlatAdamEMA: the {who} parameters no longer alias the optimizer's flat buffer (the model "
                                   "was moved / cast / had .data reassigned after the optimizer was built); rebuild the "
                                   "optimizer (LatteStep) after such a change")

    @staticmethod
    def _flatten(params):
        if not all(p.is_cuda and p.dtype == torch.float32 for p in params):
            raise ValueError("FlatAdamEMA: fp32 GPU parameters only (move the model to the GPU first)")
        flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
        off = 0
        for p in params:
            p.data = flat[off:off + p.numel()].view_as(p)   # the parameter now aliases its slice of the flat buffer
            off += p.numel()
        return flat, off

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            p.grad = None

    def gather_grads(self) -> torch.Tensor:
        """p.grad of every parameter -> the flat gradient buffer (one multi-tensor copy)"""
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads) and self.allow_missing:
            have = [(v, g) for v, g in zip(self.g_views, grads) if g is not None]
            for v, g in zip(self.g_views, grads):
                if g is None:
                    v.zero_()
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
            return self.flat_g
        if any(g is None for g in grads):
            missing = [i for i, g in enumerate(grads) if g is None]
            raise RuntimeError(f"FlatAdamEMA: parameters {missing[:4]}... have no gradient; every norm layer of SPVCNN is on "
                               "the path of the TTA losses, so this is a broken graph")
        torch._foreach_copy_(self.g_views, grads)
        return self.flat_g

    def state_dict(self) -> Dict:
        """optimizer state for checkpoint / resume (the parameters themselves are in the models' state_dicts)"""
        return {"steps": self.steps, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "hyper": {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                          "ema_alpha": self.ema_alpha}}

    def load_state_dict(self, state: Dict) -> None:
        if state["exp_avg"].numel() != self.exp_avg.numel():
            raise ValueError(f"FlatAdamEMA.load_state_dict: {state['exp_avg'].numel()} scalars in the checkpoint, "
                             f"{self.exp_avg.numel()} in this optimizer")
        self.steps = int(state["steps"])
        self.exp_avg.copy_(state["exp_avg"])
        self.exp_avg_sq.copy_(state["exp_avg_sq"])
        for k, v in state.get("hyper", {}).items():
            setattr(self, k, tuple(v) if k == "betas" else v)

    def step(self, grad_scale: float = 1.0) -> None:
        self._check_aliasing()
        self.steps += 1
        ops.adam_ema_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.flat_e, self.lr, self.betas[0],
                          self.betas[1], self.eps, self.weight_decay, self.steps, self.ema_alpha, grad_scale)


def shard_scan_offset(rank: int, scans_per_rank: int) -> int:
    """first global scan id of a rank: ranks own whole, disjoint sequences (no halo, no data-path collective)"""
    return rank * scans_per_rank


def shard_sequences(n_seq: int, world: int, rank: int):
    """-> (first sequence, number of sequences) of `rank` when `n_seq` whole sequences are dealt to `world` ranks in
    contiguous, disjoint runs (BASELINE configs[3]: 64 scans = 8 sequences x 8 frames over 2/4/8 GPUs).  A sequence is
    never split: HR_Block only relates frames of one scene (latte/models/hr_module.py:192-195), so no halo is needed."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    lo = (n_seq * rank) // world
    hi = (n_seq * (rank + 1)) // world
    return lo, hi - lo


def _record_stream_tree(obj, stream, seen=None) -> None:
    """tensor.record_stream(stream) for every tensor reachable from obj (dicts, sequences, attribute dicts): memory
    allocated on a side stream and used on `stream` must not be handed out again before `stream` is done with it."""
    if seen is None:
        seen = set()
    if obj is None or id(obj) in seen or isinstance(obj, (int, float, str, bool)):
        return
    seen.add(id(obj))
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream_tree(v, stream, seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream_tree(v, stream, seen)
    elif hasattr(obj, "__dict__") and type(obj).__module__.startswith("latte_amd"):
        _record_stream_tree(vars(obj), stream, seen)


# diagnostics only (tools): keep the index structures of the batch across steps to see what building them costs
_KEEP_CACHE = bool(__import__("os").environ.get("LATTE_DIAG_KEEP_INDEX_CACHE"))


_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}

# Stream discipline of a rank (DESIGN 6b "Streams: the invariants").  A process owns few hardware queues (GPU_MAX_HW_QUEUES =
# 4 on the box) and HIP deals streams onto them round-robin, so the product creates exactly ONE stream of its own per device
# per process -- the side stream below -- and otherwise lives on the caller's current stream; with the nccl backend torch's
# process group adds one collective stream: main + side + collective = 3 <= 4 queues.  MAX_STEPS_IN_FLIGHT bounds how far
# the host may run ahead of the GPU (step i is not issued before step i - MAX_STEPS_IN_FLIGHT has finished on the GPU).
MAX_STEPS_IN_FLIGHT = 2


def ensure_side_stream(device=None) -> "torch.cuda.Stream":
    """The side stream of `device` (default: the current device), created on first use -- ONE per device per process,
    high priority.  Call it right after torch.cuda.set_device() and BEFORE torch.distributed.init_process_group in a
    multi-rank process, so that the order in which the process creates its streams (and with it the stream -> hardware
    queue assignment) is the same on every rank and in every run: main (the null stream, exists), side, collective."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    s = _SIDE_STREAMS.get(dev)
    if s is None:
        prio = int(__import__("os").environ.get("LATTE_DIAG_SIDE_PRIORITY", "-1"))  # small dependent kernels: ahead of the big ones
        with torch.cuda.device(dev):
            s = _SIDE_STREAMS[dev] = torch.cuda.Stream(priority=prio)
    return s


class LatteStep:
    def __init__(self, model_3d: nn.Module, num_classes: int, hr_cfg: Dict, lr: float = 1e-3, conf_perc=None,
                 lambda_hr_23: float = 0.3, lambda_hr_32: float = 0.3, lambda_ety_ps: float = 1.0, class_weights=None,
                 process_group=None, side_stream: bool = True, fused_optimizer: bool = True, sync_bn: bool = False,
                 update_params: str = "norm", force_collective: bool = False):
        # update_params: cfg.TRAIN.UPDATE_PARAMS (latte/tta/tta.py:129-138) -- "norm" (every shipped latte.yaml: only the
        # norm layers' gamma / beta train) or "all" (every parameter trains: conv kernels through lt_spconv_bwd_weight)
        self.update_params = str(update_params).lower()
        if self.update_params == "norm":
            self.model, _ = configure_model_norm(model_3d, reset_dropout=True)
        elif self.update_params == "all":
            self.model, _ = configure_model_all(model_3d, reset_dropout=True)
        else:
            raise IndexError(f"Required update params has not been defined yet: {update_params}")   # the reference's error
        self.ema_model, _ = create_ema_model(self.model)
        self.hr = HR_Block(**hr_cfg)
        self.cr = CR_Block(reduction="mean", conf_mask=True, ety_weight=True)
        if self.update_params == "norm":
            self.params = _norm_params(self.model)
            self._ema_pairs = (_norm_params(self.ema_model), self.params)
        else:
            self.params = list(self.model.parameters())
            self._ema_pairs = (list(self.ema_model.parameters()), self.params)
        assert {id(p) for p in self.params} == {id(p) for p in self.model.parameters() if p.requires_grad}
        # fused_optimizer (default on the GPU): Adam + EMA as one launch over flat buffers; off: torch.optim.Adam and two
        # multi-tensor EMA ops (the A/B and the parity reference of tests/test_gpu_step.py)
        self.fused = bool(fused_optimizer) and self.params[0].is_cuda
        if self.fused:
            self.optimizer = FlatAdamEMA(self.params, self._ema_pairs[0], lr=lr, ema_alpha=0.99,
                                         allow_missing_grads=self.update_params != "norm")
        else:
            self.optimizer = torch.optim.Adam(self.params, lr=lr)
        self.conf_perc = hr_cfg.get("conf_perc") if conf_perc is None else conf_perc
        self.lam23, self.lam32, self.lam_ps = lambda_hr_23, lambda_hr_32, lambda_ety_ps
        self.class_weights = class_weights
        self.num_classes = num_classes
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        # rehearsals: issue the gradient all-reduce even in a one-rank group (the stream / thread topology of an N-rank
        # run -- main, side, the process group's collective stream and watchdog thread -- on a one-GPU box)
        self.force_collective = bool(force_collective) and process_group is not None
        # sync_bn (N > 1 only): the exact-parity mode of SURVEY 8e -- BatchNorm statistics and their backward sums are
        # all-reduced per layer, so the sharded step computes what ONE GPU would on the whole batch; the BatchNorm
        # parameter gradients are then global sums on every rank and are not all-reduced again
        self.sync_bn = bool(sync_bn) and self.world > 1
        if self.sync_bn and self.update_params != "norm":
            raise ValueError("sync_bn makes the BatchNorm gradients global sums; with update_params='all' the other "
                             "gradients would still need the all-reduce -- not supported together")
        self.timers = None
        self.trace = None
        self.debug_out = None   # tests: when a dict, forward_backward leaves its discrete decisions (pseudo labels, masks) in it
        # input-only, sync-heavy work (index structures, in-FOV tables, HR_Block.geometry) on a side stream: see
        # forward_backward
        self.hr_side_stream = side_stream
        self._side = None
        self._pre = None   # (batch, tables, geo23, geo32) built ahead by prefetch()
        if self.hr_side_stream:
            # two Python threads feed the GPU (this one and the prefetch helper): hand the GIL over every 0.5 ms instead of
            # CPython's default 5 ms, so neither stalls the other's launch stream for long
            import sys
            want = float(__import__("os").environ.get("LATTE_DIAG_SWITCH_INTERVAL", "5e-4"))
            if sys.getswitchinterval() > want:
                sys.setswitchinterval(want)
        self._inflight: List = []   # end-of-step events of the iterations the GPU may still be working on
        self.host_wait_s = 0.0      # seconds step() spent blocked on the run-ahead bound (the host was ahead of the GPU)
        if self.hr_side_stream and self.params[0].is_cuda:
            ensure_side_stream(self.params[0].device)   # created now, not inside the first step (deterministic creation order)

    # -------------------------------------------------------------------------------------------------
    def _tick(self, name: str):
        """segment timer for tools/step_breakdown.py (synchronises; off unless self.timers is a dict)"""
        if self.trace is not None:
            # non-synchronising trace for tools/timeline.py: host time of the call + an event on the current stream
            import time
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.trace.append((name, time.perf_counter(), ev))
            return
        if self.timers is None:
            return
        import time
        torch.cuda.synchronize()
        now = time.perf_counter()
        if getattr(self, "_t_last", None) is not None:
            self.timers[name] = self.timers.get(name, 0.0) + (now - self._t_last)
        self._t_last = now

    def _batch_tables(self, batch: Dict) -> Dict:
        """Per-batch index tables that depend on the inputs only: in-FOV point indices (the bool maps of
        latte/data/utils/all_to_partial.py:4-37 turned into one index vector, so later gathers/scatters need no host
        sync), the per-scan in-FOV points, the unique-voxel -> raw-point gather index of inverse_to_all."""
        pc2img = batch["pc2img_idx"]
        mask = torch.cat(list(pc2img), 0)
        fov_idx = torch.nonzero(mask).squeeze(1)      # the one host sync of the table (on the stream it is built on)
        counts = [int(t.shape[0]) for t in batch["img_indices"]]
        assert sum(counts) == fov_idx.numel(), "img_indices and pc2img_idx disagree"
        ori_all = torch.cat([p.reshape(-1, 3) for p in batch["ori_points"]], 0)
        ori_2d = list(torch.split(ori_all[fov_idx], counts, 0))
        sizes = [int(i.shape[0]) for i in batch["indices"]]
        inv, off = [], 0
        for n, iv in zip(sizes, batch["inverse_map"]):
            iv = iv if torch.is_tensor(iv) else torch.as_tensor(iv)
            inv.append(iv.to(mask.device, dtype=torch.int64) + off)
            off += n
        inv_all = torch.cat(inv, 0)
        # CSR of inv_all (raw points of every unique voxel, ascending): the heads' backward sums point gradients per voxel
        # over it instead of an atomic index_add
        inv_csr = ops.csr_build(inv_all.to(torch.int32), int(sum(sizes))) if inv_all.is_cuda else None
        return {"mask": mask, "fov_idx": fov_idx, "ori_2d": ori_2d, "inv_all": inv_all, "inv_csr": inv_csr}

    def _side_stream(self):
        if not (self.hr_side_stream and self.timers is None):
            return None
        if self._side is None:
            # ONE side stream per device for the whole process: a process has few hardware queues (GPU_MAX_HW_QUEUES = 4 on
            # the box) and HIP deals streams onto them round-robin, so a fresh stream per LatteStep instance eventually
            # lands on the main stream's queue and the "side" work queues up behind the convs (measured: the 4th
            # instance of a process ran 45 ms per step instead of 29)
            self._side = ensure_side_stream(self.params[0].device)
        # the invariant the stream plan rests on: this instance never holds a stream the process registry does not know
        assert self._side is _SIDE_STREAMS.get(self._side.device.index), "LatteStep: side stream outside the per-device registry"
        if torch.cuda.current_stream() == self._side:
            raise RuntimeError("LatteStep: called with the side stream as the current stream -- main and side must be two "
                               "streams (the side stream waits for events of the main one and vice versa)")
        return self._side

    def _hr_inputs(self, batch: Dict, tab: Dict):
        extra = {k: batch[k] for k in ("pose", "scene_name", "scan_ID", "img_indices")}
        a2 = dict(seg_logit=batch["logit_2d_student"], ori_points=tab["ori_2d"], **extra)
        h2 = dict(seg_logit=batch["logit_2d_teacher"], ori_points=tab["ori_2d"], **extra)
        a3 = dict(ori_points=batch["ori_points"], **extra)
        h3 = dict(ori_points=batch["ori_points"], **extra)
        return a2, h2, a3, h3

    @staticmethod
    def _ready_event(batch: Dict):
        """The event after which the inputs of `batch` (lidar.C/F, ori_points, pc2img_idx, 2-D logits) are complete:
        batch["ready_event"] when the producer recorded one on ITS stream (a loader's H2D copy stream,
        latte_amd.workload.build_batch), else an event recorded now on the caller's current stream -- which is where a
        batch that is still being produced asynchronously (quantize_scan_gpu, non_blocking copies) is being produced.
        The side stream waits for it before its first read: torch streams do not order against each other by themselves."""
        ev = batch.get("ready_event")
        if ev is None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        return ev

    def prefetch(self, batch: Dict) -> None:
        """Everything of one iteration that depends on the INPUTS of `batch` only -- the index structures of the backbone
        (voxel sets, kernel maps, conv plans, point<->voxel maps and CSRs), the in-FOV tables, the geometry of both HR
        calls: ~350 small kernels and ~25 host syncs -- built on the side stream by a HELPER THREAD, starting now.
        step(batch) later joins the thread and finds them ready; it builds them itself otherwise.  Nothing is reused: the
        structures are rebuilt from the coordinates for every iteration, one iteration ahead.

        Why a thread: the step is host-bound otherwise (bench.py's host_issue_ms_per_step = ms_per_step).  The
        input-only work costs the host 11 of its 27 ms per step, nearly all of it blocked in the syncs that return
        data-dependent sizes (each one drains the side stream); ctypes and torch release the GIL while they wait, so a
        second thread turns that waiting into issue time for the main stream.  The helper touches no state of the step
        in flight: it builds into a private SparseTensor view of the batch's lidar tensor and hands the result over at
        the join."""
        side = self._side_stream()
        if side is None:
            if callable(batch):
                raise RuntimeError("LatteStep.prefetch: a batch factory needs the side stream")
            return
        import threading
        from .tensor import SparseTensor
        # `batch` may be a FACTORY: a callable that produces the batch dict (raw points -> H2D -> quantize -> collate,
        # latte_amd.workload.batch_from_raw).  It then runs in the helper thread ON THE SIDE STREAM, one iteration ahead --
        # where a loader would put the input voxelization -- and step(None) picks its batch up.
        factory = batch if callable(batch) else None
        dev = self.params[0].device if factory is not None else batch["lidar"].C.device
        box: Dict = {}
        ready = None if factory is not None else self._ready_event(batch)   # on the calling thread: its current stream is the producer's

        def body():
            try:
                if _KEEP_CACHE and factory is None and getattr(self, "_diag_kept", None) is not None \
                        and self._diag_kept[0] is batch \
                        and __import__("os").environ.get("LATTE_DIAG_KEEP_INDEX_CACHE") not in ("prepare", "geometry"):
                    # diagnostics only (LATTE_DIAG_KEEP_INDEX_CACHE, not legitimate): hand the structures of the last
                    # prefetch of this very batch over again -- the step then runs WITHOUT the side stream's work, which
                    # bounds what that work costs the main stream
                    box.update(self._diag_kept[1])
                    return
                with torch.cuda.device(dev), torch.no_grad(), torch.cuda.stream(side):
                    if factory is not None:
                        b = factory()            # H2D + input voxelization + collate, on the side stream
                    else:
                        b = batch
                        side.wait_event(ready)
                    kept = getattr(self, "_diag_kept", None) if (_KEEP_CACHE and factory is None) else None
                    part = __import__("os").environ.get("LATTE_DIAG_KEEP_INDEX_CACHE", "")
                    if kept is not None and kept[0] is not batch:
                        kept = None
                    tab = self._batch_tables(b)
                    x = b["lidar"]
                    view = SparseTensor(x.F, x.C, x.s)   # same tensors, own cache slot
                    if kept is not None and part == "prepare":      # diagnostics: only the backbone's structures are reused
                        view._lt_cache = kept[1]["cache"]
                    self.model.net_3d.prepare(view)
                    if kept is not None and part == "geometry":     # diagnostics: only the HR geometry is reused
                        box["geo23"], box["geo32"] = kept[1]["geo23"], kept[1]["geo32"]
                    else:
                        a2, h2, a3, h3 = self._hr_inputs(b, tab)
                        box["geo23"] = self.hr.geometry(a2, "2D", h3, "3D")
                        box["geo32"] = self.hr.geometry(a3, "3D", h2, "2D")
                    box["batch"], box["tab"], box["cache"] = b, tab, view._lt_cache
                    # what the consumer waits for: this batch's structures, not whatever the side stream does next
                    box["done"] = torch.cuda.Event()
                    box["done"].record(side)
                    if _KEEP_CACHE and factory is None:
                        self._diag_kept = (batch, dict(box))
            except BaseException as e:  # re-raised by the step that joins
                box["error"] = e

        th = threading.Thread(target=body, name="latte-prefetch", daemon=True)
        th.start()
        self._pre = (batch, th, box)

    def _take_prefetched(self, batch):
        """-> (tables, geo23, geo32, done event, batch) built ahead for `batch` by prefetch() (joins the helper thread,
        installs the index structures on the batch), or None.  batch None: whatever batch the prefetched factory made."""
        pre, self._pre = self._pre, None
        if pre is None:
            return None
        b, th, box = pre
        th.join()
        if "error" in box:
            raise box["error"]
        if self._side_stream() is None:
            return None
        if batch is None and callable(b):
            batch = box["batch"]
            _record_stream_tree(batch, torch.cuda.current_stream())   # its tensors were allocated on the side stream
        elif b is not batch:
            return None
        batch["lidar"]._lt_cache = box["cache"]
        return box["tab"], box["geo23"], box["geo32"], box["done"], batch

    def forward_backward(self, batch: Dict, pre=None):
        """-> dict of losses.  batch: see latte_amd.workload.build_batch.

        Stream plan (side stream on, the default): everything that depends on the INPUTS only and needs host syncs --
        the index structures of the backbone, the in-FOV tables, the geometry of both HR calls -- runs on a
        high-priority side stream; the main stream carries the two forwards, the HR feature kernels, the losses and
        the backward without a single host sync, so the host stays ahead of the GPU for the whole iteration."""
        if self.timers is not None:
            import time
            torch.cuda.synchronize()
            self._t_last = time.perf_counter()
        ema_logit_2d = batch["logit_2d_teacher"]
        preds_2d = {"seg_logit": batch["logit_2d_student"]}
        main = torch.cuda.current_stream()
        side = self._side_stream()
        geo23 = geo32 = None
        if pre is not None:
            # built one iteration ahead by prefetch(); wait for THAT work only (the helper of the next batch may already be
            # queueing on the side stream)
            tab, geo23, geo32, done = pre[:4]
            main.wait_event(done)
            _record_stream_tree(batch["lidar"]._lt_cache, main)
            _record_stream_tree(tab, main)
            _record_stream_tree(geo23, main)
            _record_stream_tree(geo32, main)
        elif side is not None:
            side.wait_event(self._ready_event(batch))
            with torch.cuda.stream(side):
                tab = self._batch_tables(batch)
                self.model.net_3d.prepare(batch["lidar"])
            main.wait_stream(side)
            _record_stream_tree(batch["lidar"]._lt_cache, main)
            _record_stream_tree(tab, main)
        else:
            tab = self._batch_tables(batch)
        fov_idx = tab["fov_idx"]
        a2, h2, a3, h3 = self._hr_inputs(batch, tab)

        with torch.no_grad():
            ema3d = self.ema_model.forward_points(batch, tab["inv_all"])   # heads + inverse_to_all in one kernel
        self._tick("teacher_fwd")
        late_geo = side is not None and geo23 is None
        if late_geo:
            # the teacher forward is queued on the main stream; the sync-heavy geometry of both HR calls runs beside it
            with torch.cuda.stream(side):
                geo23 = self.hr.geometry(a2, "2D", h3, "3D")
                geo32 = self.hr.geometry(a3, "3D", h2, "2D")
        preds_3d = self.model.forward_points(batch, tab["inv_all"], tab.get("inv_csr"))
        self._tick("student_fwd")
        if late_geo:
            main.wait_stream(side)
            _record_stream_tree(geo23, main)
            _record_stream_tree(geo32, main)
        a3["seg_logit"] = preds_3d["seg_logit"]
        h3["seg_logit"] = ema3d["seg_logit"]
        hr23 = self.hr(a2, "2D", h3, "3D", geo=geo23)
        hr32 = self.hr(a3, "3D", h2, "2D", geo=geo32)
        self._tick("hr_x2")

        # ---- pseudo labels (latte.py:207-238); [pc2img_mask] indexing of the reference = gathers/scatters by fov_idx
        with torch.no_grad():
            st3, st2 = hr23["hs_pc_ety"], hr32["hs_pc_ety"]
            thr2 = thr3 = None
            if self.conf_perc is not None and self.conf_perc >= 0:
                thr3 = ops.quantile(st3, self.conf_perc)
                thr2 = ops.quantile(st2, self.conf_perc)
            # softmax x2, point entropies, ST-entropy substitution, exp(-entropy) blend inside the FOV, max: 2 kernels
            pc2, pc3, ps_conf, ps_label = ops.pseudo_label_fuse(ema_logit_2d, ema3d["seg_logit"], fov_idx, st2, st3,
                                                                thr2, thr3)
            ps_label = refine_pseudo_labels(ps_conf, ps_label, num_classes=self.num_classes)

        self._tick("pseudo_labels")
        loss_2d, loss_3d = [], []
        hr23.update({"ac_pc_ety": pc2, "tr_pc_ety": pc3})
        l23 = self.cr(hr23)
        loss_2d.append(self.lam23 * l23)
        hr32.update({"ac_pc_ety": pc3, "tr_pc_ety": pc2})
        l32 = self.cr(hr32)
        loss_3d.append(self.lam32 * l32)
        pl2 = cross_entropy(preds_2d["seg_logit"], ps_label[fov_idx].long(), weight=self.class_weights)
        pl3 = cross_entropy(preds_3d["seg_logit"], ps_label.long(), weight=self.class_weights)
        loss_2d.append(pl2)
        loss_3d.append(pl3)
        self._tick("cr_ce")
        if self.debug_out is not None:
            self.debug_out.update(ps_label=ps_label, conf_mask_23=hr23["tr_conf_mask"], conf_mask_32=hr32["tr_conf_mask"])
        (sum(loss_3d) + sum(loss_2d)).backward()
        self._tick("backward")
        return {"hr_23": l23.detach(), "hr_32": l32.detach(), "pl_2d": pl2.detach(), "pl_3d": pl3.detach()}

    def allreduce_grads(self):
        allreduce_mean_grads(self.params, self.world, self.pg)

    def _update(self) -> None:
        """all-reduce (N > 1), optimizer step, EMA teacher update (latte.py:317-323)"""
        if not self.fused:
            if self.sync_bn:
                for p in self.params:      # already the global sums of the per-rank losses: the mean over ranks
                    p.grad.div_(self.world)
            else:
                self.allreduce_grads()
            self.optimizer.step()
            if self.update_params == "norm":
                update_ema_norm(self.ema_model, self.model, 0.99, params=self._ema_pairs)
            else:
                update_ema_all(self.ema_model, self.model, 0.99, params=self._ema_pairs)
            return
        flat = self.optimizer.gather_grads()
        if (self.world > 1 or self.force_collective) and not self.sync_bn:
            # the only collective of the path: ONE all-reduce (RCCL over xGMI) of the flat gradient buffer; the mean is
            # folded into the update kernel (grad_scale)
            if flat.is_cuda and torch.distributed.get_backend(self.pg) == "gloo":
                host = flat.cpu()
                torch.distributed.all_reduce(host, group=self.pg)
                flat.copy_(host)
            else:
                torch.distributed.all_reduce(flat, group=self.pg)
        self.optimizer.step(grad_scale=1.0 / self.world)
        if self.update_params != "norm":
            from . import functional as LF
            LF.bump_pack_epoch()   # conv kernels of student and teacher were written by the kernel: re-pack on next use

    def _with_sync_bn(self, fn, *args):
        from . import functional as LF
        if not self.sync_bn:
            return fn(*args)
        prev, LF.SYNC_BN_GROUP = LF.SYNC_BN_GROUP, self.pg
        try:
            return fn(*args)
        finally:
            LF.SYNC_BN_GROUP = prev

    def step(self, batch: Dict | None, next_batch=None):
        """One TTA iteration on `batch`.  next_batch (optional): the batch of the NEXT iteration -- a helper thread starts
        building its input-only structures on the side stream right away (prefetch), beside this iteration -- or a
        callable that makes it (see prefetch); the next call is then step(None, ...): "the batch announced last time"."""
        self._tick("step_begin")
        # bound the host's run-ahead: iteration i is not issued before iteration i - MAX_STEPS_IN_FLIGHT has finished on
        # the GPU (a host wait on an event recorded at the end of that step; free while the step is GPU-bound, the host
        # is < 1 step ahead then).  It keeps the number of outstanding packets / completion signals / allocator blocks of
        # every stream bounded by two iterations' worth, whatever the caller's loop does.
        if len(self._inflight) >= MAX_STEPS_IN_FLIGHT:
            t0 = time.perf_counter()
            while len(self._inflight) >= MAX_STEPS_IN_FLIGHT:
                self._inflight.pop(0).synchronize()
            self.host_wait_s += time.perf_counter() - t0   # blocked on the GPU, not working (bench: host_busy_ms_per_step)
        # one TTA iteration = one new batch: index structures are built once per iteration (ahead of it by prefetch(), or
        # by the teacher forward) and shared with the student forward, never carried over from an earlier iteration
        pre = self._take_prefetched(batch)
        if batch is None:
            if pre is None:
                raise ValueError("LatteStep.step(None): no batch factory was announced by the previous step / prefetch()")
            batch = pre[4]
        if pre is None and not _KEEP_CACHE:
            batch["lidar"]._lt_cache = None
        if next_batch is not None:
            self.prefetch(next_batch)
        self.optimizer.zero_grad(set_to_none=True)
        losses = self._with_sync_bn(self.forward_backward, batch, pre)
        self._update()
        self._tick("optim_ema")
        if self.params[0].is_cuda:
            ev = torch.cuda.Event()
            ev.record()          # on the main (current) stream: the end of this iteration
            self._inflight.append(ev)
        return losses

    def close(self) -> None:
        """join a helper thread that is still running (end of the adaptation loop)"""
        pre, self._pre = self._pre, None
        if pre is not None:
            pre[1].join()
