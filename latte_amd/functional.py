"""Autograd-aware sparse ops on top of the C-ABI (host-side glue; all arithmetic is in liblatte_hip.so).

Mirrors what torchsparse v1.4.0's torch.autograd.Functions provide to the reference's 3-D backbone
(latte/models/basic_blocks.py, latte/models/torchsparse_utils.py) -- same forward semantics, same gradients.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.autograd import Function

from . import ops

__all__ = ["KernelMap", "get_kernel_offsets", "sparse_conv", "conv_bn", "batch_norm", "spvoxelize", "spdevoxelize",
           "invalidate_packed", "bump_pack_epoch"]

_OFFSET_CACHE: dict = {}


def _ntuple(v, n=3):
    if isinstance(v, (tuple, list)):
        assert len(v) == n
        return tuple(int(x) for x in v)
    return tuple(int(v) for _ in range(n))


def get_kernel_offsets(size, stride=1, dilation=1, device="cpu") -> torch.Tensor:
    """Kernel offsets in torchsparse v1.4.0 order (reference use: latte/models/torchsparse_utils.py:67):
    odd kernel volume -> x fastest, even -> z fastest; values scaled by tensor stride * dilation."""
    size, stride, dilation = _ntuple(size), _ntuple(stride), _ntuple(dilation)
    key = (size, stride, dilation, str(device))
    hit = _OFFSET_CACHE.get(key)
    if hit is not None:
        return hit
    ax = [np.arange(-size[k] // 2 + 1, size[k] // 2 + 1) * stride[k] * dilation[k] for k in range(3)]
    if int(np.prod(size)) % 2 == 1:
        offs = [[x, y, z] for z in ax[2] for y in ax[1] for x in ax[0]]
    else:
        offs = [[x, y, z] for x in ax[0] for y in ax[1] for z in ax[2]]
    t = torch.tensor(offs, dtype=torch.int32, device=device)
    _OFFSET_CACHE[key] = t
    return t


class KernelMap:
    """Kernel map of one (tensor_stride, kernel_size, stride, dilation) conv.

    Product format: dense neighbour table `results` int32 [K, Vout].  torchsparse's list view
    [nbmaps, nbsizes, (Vin, Vout)] is materialised lazily (kmap[0], kmap[1], kmap[2])."""

    def __init__(self, results: torch.Tensor, n_in: int, n_out: int, submanifold: bool):
        self.results = results
        self.n_in, self.n_out = int(n_in), int(n_out)
        self.submanifold = submanifold  # stride 1, odd kernel: results[K-1-k] is its own reverse table
        self._rev = None
        self._pairs = None
        self._order = None
        self._rev_order = None
        self._plan = None
        self._rev_plan = None
        self._npairs = None

    @property
    def K(self):
        return self.results.shape[0]

    def reverse(self) -> torch.Tensor:
        if self._rev is None:
            self._rev = ops.kmap_reverse(self.results, self.n_in)
        return self._rev

    def order(self) -> torch.Tensor:
        """mask-sorted processing order of `results` rows (see lt_kmap_order)."""
        if self._order is None:
            self._order = ops.kmap_order(self.results)
        return self._order

    def reverse_order(self) -> torch.Tensor:
        if self._rev_order is None:
            self._rev_order = ops.kmap_order(self.reverse())
        return self._rev_order

    def plan(self) -> torch.Tensor:
        """conv plan of `results` (built once, shared by every conv on this map)."""
        if self._plan is None:
            self._plan = ops.kmap_plan(self.results)
        return self._plan

    def reverse_plan(self) -> torch.Tensor:
        if self._rev_plan is None:
            self._rev_plan = ops.kmap_plan(self.reverse())
        return self._rev_plan

    def pairs(self):
        if self._pairs is None:
            self._pairs = ops.kmap_pairs(self.results)
        return self._pairs

    def __getitem__(self, i):
        if i == 0:
            return self.pairs()[0]
        if i == 1:
            return self.pairs()[1]
        if i == 2:
            return (self.n_in, self.n_out)
        raise IndexError(i)


def _run_conv(feats, packed, kmap: KernelMap, use_reverse: bool, cout: int):
    """out[o] = sum_k feats[table[k,o]] @ E_k over `results` (or the reverse table)."""
    if ops.CONV_PAIRS is not None:  # bench.py's flop counter (untimed pass)
        if kmap._npairs is None:
            kmap._npairs = (kmap.results >= 0).sum()
        ops.CONV_PAIRS.append(kmap._npairs)
    if feats.shape[1] % 16 == 0:
        n_out = kmap.n_in if use_reverse else kmap.n_out
        plan = kmap.reverse_plan() if use_reverse else kmap.plan()
        return ops.spconv_plan(feats, packed, plan, kmap.K, n_out, cout)
    if use_reverse:
        return ops.spconv(feats, packed, kmap.reverse(), cout, order=kmap.reverse_order())
    return ops.spconv(feats, packed, kmap.results, cout, order=kmap.order())


# Packed-weight cache invalidation.  A conv kernel's MFMA-fragment image is cached per module and re-made when the tag
# (parameter version, data pointer, PACK_EPOCH) changes.  torch bumps `_version` for in-place ops on the parameter itself
# (optimizer steps, load_state_dict's copy_), but NOT for writes through `.data` (the reference's update_ema_all /
# create_ema_model: `ema_param.data[:] = ...`, latte/common/utils/ema_util.py:44-71,74-96; Conv3d.reset_parameters).
# Those are covered by (a) Conv3d.invalidate_packed() in reset_parameters / _load_from_state_dict / _apply and
# invalidate_packed(model) for callers, and (b) PACK_EPOCH: a global optimizer-step hook bumps it whenever an optimizer
# that holds a sparse-conv kernel steps (TRAIN.UPDATE_PARAMS == "all": the EMA write to the teacher's kernels follows that
# step, tools/latte.py:317-326), so every image is re-made once per iteration there and never in the norm-only mode.
PACK_EPOCH = 0
_KERNEL_IDS: dict = {}          # id(kernel parameter) -> weakref, filled by the first conv call on a kernel
_OPT_HOLDS_KERNELS = None       # WeakKeyDictionary optimizer -> (n params, bool)


def bump_pack_epoch() -> None:
    """Declare every cached packed-weight image stale (call after writing conv kernels through `.data`)."""
    global PACK_EPOCH
    PACK_EPOCH += 1


def invalidate_packed(model) -> None:
    """Drop the packed-weight images of every sparse conv of `model`."""
    for m in model.modules():
        inv = getattr(m, "invalidate_packed", None)
        if inv is not None:
            inv()


def _register_kernel(weight) -> None:
    import weakref
    k = id(weight)
    if k not in _KERNEL_IDS:
        _KERNEL_IDS[k] = weakref.ref(weight, lambda _r, k=k: _KERNEL_IDS.pop(k, None))
        _install_optimizer_hook()


def _install_optimizer_hook() -> None:
    global _OPT_HOLDS_KERNELS
    if _OPT_HOLDS_KERNELS is not None:
        return
    import weakref
    from torch.optim.optimizer import register_optimizer_step_post_hook
    _OPT_HOLDS_KERNELS = weakref.WeakKeyDictionary()

    def post_step(opt, args, kwargs):
        n = sum(len(g["params"]) for g in opt.param_groups)
        hit = _OPT_HOLDS_KERNELS.get(opt)
        if hit is None or hit[0] != n:
            holds = any(id(p) in _KERNEL_IDS for g in opt.param_groups for p in g["params"])
            _OPT_HOLDS_KERNELS[opt] = hit = (n, holds)
        if hit[1]:
            bump_pack_epoch()

    register_optimizer_step_post_hook(post_step)


def _packed(weight, mode: int, cache: dict | None):
    """MFMA-fragment image of `weight`, re-packed only when the parameter changed (under TRAIN.UPDATE_PARAMS == "norm"
    the conv kernels are frozen: latte/common/utils/ema_util.py:16-24); see PACK_EPOCH above for what "changed" covers."""
    w = weight if weight.dim() == 3 else weight.unsqueeze(0)
    # effective (Cin, Cout) of the product this image feeds: modes 1, 2 are the transposed (backward-input) matrices
    ce_in, ce_out = (w.shape[1], w.shape[2]) if mode == 0 else (w.shape[2], w.shape[1])
    pmode = mode + 4 if (ops.CONV_PRECISION != "f32" and ops.conv_uses_bf16(ce_in, ce_out)) else mode
    if cache is None:
        return ops.pack_weights(w, pmode)
    tag = (weight._version, weight.data_ptr(), PACK_EPOCH, pmode)
    hit = cache.get(mode)
    if hit is not None and hit[0] == tag:
        return hit[1]
    _register_kernel(weight)
    packed = ops.pack_weights(w, pmode)
    cache[mode] = (tag, packed)
    return packed


class _SparseConv(Function):
    @staticmethod
    def forward(ctx, feats, weight, kmap: KernelMap, transposed: bool, wcache=None):
        w = weight if weight.dim() == 3 else weight.unsqueeze(0)
        cout = w.shape[2]
        ctx.wcache = wcache
        if feats.shape[1] == 4 and cout in (16, 32) and not transposed and feats.is_cuda:
            # the network's first conv: streaming kernel, unpacked weights
            if ops.CONV_PAIRS is not None:
                if kmap._npairs is None:
                    kmap._npairs = (kmap.results >= 0).sum()
                ops.CONV_PAIRS.append(kmap._npairs)
            out = ops.spconv_stem(feats, w, kmap.plan(), kmap.K, kmap.n_out)
        else:
            out = _run_conv(feats, _packed(weight, 0, wcache), kmap, transposed, cout)
        ctx.kmap, ctx.transposed = kmap, transposed
        ctx.save_for_backward(feats, weight)
        return out

    @staticmethod
    def backward(ctx, gout):
        feats, weight = ctx.saved_tensors
        kmap, transposed = ctx.kmap, ctx.transposed
        w = weight if weight.dim() == 3 else weight.unsqueeze(0)
        gout = gout.contiguous()
        gfeats = gweight = None
        if ctx.needs_input_grad[0]:
            cin = w.shape[1]
            if kmap.submanifold and not transposed:
                gfeats = _run_conv(gout, _packed(weight, 2, ctx.wcache), kmap, False, cin)
            elif not transposed:
                gfeats = _run_conv(gout, _packed(weight, 1, ctx.wcache), kmap, True, cin)
            else:
                gfeats = _run_conv(gout, _packed(weight, 1, ctx.wcache), kmap, False, cin)
        if ctx.needs_input_grad[1]:
            # dW is only reached when TRAIN.UPDATE_PARAMS == "all" (latte/common/utils/ema_util.py:44-71); the shipped
            # Latte configs train norm layers only.  lt_spconv_bwd_weight over the pair list of the map.
            nbmaps, nbsizes = kmap.pairs()
            if w.shape[1] % 16 == 0 and w.shape[2] % 16 == 0:
                gw = ops.spconv_bwd_weight(feats, gout, nbmaps, nbsizes, w.shape[0], w.shape[1], w.shape[2], transposed)
            else:
                # the 4-channel stem (and other widths the MFMA tiling does not take): per-offset GEMMs in torch
                sizes = nbsizes.tolist()
                gw = torch.zeros_like(w)
                off = 0
                for k, m in enumerate(sizes):
                    if m == 0:
                        continue
                    p = nbmaps[off:off + m].long()
                    off += m
                    i_in, i_out = (p[:, 1], p[:, 0]) if transposed else (p[:, 0], p[:, 1])
                    gw[k] = feats[i_in].t() @ gout[i_out]
            gweight = gw if weight.dim() == 3 else gw[0]
        return gfeats, gweight, None, None, None


def sparse_conv(feats, weight, kmap: KernelMap, transposed: bool = False, wcache: dict | None = None):
    return _SparseConv.apply(feats, weight, kmap, transposed, wcache)


# Cross-rank BatchNorm statistics (SURVEY 8e "optional exact-parity mode").  None: every rank normalises with the
# statistics of its own scans (the default of the sharded step; a documented deviation from one big batch).  A process
# group: every BatchNorm all-reduces [sum x, sum x^2, rows] in the forward and [sum g, sum g*xhat] in the backward, so a
# batch split over ranks gives the features and gradients of the unsplit batch (to fp32 rounding).  The parameter
# gradients then ARE the global sums on every rank: the step must not all-reduce them again (LatteStep handles that).
SYNC_BN_GROUP = None


def _allreduce_small(t: torch.Tensor, group) -> torch.Tensor:
    """sum all-reduce of a small device tensor (RCCL on GPU groups; through the host for gloo rehearsals)"""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, group=group)
        return h.to(t.device)
    dist.all_reduce(t, group=group)
    return t


_ROWS_SCALARS: dict = {}


def _rows_scalar(rows: int, device) -> torch.Tensor:
    """fp64 [1] device tensor holding a row count, cached per (count, device): the same ~10 counts recur in every layer
    of every step, and a fresh torch.tensor(..., device=) is a host-to-device copy per BatchNorm"""
    key = (int(rows), str(device))
    t = _ROWS_SCALARS.get(key)
    if t is None:
        if len(_ROWS_SCALARS) > 256:
            _ROWS_SCALARS.clear()
        t = _ROWS_SCALARS[key] = torch.tensor([float(rows)], dtype=torch.float64, device=device)
    return t


def sync_batch_stats(mean: torch.Tensor, var: torch.Tensor, rows: int, group):
    """local (mean, biased var, rows) of every rank -> global (mean, var, total rows); combined in fp64 from
    [sum x, sum x^2, rows].  A rank with rows == 0 contributes zeros: every rank issues the same collective whatever its
    local row count is (a shard with an empty coarse level must not leave its peers waiting)."""
    m, v = mean.double(), var.double()
    vec = torch.cat([m * rows, (v + m * m) * rows, _rows_scalar(rows, mean.device)])
    vec = _allreduce_small(vec, group)
    C = mean.numel()
    n = vec[2 * C]
    gm = vec[:C] / n
    gv = (vec[C:2 * C] / n - gm * gm).clamp_(min=0.0)
    return gm.float(), gv.float(), n


class _BatchNorm(Function):
    """BN with batch statistics, optional fused residual add and ReLU: y = relu(bn(x) + residual); post_add (with relu and
    a residual): y = relu(bn(x)) + residual, the sum after the activation (the point branch of SPVCNN)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, eps, relu, stats_out, post_add=False):
        if x.shape[0] > 0:
            mean, var = ops.bn_stats(x)
        else:   # cross-rank statistics only (batch_norm refuses otherwise): this rank has no row at this level
            mean, var = x.new_zeros(x.shape[1]), x.new_zeros(x.shape[1])
        ctx.sync = None
        if SYNC_BN_GROUP is not None:
            mean, var, n_total = sync_batch_stats(mean, var, x.shape[0], SYNC_BN_GROUP)
            ctx.sync = (SYNC_BN_GROUP, n_total, x.shape[0])
        post_add = bool(post_add and relu and residual is not None)
        if x.shape[0] > 0:
            y = ops.bn_apply(x, mean, var, gamma, beta, eps, residual=residual, relu=relu, post_add=post_add)
        else:
            y = torch.empty_like(x)
        ctx.eps, ctx.relu, ctx.post_add = eps, relu, post_add
        ctx.has_res = residual is not None
        # ReLU mask in the backward: from y when a residual was added before the ReLU, otherwise rebuilt from x (y is not
        # read again; with post_add y is not the activation's output, so it cannot serve as the mask)
        ctx.save_for_backward(x, y if (relu and residual is not None and not post_add) else None, mean, var, gamma, beta)
        if stats_out is not None:
            stats_out.append((mean, var))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, var, gamma, beta = ctx.saved_tensors
        dy = dy.contiguous()
        need_dx = ctx.needs_input_grad[0]
        need_dres = ctx.has_res and ctx.needs_input_grad[3]
        sync = None
        if ctx.sync is not None:
            group, n_total, rows = ctx.sync
            if rows == 0:   # no local row: still take part in the backward's collective, with zero sums
                z = torch.zeros(2 * x.shape[1], dtype=torch.float32, device=x.device)
                both = _allreduce_small(z, group)
                C = x.shape[1]
                e = torch.empty_like(x)
                return (e if need_dx else None), both[C:].contiguous(), both[:C].contiguous(), (e if need_dres else None), \
                    None, None, None, None

            def sync(sum_g, sum_gx):
                both = _allreduce_small(torch.cat([sum_g, sum_gx]), group)
                C = sum_g.numel()
                return both[:C].contiguous(), both[C:].contiguous(), (rows / n_total).float()
        if ctx.post_add:
            # the residual bypasses the ReLU: its gradient is dy itself
            dx, dgamma, dbeta, _ = ops.bn_bwd(dy, x, None, mean, var, gamma, ctx.eps, True, need_dx=need_dx,
                                              need_dres=False, beta=beta, sync=sync)
            return dx, dgamma, dbeta, (dy if need_dres else None), None, None, None, None
        dx, dgamma, dbeta, dres = ops.bn_bwd(dy, x, y, mean, var, gamma, ctx.eps, ctx.relu, need_dx=need_dx,
                                             need_dres=need_dres, beta=beta, sync=sync)
        return dx, dgamma, dbeta, dres, None, None, None, None


_DIAG_NO_APPLY = False   # diagnostics (spvcnn._DIAG_SKIP_FIRST_APPLY): composites skip the BatchNorm apply pass, WRONG results


class _ConvBN(Function):
    """y = relu(bn(conv(feats)) + residual) as ONE autograd node over the composite entry points lt_conv_bn_fwd/bwd
    (frozen conv kernel: the norm-only TTA configuration; anything else takes the unfused ops)."""

    @staticmethod
    def forward(ctx, feats, weight, gamma, beta, residual, kmap, transposed, relu, eps, wcache):
        w = weight if weight.dim() == 3 else weight.unsqueeze(0)
        cin, cout = w.shape[1], w.shape[2]
        packed = _packed(weight, 0, wcache)
        if ops.CONV_PAIRS is not None:
            if kmap._npairs is None:
                kmap._npairs = (kmap.results >= 0).sum()
            ops.CONV_PAIRS.append(kmap._npairs)
        n_out = kmap.n_in if transposed else kmap.n_out
        plan = kmap.reverse_plan() if transposed else kmap.plan()
        conv_out, y, mean, var = ops.conv_bn_fwd(feats, packed, plan, kmap.K, n_out, cout, gamma.detach(), beta.detach(),
                                                 eps, residual, relu)
        ctx.kmap, ctx.transposed, ctx.relu, ctx.eps, ctx.wcache = kmap, transposed, relu, eps, wcache
        ctx.cin, ctx.n_in = cin, feats.shape[0]
        ctx.has_res = residual is not None
        ctx.save_for_backward(conv_out, y if (relu and residual is not None) else None, mean, var, gamma, weight, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        conv_out, y, mean, var, gamma, weight, beta = ctx.saved_tensors
        kmap, transposed = ctx.kmap, ctx.transposed
        dy = dy.contiguous()
        packed_bwd = plan_bwd = None
        if ctx.needs_input_grad[0]:
            if ops.CONV_PAIRS is not None:
                ops.CONV_PAIRS.append(kmap._npairs)
            if kmap.submanifold and not transposed:
                packed_bwd, plan_bwd = _packed(weight, 2, ctx.wcache), kmap.plan()
            elif not transposed:
                packed_bwd, plan_bwd = _packed(weight, 1, ctx.wcache), kmap.reverse_plan()
            else:
                packed_bwd, plan_bwd = _packed(weight, 1, ctx.wcache), kmap.plan()
        need_dres = ctx.has_res and ctx.needs_input_grad[4]
        din, dgamma, dbeta, dres = ops.conv_bn_bwd(dy, conv_out, y, mean, var, gamma.detach(), ctx.eps, ctx.relu, need_dres,
                                                   packed_bwd, plan_bwd, kmap.K, ctx.n_in, ctx.cin, beta=beta.detach())
        return din, None, dgamma, dbeta, dres, None, None, None, None, None


def conv_bn(feats, weight, gamma, beta, kmap: KernelMap, transposed=False, residual=None, relu=False, eps=1e-5,
            wcache=None):
    if not torch.is_grad_enabled():
        # the teacher forward (latte/tta/tools/latte.py:156-159 runs it under no_grad): no autograd node, nothing saved,
        # BatchNorm applied in place on the conv output
        w = weight if weight.dim() == 3 else weight.unsqueeze(0)
        if ops.CONV_PAIRS is not None:
            if kmap._npairs is None:
                kmap._npairs = (kmap.results >= 0).sum()
            ops.CONV_PAIRS.append(kmap._npairs)
        n_out = kmap.n_in if transposed else kmap.n_out
        plan = kmap.reverse_plan() if transposed else kmap.plan()
        return ops.conv_bn_fwd(feats.contiguous(), _packed(weight, 0, wcache), plan, kmap.K, n_out, w.shape[2], gamma.detach(),
                               beta.detach(), float(eps), residual, bool(relu), keep_conv_out=False)[1]
    return _ConvBN.apply(feats.contiguous(), weight, gamma, beta, residual, kmap, bool(transposed), bool(relu),
                         float(eps), wcache)


def batch_norm(x, gamma, beta, eps=1e-5, residual=None, relu=False, stats_out=None, post_add=False):
    if x.shape[0] <= 1 and SYNC_BN_GROUP is None:
        # same refusal as torch.nn.functional.batch_norm in training mode; with cross-rank statistics the LOCAL row count
        # says nothing (and every rank must reach the collective), so the call goes through
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
    return _BatchNorm.apply(x, gamma, beta, residual, float(eps), bool(relu), stats_out, bool(post_add))


class _GatherRows(Function):
    """x[idx] (rows).  torch's own backward of advanced indexing is index_put_(accumulate=True): a sort of the indices
    plus a segmented kernel, ~170 us at 209 k x 16; an index_add_ into zeros is ~15 us (float atomics: the order of the
    2-3 addends of a row is not fixed -- the same run-to-run noise the step's other index_adds already have)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.n = x.shape[0]
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        out = torch.zeros((ctx.n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        out.index_add_(0, idx, g.contiguous())
        return out, None


class _HeadGather(Function):
    """Both segmentation heads evaluated at the raw points: out_h = feats[idx] @ W_h^T + b_h (lt_head_gather_fwd).
    Backward: the point gradients are summed per feature row over a CSR of idx (atomic-free, ascending point order:
    lt_segment_reduce), the rest is two small dense products."""

    @staticmethod
    def forward(ctx, feats, w1, b1, w2, b2, idx, csr, packed, bias):
        c1 = w1.shape[0]
        c2 = w2.shape[0] if w2 is not None else 0
        out1, out2 = ops.head_gather_fwd(feats, idx, packed, bias, c1, c2)
        ctx.save_for_backward(feats, w1, w2, idx)
        ctx.csr = csr
        ctx.set_materialize_grads(False)   # a head the loss does not use (seg_logit2 in Latte) costs nothing in the backward
        if out2 is None:
            ctx.mark_non_differentiable()
            return out1, feats.new_zeros(0)
        return out1, out2

    @staticmethod
    def backward(ctx, g1, g2):
        feats, w1, w2, idx = ctx.saved_tensors
        V = feats.shape[0]
        csr = ctx.csr() if callable(ctx.csr) else ctx.csr
        if csr is None:
            csr = ops.csr_build(idx.to(torch.int32), V)
        perm, off = csr
        gs1 = ops.segment_reduce(g1.contiguous(), perm, off, "sum") if g1 is not None else None    # [V, C1]
        gs2 = ops.segment_reduce(g2.contiguous(), perm, off, "sum") if (g2 is not None and w2 is not None) else None
        gfeats = gw1 = gb1 = gw2 = gb2 = None
        if ctx.needs_input_grad[0]:
            if gs1 is not None:
                gfeats = gs1 @ w1
            if gs2 is not None:
                gfeats = gs2 @ w2 if gfeats is None else torch.addmm(gfeats, gs2, w2)
            if gfeats is None:
                gfeats = torch.zeros_like(feats)
        if gs1 is not None:
            if ctx.needs_input_grad[1]:
                gw1 = gs1.t() @ feats
            if ctx.needs_input_grad[2]:
                gb1 = gs1.sum(0)
        if gs2 is not None:
            if ctx.needs_input_grad[3]:
                gw2 = gs2.t() @ feats
            if ctx.needs_input_grad[4]:
                gb2 = gs2.sum(0)
        return gfeats, gw1, gb1, gw2, gb2, None, None, None, None


def head_gather(feats, w1, b1, w2, b2, idx, csr, packed, bias):
    """-> (logits of head 1 [N,C], logits of head 2 [N,C] | None) at the rows feats[idx]; see _HeadGather."""
    o1, o2 = _HeadGather.apply(feats.contiguous(), w1, b1, w2, b2, idx, csr, packed, bias)
    return o1, (o2 if w2 is not None else None)


def gather_rows(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """x[idx] for an int64 row index vector, with a cheap backward."""
    if not (x.requires_grad and torch.is_grad_enabled()):
        return x[idx]
    return _GatherRows.apply(x, idx)


class _Voxelize(Function):
    @staticmethod
    def forward(ctx, feats, idx, counts, csr):
        ctx.n = feats.shape[0]
        ctx.save_for_backward(idx, counts)
        if csr is not None and feats.shape[1] % 4 == 0:
            return ops.voxelize_csr_fwd(feats, csr, counts.numel())
        return ops.voxelize_fwd(feats, idx, counts)

    @staticmethod
    def backward(ctx, gout):
        idx, counts = ctx.saved_tensors
        return ops.voxelize_bwd(gout.contiguous(), idx, counts, ctx.n), None, None, None


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous()


def spvoxelize(feats, idx, counts, csr=None):
    """F.spvoxelize (latte/models/torchsparse_utils.py:20,23,54); idx any int dtype, counts int32.  csr (optional):
    ops.csr_build(idx, V) of this map -> the atomic-free kernel."""
    return _Voxelize.apply(feats.contiguous(), _i32(idx), _i32(counts), csr)


class _Devoxelize(Function):
    @staticmethod
    def forward(ctx, feats, idx, w, csr):
        ctx.v = feats.shape[0]
        ctx.csr = csr
        ctx.save_for_backward(idx, w)
        return ops.devoxelize_fwd(feats, idx, w)

    @staticmethod
    def backward(ctx, gout):
        idx, w = ctx.saved_tensors
        csr = ctx.csr() if callable(ctx.csr) else ctx.csr
        if csr is not None and gout.shape[1] % 4 == 0:
            if len(csr) == 3:   # (cells, point->voxel CSR, point->voxel index): no CSR of this map needed
                return ops.devoxelize_cell_bwd(gout.contiguous(), csr[0], csr[1], csr[2], idx, w), None, None, None
            return ops.devoxelize_csr_bwd(gout.contiguous(), csr, w, ctx.v), None, None, None
        return ops.devoxelize_bwd(gout.contiguous(), idx, w, ctx.v), None, None, None


def spdevoxelize(feats, idx, w, csr=None):
    """F.spdevoxelize (latte/models/torchsparse_utils.py:81,93); idx [N,8] any int dtype, w [N,8] fp32.  csr (optional):
    ops.csr_build(idx.view(-1), V) of this map, or (ops.devoxelize_cells table, the point->voxel CSR of the level, the
    point->voxel index), or a callable returning either -> an atomic-free backward kernel."""
    return _Devoxelize.apply(feats.contiguous(), _i32(idx), w.contiguous(), csr)
