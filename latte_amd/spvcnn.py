"""SPVCNN 3-D U-Net backbone + Net3DSeg head on the MI355X sparse ops.

Host-side mirror of the reference's module API (latte/models/spvcnn.py:16-165, latte/models/basic_blocks.py:13-76,
latte/models/xmuda_arch.py:289-342, latte/data/collate.py:34-55): same constructor arguments, same module tree and
parameter names (so reference checkpoints load with load_state_dict), same forward outputs.  The graph is described
by a table instead of hand-written layers, and conv -> BN -> ReLU (+ residual) chains call the fused HIP
BN/ReLU/residual kernel instead of three separate modules."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import sparse_modules as spnn
from .point_voxel import initial_voxelize, p2v_index, point_to_voxel, v2p_cells, v2p_index, voxel_to_point
from .tensor import PointTensor, SparseTensor, cat

__all__ = ["SPVCNN", "Net3DSeg", "BasicConvolutionBlock", "BasicDeconvolutionBlock", "ResidualBlock",
           "inverse_to_all"]


# tests only: when a dict, every conv -> BN (+residual) (+ReLU) chain and every point-branch BN + ReLU stores its output
# under id(BatchNorm module) (tests/test_gpu_spvcnn.py reads the ReLU activity pattern of a forward from it)
ACT_TRACE = None


def _conv_bn(seq, x: SparseTensor, i: int, relu: bool, residual=None) -> SparseTensor:
    y = _conv_bn_impl(seq, x, i, relu, residual)
    if ACT_TRACE is not None:
        ACT_TRACE[id(seq[i + 1])] = y.feats.detach()
    return y


def _conv_bn_impl(seq, x: SparseTensor, i: int, relu: bool, residual=None) -> SparseTensor:
    """seq[i] = Conv3d, seq[i+1] = BatchNorm.  Hot case (frozen kernel, batch statistics, Cin % 16 == 0, on the GPU): one
    composite call -- conv with the BN statistics in its epilogue, finalize, BN(+residual)(+ReLU).  Otherwise conv, then
    the fused BN(+residual)(+ReLU) kernel."""
    conv, bn = seq[i], seq[i + 1]
    batch_stats = bn.training or not bn.track_running_stats or bn.running_mean is None
    updates_running = bn.training and bn.track_running_stats and bn.running_mean is not None
    from . import functional as LF
    if (conv.kernel_volume > 1 and conv.bias is None and x.feats.is_cuda and x.feats.shape[1] % 16 == 0
            and batch_stats and not updates_running and not conv.kernel.requires_grad and x.feats.shape[0] > 1
            and LF.SYNC_BN_GROUP is None):   # cross-rank statistics: the all-reduce sits between conv and BN apply
        y = spnn.conv3d_bn(x, conv, bn, relu=relu, residual=residual)
        if y is not None:
            return y
    y = conv(x)
    y.feats = bn.bn_feats(y.feats, residual=residual, relu=relu)
    return y


# diagnostics (tools only, WRONG results): the first conv of every residual block runs conv + statistics but NOT the BatchNorm
# apply pass -- what a consumer that normalises on load would leave of that layer: the measured upper bound of that fusion
_DIAG_SKIP_FIRST_APPLY = bool(__import__("os").environ.get("LATTE_DIAG_SKIP_FIRST_APPLY"))


def _conv_stats_only(seq, x: SparseTensor) -> SparseTensor:
    from . import functional as LF
    conv, bn = seq[0], seq[1]
    kmap, out_coords = spnn.build_kernel_map(x, conv.kernel_size, conv.stride, conv.dilation)
    LF._DIAG_NO_APPLY = True
    try:
        out = LF.conv_bn(x.feats, conv.kernel, bn.weight, bn.bias, kmap, False, residual=None, relu=True, eps=bn.eps,
                         wcache=conv._packed_cache)
    finally:
        LF._DIAG_NO_APPLY = False
    y = SparseTensor(out, out_coords, x.stride)
    y.cmaps, y.kmaps = x.cmaps, x.kmaps
    return y


class BasicConvolutionBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride),
                                 spnn.BatchNorm(outc), spnn.ReLU(True))

    def forward(self, x):
        return _conv_bn(self.net, x, 0, relu=True)


class BasicDeconvolutionBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1):
        super().__init__()
        self.net = nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=ks, stride=stride, transposed=True),
                                 spnn.BatchNorm(outc), spnn.ReLU(True))

    def forward(self, x):
        return _conv_bn(self.net, x, 0, relu=True)


class ResidualBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1):
        super().__init__()
        self.net = nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=ks, dilation=dilation, stride=stride),
                                 spnn.BatchNorm(outc), spnn.ReLU(True),
                                 spnn.Conv3d(outc, outc, kernel_size=ks, dilation=dilation, stride=1),
                                 spnn.BatchNorm(outc))
        if inc == outc and stride == 1:
            self.downsample = nn.Sequential()
        else:
            self.downsample = nn.Sequential(spnn.Conv3d(inc, outc, kernel_size=1, dilation=1, stride=stride),
                                            spnn.BatchNorm(outc))
        self.ReLU = spnn.ReLU(True)

    def forward(self, x):
        if len(self.downsample) == 0:
            shortcut = x.feats
        else:
            shortcut = _conv_bn(self.downsample, x, 0, relu=False).feats
        if _DIAG_SKIP_FIRST_APPLY and x.feats.is_cuda:
            y = _conv_stats_only(self.net, x)   # upper bound of what normalise-on-load could save (WRONG results)
        else:
            y = _conv_bn(self.net, x, 0, relu=True)
        # relu(bn(conv(y)) + shortcut) in one kernel
        return _conv_bn(self.net, y, 3, relu=True, residual=shortcut)


_BASE_CHANNELS = (32, 32, 64, 128, 256, 256, 128, 96, 96)


class SPVCNN(nn.Module):
    """Point-voxel U-Net: stem, 4 down stages (2^3 s2 conv + 2 residual blocks), 4 up stages (2^3 s2 transposed
    conv, skip concat, 2 residual blocks), 3 point-MLP branches.  kwargs: cr (channel ratio, default 1.0)."""

    def __init__(self, **kwargs):
        super().__init__()
        cr = kwargs.get("cr", 1.0)
        cs = [int(cr * c) for c in _BASE_CHANNELS]
        if any(c < 16 or c % 16 for c in cs):
            # the MFMA conv kernels tile channels by 16 (csrc/spconv.hip); the reference's configs use cr = 1.0 (0.5 in its
            # smoke test, latte/models/xmuda_arch.py:485-599)
            raise ValueError(f"SPVCNN(cr={cr}): channel widths {cs} must be multiples of 16 on this backend "
                             "(cr = 1.0 and 0.5 are; 0.25 or 0.75 are not)")
        self.pres = self.vres = 0.05
        self.out_channels = cs[8]

        self.stem = nn.Sequential(spnn.Conv3d(4, cs[0], kernel_size=3, stride=1), spnn.BatchNorm(cs[0]),
                                  spnn.ReLU(True), spnn.Conv3d(cs[0], cs[0], kernel_size=3, stride=1),
                                  spnn.BatchNorm(cs[0]), spnn.ReLU(True))
        # encoder: (in, out) per stage
        for n, (ci, co) in enumerate([(cs[0], cs[1]), (cs[1], cs[2]), (cs[2], cs[3]), (cs[3], cs[4])], start=1):
            setattr(self, f"stage{n}", nn.Sequential(
                BasicConvolutionBlock(ci, ci, ks=2, stride=2, dilation=1),
                ResidualBlock(ci, co, ks=3, stride=1, dilation=1),
                ResidualBlock(co, co, ks=3, stride=1, dilation=1)))
        # decoder: (in, out, skip channels)
        for n, (ci, co, sk) in enumerate([(cs[4], cs[5], cs[3]), (cs[5], cs[6], cs[2]), (cs[6], cs[7], cs[1]),
                                          (cs[7], cs[8], cs[0])], start=1):
            setattr(self, f"up{n}", nn.ModuleList([
                BasicDeconvolutionBlock(ci, co, ks=2, stride=2),
                nn.Sequential(ResidualBlock(co + sk, co, ks=3, stride=1, dilation=1),
                              ResidualBlock(co, co, ks=3, stride=1, dilation=1))]))
        self.point_transforms = nn.ModuleList([
            nn.Sequential(nn.Linear(a, b), spnn.BatchNorm(b), nn.ReLU(True))  # BatchNorm1d subclass: same state_dict keys
            for a, b in [(cs[0], cs[4]), (cs[4], cs[6]), (cs[6], cs[8])]])
        self.weight_initialization()
        self.dropout = nn.Dropout(0.3, True)

    def weight_initialization(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stem(self, x):
        x = _conv_bn(self.stem, x, 0, relu=True)
        return _conv_bn(self.stem, x, 3, relu=True)

    def _pt(self, i: int, f: torch.Tensor, add_to: torch.Tensor) -> torch.Tensor:
        """add_to + point_transforms[i](f), point_transforms[i] = Linear -> BatchNorm1d -> ReLU
        (latte/models/spvcnn.py:95-111); the GEMM stays in torch/hipBLASLt, BN(batch stats) + ReLU + the sum with the
        devoxelized features is one HIP kernel."""
        seq = self.point_transforms[i]
        if not f.is_cuda:
            return add_to + seq(f)
        if ACT_TRACE is not None:
            t = seq[1].bn_feats(seq[0](f), relu=True)
            ACT_TRACE[id(seq[1])] = t.detach()
            return add_to + t
        return seq[1].bn_feats(seq[0](f), residual=add_to, relu=True, post_add=True)

    def _up(self, up, y, skip):
        y = up[0](y)
        return up[1](cat([y, skip]))

    @staticmethod
    def _batch_cache(x: SparseTensor):
        """index structures (voxel sets, point<->voxel maps, kernel maps, conv plans) depend on x.C only: they are built
        by the first pass over a batch and reused by later ones (teacher then student see the same `lidar`)"""
        cache = getattr(x, "_lt_cache", None)
        if cache is None or cache.get("coords_ptr") != (x.C.data_ptr(), tuple(x.C.shape)):
            cache = {"coords_ptr": (x.C.data_ptr(), tuple(x.C.shape)), "point": None}
            x._lt_cache = cache
        z = PointTensor(x.F, x.C.float())
        if cache["point"] is None:
            cache["point"] = (z.idx_query, z.weights, z.additional_features)
        else:
            z.idx_query, z.weights, z.additional_features = cache["point"]
        return cache, z

    @torch.no_grad()
    def prepare(self, x: SparseTensor, backward: bool = True) -> None:
        """Build every index structure the forward AND backward over this batch will need (no feature work): the voxel
        set of each level, the 3^3 map and plan of each level, the 2^3 stride-2 maps with forward and reverse plans, the
        point<->voxel maps of strides 1, 4 and 16.  After this, forwards over the batch only launch feature kernels and
        may run concurrently on different streams (LatteStep runs teacher and student side by side)."""
        cache, z = self._batch_cache(x)
        if cache.get("prepared"):
            return
        t = initial_voxelize(z, self.pres, self.vres, cache)
        levels = {1: t}
        for s in (1, 2, 4, 8):
            km3, _ = spnn.build_kernel_map(t, 3, 1, 1)
            km3.plan()
            km2, oc = spnn.build_kernel_map(t, 2, 2, 1)
            km2.plan()
            km2.reverse_plan()
            nt = SparseTensor(t.feats[:0], oc, 2 * s)
            nt.cmaps, nt.kmaps = t.cmaps, t.kmaps
            nt.cmaps.setdefault(nt.stride, oc)
            t = levels[2 * s] = nt
        spnn.build_kernel_map(t, 3, 1, 1)[0].plan()
        for s in (1, 16, 4):  # the order the forward asks for them
            if z.idx_query.get(levels[s].s) is None:
                z.idx_query[levels[s].s], z.weights[levels[s].s] = v2p_index(levels[s], z)
            p2v_index(levels[s], z)
            if backward:
                v2p_cells(levels[s], z)   # one [V, 8] table per level: the devoxelize backward gathers over the p2v CSR
        if x.feats.is_cuda and min(int(levels[s].C.shape[0]) for s in levels) >= 2:
            # what the one-call pass (netprog / lt_net_forward) takes: every structure above as one struct of pointers
            from . import netprog
            cache["levels"] = levels
            cache["net_maps"] = netprog.build_maps(cache, z, levels)
        cache["prepared"] = True

    def _program(self):
        prog = self.__dict__.get("_lt_prog")
        if prog is None:
            from . import netprog
            prog = self.__dict__["_lt_prog"] = netprog.NetProgram(self)
        return prog

    def forward(self, x: SparseTensor) -> torch.Tensor:
        if x.feats.is_cuda:
            # the shipped TTA configuration (frozen kernels, batch statistics, no dropout): the whole pass is one call into
            # the library (netprog.py / lt_net_forward); everything else walks the module tree below
            prog = self._program()
            if prog.eligible(x.feats):
                from . import netprog
                need_bwd = netprog.wants_grad(prog)
                self.prepare(x, backward=need_bwd)
                cache, z = self._batch_cache(x)
                maps = cache.get("net_maps")
                if maps is not None and need_bwd and not maps.struct.v2p_cells[0]:
                    # the batch was prepared for inference only: add what the backward reads
                    for s in (1, 16, 4):
                        v2p_cells(cache["levels"][s], z)
                    maps = cache["net_maps"] = netprog.build_maps(cache, z, cache["levels"])
                if maps is not None:
                    return netprog.run(prog, x.feats, maps, cache["levels"])
        return self._forward_modules(x)

    def _forward_modules(self, x: SparseTensor) -> torch.Tensor:
        cache, z = self._batch_cache(x)
        x0 = self._stem(initial_voxelize(z, self.pres, self.vres, cache))
        z0 = voxel_to_point(x0, z, nearest=False)

        x1 = self.stage1(point_to_voxel(x0, z0))
        x2 = self.stage2(x1)
        x3 = self.stage3(x2)
        x4 = self.stage4(x3)
        z1 = voxel_to_point(x4, z0)
        z1.F = self._pt(0, z0.F, z1.F)

        y1 = point_to_voxel(x4, z1)
        y1.F = self.dropout(y1.F)
        y1 = self._up(self.up1, y1, x3)
        y2 = self._up(self.up2, y1, x2)
        z2 = voxel_to_point(y2, z1)
        z2.F = self._pt(1, z1.F, z2.F)

        y3 = point_to_voxel(y2, z2)
        y3.F = self.dropout(y3.F)
        y3 = self._up(self.up3, y3, x1)
        y4 = self._up(self.up4, y3, x0)
        z3 = voxel_to_point(y4, z2)
        z3.F = self._pt(2, z2.F, z3.F)
        return z3.F


class Net3DSeg(nn.Module):
    """SPVCNN + linear segmentation head(s); forward(data_batch) -> {'feats','seg_logit'[,'seg_logit2']}."""

    def __init__(self, num_classes, dual_head, backbone_3d, backbone_3d_kwargs, da_method=None, pretrained=False):
        super().__init__()
        if backbone_3d != "SPVCNN":
            raise NotImplementedError(f"3D backbone {backbone_3d} not supported")
        if pretrained:
            raise NotImplementedError("pretrained=True loads a checkpoint that is not part of this repository")
        self.backbone_3d = backbone_3d
        self.net_3d = SPVCNN(**backbone_3d_kwargs)
        self.linear = nn.Linear(self.net_3d.out_channels, num_classes)
        self.dual_head = dual_head
        if dual_head:
            self.linear2 = nn.Linear(self.net_3d.out_channels, num_classes)
        self.da_method = da_method

    def forward(self, data_batch, reverse=False):
        feats = self.net_3d(data_batch["lidar"])
        preds = {"feats": feats, "seg_logit": self.linear(feats)}
        if self.dual_head:
            preds["seg_logit2"] = self.linear2(feats)
        return preds

    def _packed_heads(self):
        """MFMA-fragment image of E = [W1^T | W2^T | 0] and the padded bias, re-made when a head parameter changes (same
        invalidation rules as the conv kernels: functional.PACK_EPOCH)."""
        from . import functional as LF
        from . import ops
        heads = [self.linear] + ([self.linear2] if self.dual_head else [])
        tag = tuple((h.weight._version, h.weight.data_ptr(), h.bias._version, h.bias.data_ptr()) for h in heads) + (LF.PACK_EPOCH,)
        hit = getattr(self, "_head_cache", None)
        if hit is not None and hit[0] == tag:
            return hit[1], hit[2]
        with torch.no_grad():
            c = sum(h.weight.shape[0] for h in heads)
            width = (c + 15) // 16 * 16
            E = torch.zeros(self.net_3d.out_channels, width, device=self.linear.weight.device)
            bias = torch.zeros(width, device=E.device)
            o = 0
            for h in heads:
                n = h.weight.shape[0]
                E[:, o:o + n] = h.weight.t()
                bias[o:o + n] = h.bias
                o += n
            packed = ops.pack_weights(E.unsqueeze(0).contiguous(), 0)
        self._head_cache = (tag, packed, bias)
        return packed, bias

    def forward_points(self, data_batch, inv_all: torch.Tensor, csr=None):
        """forward() followed by inverse_to_all (latte/data/collate.py:34-55) on the logits, as one kernel: the heads are
        evaluated at the raw points, out[p] = head(feats[inv_all[p]]) (lt_head_gather_fwd).  inv_all: int64 [N] row of
        every raw point in the batch's unique-voxel numbering (LatteStep._batch_tables); csr (optional): ops.csr_build of
        it, for the backward.  -> {'feats' [V,C'], 'seg_logit' [N,C], 'seg_logit2' [N,C]}"""
        from . import functional as LF
        feats = self.net_3d(data_batch["lidar"])
        if (not feats.is_cuda or feats.shape[1] % 16 != 0 or feats.shape[1] > 128
                or self.linear.weight.shape[0] * (2 if self.dual_head else 1) > 64):
            out = {"feats": feats, "seg_logit": LF.gather_rows(self.linear(feats), inv_all)}
            if self.dual_head:
                out["seg_logit2"] = LF.gather_rows(self.linear2(feats), inv_all)
            return out
        packed, bias = self._packed_heads()
        l2 = self.linear2 if self.dual_head else None
        o1, o2 = LF.head_gather(feats, self.linear.weight, self.linear.bias, l2.weight if l2 is not None else None,
                                l2.bias if l2 is not None else None, inv_all, csr, packed, bias)
        out = {"feats": feats, "seg_logit": o1}
        if self.dual_head:
            out["seg_logit2"] = o2
        return out


def inverse_to_all(seg_logit: torch.Tensor, data_batch) -> torch.Tensor:
    """Unique-voxel rows -> all raw points (latte/data/collate.py:34-55): one gather with a batch-offset index
    instead of a Python loop of per-scan gathers."""
    sizes = [int(i.shape[0]) for i in data_batch["indices"]]
    idx, off = [], 0
    for n, inv in zip(sizes, data_batch["inverse_map"]):
        inv = inv if torch.is_tensor(inv) else torch.as_tensor(inv)
        idx.append(inv.to(seg_logit.device, dtype=torch.int64) + off)
        off += n
    return seg_logit[torch.cat(idx, 0)]
