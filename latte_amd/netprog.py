"""The SPVCNN backbone pass as ONE call into the library (lt_net_forward / lt_net_backward, include/latte_hip_net.h).

Reference: latte/models/spvcnn.py:113-165 (forward), latte/models/basic_blocks.py:13-76, latte/models/torchsparse_utils.py.
Module by module a pass costs ~5 ms of Python (49 conv+BN composites at ~80 us of wrappers each): three passes per TTA
iteration, the whole step on a one-scan batch.  Here the network is described once as an op list over numbered buffers
(`NetProgram`, built from the module tree, so reference checkpoints and the module API are untouched), the batch's index
structures are handed over as one struct, and a pass is one ctypes call that releases the GIL while it issues ~300 kernels.

Used by SPVCNN.forward when the configuration is the one Latte ships (frozen conv / linear weights, every BatchNorm on
batch statistics, dropout off: latte/common/utils/ema_util.py:6-41); anything else takes the module-by-module path.
The 1x1x1 shortcut convs and the point-branch Linears run through the conv kernel over an identity plan here (torch /
hipBLASLt GEMMs on the module path): same mathematics, fp32 rounding differs (the Linear's bias is dropped: BatchNorm on
batch statistics cancels a per-channel constant exactly)."""
from __future__ import annotations

import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib
from . import functional as LF
from . import ops

from ._lib import I32, I64, F32, P, NetOp, NetBuf, NetLayer, NetProg, NetMaps  # noqa: F401 (the binding lives in _lib.py)

VOX, DEVOX, CONV_BN, CAT = 1, 2, 3, 4
MAP_K3, MAP_DOWN, MAP_UP, MAP_DENSE, MAP_STEM = 0, 1, 2, 3, 4
POINTS = 5

ENABLED = os.environ.get("LATTE_NET_EXEC", "1") != "0"   # A/B switch: 0 = always the module-by-module path


def identity_plan(rows: int, device) -> torch.Tensor:
    """lt_kmap_plan_identity: the K = 1 plan a dense layer over `rows` rows runs on."""
    lib = _lib.load()
    plan = torch.empty(max(int(lib.lt_kmap_plan_bytes(1, int(rows))), 256), dtype=torch.uint8, device=device)
    _lib.call("lt_kmap_plan_identity", int(rows), ops._p(plan), plan.numel(), ops._stream())
    return plan


class BatchMaps:
    """lt_net_maps_t of one batch + the tensors it points into."""

    def __init__(self, struct, keep, n_points, n_vox):
        self.struct, self.keep, self.n_points, self.n_vox = struct, keep, n_points, n_vox


def build_maps(cache: dict, z, levels: dict) -> BatchMaps:
    """From what SPVCNN.prepare() left on the batch: levels {tensor stride: SparseTensor}, z the PointTensor."""
    m = NetMaps()
    keep = []

    def ptr(t):
        keep.append(t)
        return t.data_ptr()

    dev = z.F.device
    m.n_points = z.F.shape[0]
    strides = (1, 2, 4, 8, 16)
    kmaps = levels[1].kmaps
    for l, s in enumerate(strides):
        t = levels[s]
        m.n_vox[l] = t.C.shape[0]
        s3 = (s, s, s)
        m.plan_k3[l] = ptr(kmaps[(s3, (3, 3, 3), (1, 1, 1), (1, 1, 1))].plan())
        if l < 4:
            km = kmaps[(s3, (2, 2, 2), (2, 2, 2), (1, 1, 1))]
            m.plan_down[l] = ptr(km.plan())
            m.plan_up[l] = ptr(km.reverse_plan())
        m.plan_dense[l] = ptr(identity_plan(t.C.shape[0], dev))
    m.plan_dense[POINTS] = ptr(identity_plan(z.F.shape[0], dev))
    af = z.additional_features
    for l, s in ((0, 1), (2, 4), (4, 16)):
        s3 = (s, s, s)
        perm, off = af["p2v_csr"][s3]
        m.p2v_perm[l], m.p2v_off[l] = ptr(perm), ptr(off)
        m.p2v_idx[l] = ptr(LF._i32(af["idx_query"][s3]))
        m.p2v_counts[l] = ptr(LF._i32(af["counts"][s3]))
        m.v2p_idx[l] = ptr(LF._i32(z.idx_query[s3]))
        m.v2p_w[l] = ptr(z.weights[s3].contiguous())
        cells = af.get("v2p_cells", {}).get(s3)
        if cells is not None:
            m.v2p_cells[l] = ptr(cells)
    return BatchMaps(m, keep, int(m.n_points), [int(v) for v in m.n_vox])


def _none():
    return None


class NetProgram:
    """Op list of one SPVCNN instance (built once; the layer table's pointers are refreshed per call)."""

    # a copy / pickle of the model (ema_util.create_ema_model deep-copies the student) does not carry the program: it
    # holds raw pointers into THIS model's parameters; the copy builds its own on first use
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_none, ())

    def __init__(self, net):
        self.net = net
        bufs, oplist, mods = [], [], []

        def buf(rows, ch):
            bufs.append((rows, ch))
            return len(bufs) - 1

        def op(kind, in0, out, in1=-1, layer=-1, mp=0, level=0, relu=0):
            oplist.append((kind, in0, in1, out, layer, mp, level, relu))
            return out

        def vox(x, level):
            return op(VOX, x, buf(level, bufs[x][1]), level=level)

        def devox(x, level):
            return op(DEVOX, x, buf(POINTS, bufs[x][1]), level=level)

        def conv_bn(x, conv, bn, mp, level, relu, res=-1, out_rows=None):
            cout = conv.out_features if isinstance(conv, torch.nn.Linear) else conv.out_channels
            mods.append((conv, bn, mp))
            return op(CONV_BN, x, buf(level if out_rows is None else out_rows, cout), in1=res, layer=len(mods) - 1, mp=mp,
                      level=level, relu=relu)

        def resblock(x, blk, level):
            sc = x if len(blk.downsample) == 0 else conv_bn(x, blk.downsample[0], blk.downsample[1], MAP_DENSE, level, 0)
            y = conv_bn(x, blk.net[0], blk.net[1], MAP_K3, level, 1)
            return conv_bn(y, blk.net[3], blk.net[4], MAP_K3, level, 1, res=sc)

        def stage(x, st, level):    # level: the input's
            x = conv_bn(x, st[0].net[0], st[0].net[1], MAP_DOWN, level, 1, out_rows=level + 1)
            return resblock(resblock(x, st[1], level + 1), st[2], level + 1)

        def up(y, upm, skip, level):  # level: the output's
            y = conv_bn(y, upm[0].net[0], upm[0].net[1], MAP_UP, level, 1)
            c = op(CAT, y, buf(level, bufs[y][1] + bufs[skip][1]), in1=skip)
            return resblock(resblock(c, upm[1][0], level), upm[1][1], level)

        def point_branch(i, f, add_to):
            seq = net.point_transforms[i]
            return conv_bn(f, seq[0], seq[1], MAP_DENSE, POINTS, 3, res=add_to)

        self.in_buf = buf(POINTS, 4)
        x = vox(self.in_buf, 0)
        x = conv_bn(x, net.stem[0], net.stem[1], MAP_STEM, 0, 1)
        x0 = conv_bn(x, net.stem[3], net.stem[4], MAP_K3, 0, 1)
        z0 = devox(x0, 0)
        x1 = stage(vox(z0, 0), net.stage1, 0)
        x2 = stage(x1, net.stage2, 1)
        x3 = stage(x2, net.stage3, 2)
        x4 = stage(x3, net.stage4, 3)
        z1 = point_branch(0, z0, devox(x4, 4))
        y1 = up(vox(z1, 4), net.up1, x3, 3)
        y2 = up(y1, net.up2, x2, 2)
        z2 = point_branch(1, z1, devox(y2, 2))
        y3 = up(vox(z2, 2), net.up3, x1, 1)
        y4 = up(y3, net.up4, x0, 0)
        z3 = point_branch(2, z2, devox(y4, 0))
        self.out_buf = z3

        self.mods = mods
        self.bufs_py = bufs
        self.n_bufs, self.n_ops, self.n_layers = len(bufs), len(oplist), len(mods)
        self._ops = (NetOp * self.n_ops)(*[NetOp(*o) for o in oplist])
        self._bufs = (NetBuf * self.n_bufs)(*[NetBuf(r, c) for r, c in bufs])
        self._layers = (NetLayer * self.n_layers)()
        off = 0
        self.param_slices = []   # per layer: (offset of dgamma, offset of dbeta, cout)
        for i, (conv, bn, mp) in enumerate(mods):
            L = self._layers[i]
            if isinstance(conv, torch.nn.Linear):
                L.cin, L.cout, L.K = conv.in_features, conv.out_features, 1
            else:
                L.cin, L.cout, L.K = conv.in_channels, conv.out_channels, conv.kernel_volume
            L.eps = bn.eps
            L.dgamma_off, L.dbeta_off = off, off + L.cout
            self.param_slices.append((off, off + L.cout, L.cout))
            off += 2 * L.cout
        self.n_dparams = off
        self.prog = NetProg(self._ops, self.n_ops, self._bufs, self.n_bufs, self._layers, self.n_layers, self.out_buf)
        self._ext = (P * self.n_bufs)()
        self._off = (I64 * self.n_bufs)()
        self._keep = []
        # conv launches of a pass in launch order (bench.py's flop counter): (map kind, level) of every conv-kernel launch
        self.fwd_convs = [(o[5], o[6]) for o in oplist if o[0] == CONV_BN]

    # ------------------------------------------------------------------ eligibility
    def eligible(self, feats: torch.Tensor) -> bool:
        """the configuration lt_net_* implements: frozen kernels, batch statistics everywhere, dropout off, one rank's
        own statistics; tests that trace activations take the module path"""
        from . import spvcnn as M
        if not (ENABLED and feats.is_cuda and LF.SYNC_BN_GROUP is None and M.ACT_TRACE is None
                and not M._DIAG_SKIP_FIRST_APPLY and ops.CONV_VARIANT == 0):
            return False
        d = self.net.dropout
        if d.training and d.p > 0:
            return False
        for conv, bn, _ in self.mods:
            if isinstance(conv, torch.nn.Linear):
                if conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad):
                    return False
            elif conv.bias is not None or conv.kernel.requires_grad:
                return False
            tracked = bn.track_running_stats and bn.running_mean is not None
            if tracked:   # eval mode: running estimates; train mode: they would have to be updated
                return False
        return True

    # ------------------------------------------------------------------ layer table
    def _dense_cache(self, lin):
        c = getattr(lin, "_lt_packed_cache", None)
        if c is None:
            c = lin._lt_packed_cache = {}
        return c

    def _packed_linear(self, lin, mode: int, cin: int, cout: int):
        """nn.Linear weight [out, in] as the conv kernel [1, in, out], packed like a Conv3d kernel (same tag rules)"""
        cache, weight = self._dense_cache(lin), lin.weight
        ce = (cin, cout) if mode == 0 else (cout, cin)
        pm = mode + 4 if (ops.CONV_PRECISION != "f32" and ops.conv_uses_bf16(*ce)) else mode
        tag = (weight._version, weight.data_ptr(), LF.PACK_EPOCH, pm)
        hit = cache.get(mode)
        if hit is not None and hit[0] == tag:
            return hit[1]
        with torch.no_grad():
            packed = ops.pack_weights(weight.detach().t().contiguous().unsqueeze(0), pm)
        cache[mode] = (tag, packed)
        return packed

    def _refresh_layers(self, backward: bool) -> None:
        """pointers of the layer table: BatchNorm parameters (their storage may have been re-pointed, e.g. into
        FlatAdamEMA's flat buffers) and the packed weight images (functional._packed's invalidation rules)"""
        keep = []
        for i, (conv, bn, mp) in enumerate(self.mods):
            L = self._layers[i]
            L.gamma, L.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
            if mp == MAP_STEM:
                k = conv.kernel
                if not k.is_contiguous():
                    raise ValueError("the stem kernel must be contiguous")
                L.w_fwd, L.w_bwd = k.data_ptr(), None
                continue
            bmode = 2 if mp == MAP_K3 else 1
            if isinstance(conv, torch.nn.Linear):
                fw = self._packed_linear(conv, 0, L.cin, L.cout)
                bw = self._packed_linear(conv, bmode, L.cin, L.cout) if backward else None
            else:
                fw = LF._packed(conv.kernel, 0, conv._packed_cache)
                bw = LF._packed(conv.kernel, bmode, conv._packed_cache) if backward else None
            keep.append((fw, bw))
            L.w_fwd = fw.data_ptr()
            L.w_bwd = bw.data_ptr() if bw is not None else None
        self._keep = keep

    # ------------------------------------------------------------------ passes
    def forward(self, feats: torch.Tensor, maps: BatchMaps, keep: bool):
        """-> (out [N, C'] view of the arena, arena)"""
        self._refresh_layers(backward=keep)
        lib = _lib.load()
        self._ext[self.in_buf] = feats.data_ptr()
        n = int(lib.lt_net_arena_floats(ctypes.byref(self.prog), ctypes.byref(maps.struct), self._ext, int(keep)))
        if n < 0:
            msg = lib.lt_last_error()
            raise _lib.LatteHipError(f"lt_net_arena_floats: {msg.decode() if msg else ''}")
        arena = torch.empty(n, dtype=torch.float32, device=feats.device)
        _lib.call("lt_net_forward", ctypes.byref(self.prog), ctypes.byref(maps.struct), self._ext, ops._p(arena), n, int(keep),
                  self._off, ops._stream())
        o = int(self._off[self.out_buf])
        N, C = maps.n_points, self.bufs_py[self.out_buf][1]
        return arena[o:o + N * C].view(N, C), arena

    def backward(self, feats: torch.Tensor, maps: BatchMaps, arena: torch.Tensor, gout: torch.Tensor) -> torch.Tensor:
        """-> flat [n_dparams] BatchNorm gradients (layer i: dgamma at param_slices[i][0], dbeta at [1])"""
        self._refresh_layers(backward=True)
        lib = _lib.load()
        self._ext[self.in_buf] = feats.data_ptr()
        n = int(lib.lt_net_grad_arena_floats(ctypes.byref(self.prog), ctypes.byref(maps.struct), self._ext))
        if n < 0:
            msg = lib.lt_last_error()
            raise _lib.LatteHipError(f"lt_net_grad_arena_floats: {msg.decode() if msg else ''}")
        garena = torch.empty(n, dtype=torch.float32, device=feats.device)
        dparams = torch.empty(self.n_dparams, dtype=torch.float32, device=feats.device)
        _lib.call("lt_net_backward", ctypes.byref(self.prog), ctypes.byref(maps.struct), self._ext, ops._p(arena),
                  arena.numel(), ops._p(gout), ops._p(dparams), ops._p(garena), n, ops._stream())
        return dparams

    def conv_pairs(self, levels: dict, maps: BatchMaps, backward: bool):
        """pair counts M of the conv-kernel launches of a pass, in launch order (bench.py's flop counter): the kernel map's
        pair count for a sparse conv, the row count for a dense layer.  backward: the launches of lt_net_backward."""
        strides = (1, 2, 4, 8, 16)
        kmaps = levels[1].kmaps
        out = []
        seq = []
        for (conv, bn, mp), o in zip(self.mods, [o for o in self._ops if o.kind == CONV_BN]):
            seq.append((mp, o.level, o.in0))
        need = self._grad_needed()
        if backward:
            seq = [s for s in reversed(seq) if s[0] != MAP_STEM and need[s[2]]]
        for mp, level, _ in seq:
            if mp == MAP_DENSE:
                rows = maps.n_points if level == POINTS else maps.n_vox[level]
                out.append(torch.tensor(rows, device="cuda"))
                continue
            s = strides[level]
            key = ((s, s, s), (3, 3, 3), (1, 1, 1), (1, 1, 1)) if mp in (MAP_K3, MAP_STEM) else \
                ((s, s, s), (2, 2, 2), (2, 2, 2), (1, 1, 1))
            km = kmaps[key]
            if km._npairs is None:
                km._npairs = (km.results >= 0).sum()
            out.append(km._npairs)
        return out

    def _grad_needed(self):
        need = [False] * self.n_bufs
        for o in self._ops:
            need[o.out] = o.kind == CONV_BN or need[o.in0] or (o.in1 >= 0 and need[o.in1])
        return need


class _NetFn(Function):
    """The student pass as one autograd node: forward = lt_net_forward(keep), backward = lt_net_backward; the only
    gradients are the BatchNorm parameters' (the input features are data, the kernels are frozen)."""

    @staticmethod
    def forward(ctx, prog: NetProgram, maps: BatchMaps, levels, feats, *bn_params):
        out, arena = prog.forward(feats, maps, keep=True)
        ctx.prog, ctx.maps, ctx.arena, ctx.feats, ctx.levels = prog, maps, arena, feats, levels
        return out

    @staticmethod
    def backward(ctx, gout):
        prog = ctx.prog
        if ops.CONV_PAIRS is not None:
            ops.CONV_PAIRS.extend(prog.conv_pairs(ctx.levels, ctx.maps, backward=True))
        dparams = prog.backward(ctx.feats, ctx.maps, ctx.arena, gout.contiguous())
        ctx.arena = None
        grads = []
        for g0, b0, c in prog.param_slices:
            grads.append(dparams[g0:g0 + c])
            grads.append(dparams[b0:b0 + c])
        return (None, None, None, None) + tuple(grads)


def wants_grad(prog: NetProgram) -> bool:
    """Will a pass issued now be differentiated?  (BatchNorm affine parameters are the only trainable ones here.)"""
    return torch.is_grad_enabled() and any(bn.weight.requires_grad or bn.bias.requires_grad for _, bn, _ in prog.mods)


def run(prog: NetProgram, feats: torch.Tensor, maps: BatchMaps, levels: dict) -> torch.Tensor:
    feats = feats.contiguous()
    if ops.CONV_PAIRS is not None:
        ops.CONV_PAIRS.extend(prog.conv_pairs(levels, maps, backward=False))
    if not wants_grad(prog):
        return prog.forward(feats, maps, keep=False)[0]
    if not maps.struct.v2p_cells[0]:
        raise RuntimeError("netprog.run: the batch was prepared without the backward's tables (SPVCNN.prepare(backward=True))")
    params = []
    for _, bn, _ in prog.mods:
        params.append(bn.weight)
        params.append(bn.bias)
    return _NetFn.apply(prog, maps, levels, feats, *params)
