"""ORACLE (test infrastructure, never the product path) -- CPU restatement of the SPVCNN forward pass.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows the reference graph latte/models/spvcnn.py:122-165 (forward), :29-111 (layers),
latte/models/basic_blocks.py:13-76 (blocks), latte/models/torchsparse_utils.py:11-100 (point<->voxel) and
latte/models/xmuda_arch.py:325-335 (head), with the third-party sparse ops restated in oracle/ts_ref.py.
PARITY UNPINNED w.r.t. the real torchsparse v1.4.0 (not installable here, no fixtures in the reference).

Parameters come from a state_dict with the reference's key names (net_3d.stem.0.kernel, ...), so the same
weights drive this oracle and the HIP path.  Plain torch CPU fp32; autograd works (used for gradient parity).
Dropout is the identity (p = 0: teacher configuration, latte/common/utils/ema_util.py:88-92)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import ts_ref


class _ST:
    def __init__(self, feats, coords, stride, cmaps=None, kmaps=None):
        self.F, self.C, self.s = feats, coords, stride
        self.cmaps = {} if cmaps is None else cmaps
        self.kmaps = {} if kmaps is None else kmaps


class _PT:
    def __init__(self, feats, coords):
        self.F, self.C = feats, coords
        self.idx_query, self.weights = {}, {}
        self.add_idx, self.add_counts = {}, {}


def _bn(x, w, b, eps=1e-5):
    return F.batch_norm(x, None, None, w, b, True, 0.0, eps)  # batch statistics (ema_util.py:26-29)


def _conv(x: _ST, kernel, ks, stride=1, transposed=False) -> _ST:
    """spnn.Conv3d semantics."""
    if ks == 1:
        out = _ST(x.F @ kernel, x.C, x.s, x.cmaps, x.kmaps)
    elif not transposed:
        key = (x.s, ks, stride)
        if key not in x.kmaps:
            nbmaps, nbsizes, out_coords, _ = ts_ref.build_kmap(x.C, x.s, ks, stride)
            x.kmaps[key] = (nbmaps, nbsizes, (x.C.shape[0], out_coords.shape[0]), out_coords)
        nbmaps, nbsizes, sizes, out_coords = x.kmaps[key]
        out = _ST(ts_ref.conv3d(x.F, kernel, nbmaps, nbsizes, sizes[1]), out_coords, x.s * stride, x.cmaps, x.kmaps)
    else:
        ts = x.s // stride
        nbmaps, nbsizes, sizes, _ = x.kmaps[(ts, ks, stride)]
        out = _ST(ts_ref.conv3d(x.F, kernel, nbmaps, nbsizes, sizes[0], transposed=True), x.cmaps[ts], ts, x.cmaps,
                  x.kmaps)
    out.cmaps.setdefault(out.s, out.C)
    return out


class SPVCNNRef:
    def __init__(self, state_dict, prefix="net_3d.", dtype=torch.float32):
        self.p = {k[len(prefix):]: v.detach().clone().to(dtype).cpu() for k, v in state_dict.items()
                  if k.startswith(prefix)}
        self.head = {k: v.detach().clone().to(dtype).cpu() for k, v in state_dict.items() if not k.startswith(prefix)}
        self.traced = {}
        self.acts = {}  # post BN(+residual)(+ReLU) activations by BatchNorm module name (debug / tests)
        self.pre = {}   # the ReLU inputs under the same names (detached; the gradient-parity test looks at sign flips)
        # optional {BatchNorm module name: bool tensor}: the ReLU under that name becomes x * mask (activity pattern given
        # from outside: tests/test_gpu_spvcnn.py feeds the GPU run's pattern so that both backward passes take the same
        # branch at activations that are zero to fp32 rounding)
        self.relu_masks = None
        self.dtype = dtype

    def _relu(self, x, name):
        self.pre[name] = x.detach()
        if self.relu_masks is not None:
            return x * self.relu_masks[name].to(x.dtype)
        return torch.relu(x)

    def parameters_requiring_grad(self, names):
        for n in names:
            self.p[n].requires_grad_(True)
        return [self.p[n] for n in names]

    # --- blocks -------------------------------------------------------------------------------
    def _cbr(self, x, pre, i, ks, stride=1, transposed=False, relu=True):
        y = _conv(x, self.p[f"{pre}.{i}.kernel"], ks, stride, transposed)
        y.F = _bn(y.F, self.p[f"{pre}.{i + 1}.weight"], self.p[f"{pre}.{i + 1}.bias"])
        if relu:
            y.F = self._relu(y.F, f"{pre}.{i + 1}")
        self.acts[f"{pre}.{i + 1}"] = y.F
        return y

    def _res(self, x, pre):
        y = self._cbr(x, f"{pre}.net", 0, 3)
        y = self._cbr(y, f"{pre}.net", 3, 3, relu=False)
        if f"{pre}.downsample.0.kernel" in self.p:
            sc = self._cbr(x, f"{pre}.downsample", 0, 1, relu=False).F
        else:
            sc = x.F
        y.F = self._relu(y.F + sc, f"{pre}.net.4")
        self.acts[f"{pre}.net.4"] = y.F
        return y

    def _stage(self, x, pre):
        x = self._cbr(x, f"{pre}.0.net", 0, 2, stride=2)
        x = self._res(x, f"{pre}.1")
        return self._res(x, f"{pre}.2")

    def _up(self, y, skip, pre):
        y = self._cbr(y, f"{pre}.0.net", 0, 2, stride=2, transposed=True)
        y = _ST(torch.cat([y.F, skip.F], 1), y.C, y.s, y.cmaps, y.kmaps)
        y = self._res(y, f"{pre}.1.0")
        return self._res(y, f"{pre}.1.1")

    def _pt(self, f, i):
        pre = f"point_transforms.{i}"
        y = F.linear(f, self.p[f"{pre}.0.weight"], self.p[f"{pre}.0.bias"])
        return self._relu(_bn(y, self.p[f"{pre}.1.weight"], self.p[f"{pre}.1.bias"]), f"{pre}.1")

    # --- point <-> voxel (torchsparse_utils.py) -------------------------------------------------
    def _initial_voxelize(self, z: _PT, init_res, after_res):
        nfc = torch.cat([(z.C[:, :3] * init_res) / after_res, z.C[:, -1:].clone()], 1)
        fl = torch.floor(nfc)
        pc_hash = ts_ref.sphash(fl.int().numpy())
        sparse_hash = np.unique(pc_hash)
        idx = ts_ref.sphashquery(pc_hash, sparse_hash)
        counts = ts_ref.spcount(idx.astype(np.int32), sparse_hash.shape[0])
        it, ct = torch.from_numpy(idx), torch.from_numpy(counts)
        coords = torch.round(ts_ref.spvoxelize(fl, it, ct)).int().numpy()
        x = _ST(ts_ref.spvoxelize(z.F, it, ct), coords, 1)
        x.cmaps.setdefault(1, coords)
        z.add_idx[1], z.add_counts[1] = it, ct
        z.C = nfc
        return x

    @staticmethod
    def _cells(z, s):
        c = torch.cat([torch.floor(z.C[:, :3] / s).int() * s, z.C[:, -1:].int()], 1)
        return c.numpy()

    def _p2v(self, x: _ST, z: _PT):
        if x.s not in z.add_idx:
            idx = ts_ref.sphashquery(ts_ref.sphash(self._cells(z, x.s)), ts_ref.sphash(x.C))
            z.add_idx[x.s] = torch.from_numpy(idx)
            z.add_counts[x.s] = torch.from_numpy(ts_ref.spcount(idx.astype(np.int32), x.C.shape[0]))
        return _ST(ts_ref.spvoxelize(z.F, z.add_idx[x.s], z.add_counts[x.s]), x.C, x.s, x.cmaps, x.kmaps)

    def _v2p(self, x: _ST, z: _PT):
        if x.s not in z.idx_query:
            off = ts_ref.get_kernel_offsets(2, x.s, 1)
            q = ts_ref.sphashquery(ts_ref.sphash(self._cells(z, x.s), off), ts_ref.sphash(x.C))
            qt = torch.from_numpy(q)
            w = ts_ref.calc_ti_weights(z.C, qt, x.s).t().contiguous()
            z.idx_query[x.s], z.weights[x.s] = qt.t().contiguous(), w
        nz = _PT(ts_ref.spdevoxelize(x.F, z.idx_query[x.s], z.weights[x.s]), z.C)
        nz.idx_query, nz.weights, nz.add_idx, nz.add_counts = z.idx_query, z.weights, z.add_idx, z.add_counts
        return nz

    # --- forward ----------------------------------------------------------------------------------
    def backbone(self, coords: np.ndarray, feats: torch.Tensor) -> torch.Tensor:
        z = _PT(feats.to(self.dtype), torch.from_numpy(coords.astype(np.float32)))
        x0 = self._initial_voxelize(z, 0.05, 0.05)
        self.traced["x0_coords"] = x0.C
        self.traced["idx_query_1"] = z.add_idx[1].numpy()
        x0 = self._cbr(x0, "stem", 0, 3)
        x0 = self._cbr(x0, "stem", 3, 3)
        z0 = self._v2p(x0, z)
        x1 = self._stage(self._p2v(x0, z0), "stage1")
        x2 = self._stage(x1, "stage2")
        x3 = self._stage(x2, "stage3")
        x4 = self._stage(x3, "stage4")
        for n, t in (("x1", x1), ("x2", x2), ("x3", x3), ("x4", x4)):
            self.traced[n + "_coords"] = t.C
        z1 = self._v2p(x4, z0)
        z1.F = z1.F + self._pt(z0.F, 0)
        y1 = self._up(self._p2v(x4, z1), x3, "up1")
        y2 = self._up(y1, x2, "up2")
        z2 = self._v2p(y2, z1)
        z2.F = z2.F + self._pt(z1.F, 1)
        y3 = self._up(self._p2v(y2, z2), x1, "up3")
        y4 = self._up(y3, x0, "up4")
        z3 = self._v2p(y4, z2)
        z3.F = z3.F + self._pt(z2.F, 2)
        self.traced["kmaps"] = x0.kmaps
        self.traced["idx_query"] = {s: q.numpy() for s, q in z.idx_query.items()}
        return z3.F

    def __call__(self, coords: np.ndarray, feats: torch.Tensor):
        f = self.backbone(coords, feats)
        out = {"feats": f, "seg_logit": F.linear(f, self.head["linear.weight"], self.head["linear.bias"])}
        if "linear2.weight" in self.head:
            out["seg_logit2"] = F.linear(f, self.head["linear2.weight"], self.head["linear2.bias"])
        return out
