"""Tensor-level wrappers over the C-ABI (no autograd here; see functional.py).

torch is plumbing only: it owns device memory and the stream; every op below is one or more calls into
liblatte_hip.so.  All tensors must live on a CUDA(HIP) device -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

__all__ = [
    "sphash", "HashTable", "sphashquery", "spcount", "unique_sorted", "pack_xyz_keys", "downsample_coords",
    "kmap_build", "kmap_pairs", "kmap_reverse", "kmap_order", "kmap_plan", "spconv_plan", "pack_weights", "spconv", "bn_stats", "bn_apply",
    "bn_bwd", "voxelize_fwd", "voxelize_bwd", "calc_ti_weights", "devoxelize_fwd", "devoxelize_bwd", "csr_build", "voxelize_csr_fwd", "devoxelize_csr_bwd", "devoxelize_cells", "devoxelize_cell_bwd",
]


def _stream():
    # raw handle of torch's current stream on the current device (torch.cuda.current_stream() costs ~9 us per call)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _chk(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.LatteHipError(f"{name}: tensor must be on the GPU (no CPU fallback in the product path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# Transient workspaces of the hot per-layer ops (BatchNorm partial sums, conv statistics): their contents live inside ONE
# library call, and calls on one stream execute in order, so a buffer per (thread, stream) is reused by every call instead
# of a fresh torch.empty each (3.5 us of host time per allocation, ~300 of them per step).  Never for outputs or for
# anything a later call reads (tables, plans).
_SCRATCH: dict = {}
_WS_BYTES: dict = {}


def _scratch(nbytes: int, device) -> torch.Tensor:
    import threading
    idx = device.index if device.index is not None else torch._C._cuda_getDevice()
    key = (threading.get_ident(), torch._C._cuda_getCurrentRawStream(idx), idx)
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nbytes:
        t = _SCRATCH[key] = torch.empty(max(int(nbytes) * 3 // 2, 1 << 20), dtype=torch.uint8, device=device)
    return t


def _ws_bytes(fn_name: str, *args) -> int:
    """cached *_workspace_bytes query (a ctypes round trip per layer otherwise)"""
    key = (fn_name,) + args
    v = _WS_BYTES.get(key)
    if v is None:
        if len(_WS_BYTES) > 4096:
            _WS_BYTES.clear()
        v = _WS_BYTES[key] = int(getattr(_lib.load(), fn_name)(*args))
    return v


# ------------------------------------------------------------------ hashing
def sphash(coords: torch.Tensor, offsets: torch.Tensor | None = None) -> torch.Tensor:
    coords = _chk(coords, torch.int32, "sphash coords")
    assert coords.dim() == 2 and coords.shape[1] == 4, coords.shape
    n = coords.shape[0]
    if offsets is None:
        out = torch.empty(n, dtype=torch.int64, device=coords.device)
        _lib.call("lt_sphash", _p(coords), n, None, 0, _p(out), _stream())
        return out
    offsets = _chk(offsets, torch.int32, "sphash offsets")
    assert offsets.dim() == 2 and offsets.shape[1] == 3
    K = offsets.shape[0]
    out = torch.empty((K, n), dtype=torch.int64, device=coords.device)
    _lib.call("lt_sphash", _p(coords), n, _p(offsets), K, _p(out), _stream())
    return out


class HashTable:
    """Open-addressing table over int64 keys (value = row index)."""

    def __init__(self, keys: torch.Tensor):
        keys = _chk(keys.reshape(-1), torch.int64, "HashTable keys")
        self.n = keys.numel()
        lib = _lib.load()
        self.mem = _ws(lib.lt_table_bytes(self.n), keys.device)
        _lib.call("lt_hash_build", _p(keys), self.n, _p(self.mem), self.mem.numel(), _stream())

    def query(self, queries: torch.Tensor) -> torch.Tensor:
        q = _chk(queries, torch.int64, "HashTable queries")
        out = torch.empty_like(q)
        _lib.call("lt_hash_query", _p(q), q.numel(), _p(self.mem), self.n, _p(out), _stream())
        return out


def sphashquery(queries: torch.Tensor, references: torch.Tensor) -> torch.Tensor:
    return HashTable(references).query(queries)


def spcount(idx: torch.Tensor, num: int) -> torch.Tensor:
    idx = _chk(idx.reshape(-1), torch.int32, "spcount idx")
    out = torch.empty(int(num), dtype=torch.int32, device=idx.device)
    _lib.call("lt_spcount", _p(idx), idx.numel(), _p(out), int(num), _stream())
    return out


def unique_sorted(keys: torch.Tensor, key_bits: int = 63, want_first=True, want_inverse=True):
    """-> (sorted unique keys [V], first index [V] | None, inverse [N] | None)."""
    keys = _chk(keys.reshape(-1), torch.int64, "unique_sorted keys")
    n = keys.numel()
    dev = keys.device
    lib = _lib.load()
    ws = _ws(lib.lt_unique_workspace_bytes(n), dev)
    uk = torch.empty(n, dtype=torch.int64, device=dev)
    first = torch.empty(n, dtype=torch.int64, device=dev) if want_first else None
    inv = torch.empty(n, dtype=torch.int64, device=dev) if want_inverse else None
    cnt = ctypes.c_int64(0)
    _lib.call("lt_unique_sorted", _p(keys), n, int(key_bits), _p(ws), ws.numel(), _p(uk), _p(first), _p(inv),
              ctypes.byref(cnt), _stream())
    v = cnt.value
    return uk[:v], (first[:v] if want_first else None), inv


def pack_xyz_keys(coords3: torch.Tensor) -> torch.Tensor:
    coords3 = _chk(coords3, torch.int32, "pack_xyz_keys coords")
    assert coords3.dim() == 2 and coords3.shape[1] == 3
    keys = torch.empty(coords3.shape[0], dtype=torch.int64, device=coords3.device)
    _lib.call("lt_pack_xyz_keys", _p(coords3), coords3.shape[0], _p(keys), _stream())
    return keys


# ------------------------------------------------------------------ kernel maps
def downsample_coords(coords: torch.Tensor, sample_stride: int) -> torch.Tensor:
    coords = _chk(coords, torch.int32, "downsample coords")
    V = coords.shape[0]
    lib = _lib.load()
    ws = _ws(lib.lt_downsample_workspace_bytes(V), coords.device)
    out = torch.empty((V, 4), dtype=torch.int32, device=coords.device)
    cnt = ctypes.c_int64(0)
    _lib.call("lt_downsample_coords", _p(coords), V, int(sample_stride), _p(ws), ws.numel(), _p(out),
              ctypes.byref(cnt), _stream())
    return out[:cnt.value]


def kmap_build(coords_in: torch.Tensor, coords_out: torch.Tensor, offsets: torch.Tensor, submanifold: bool = False) -> torch.Tensor:
    """submanifold=True: coords_out IS coords_in (unique rows), K odd, offsets point-symmetric -> lt_kmap_build_sub, which
    probes K/2 + 1 offsets per row and mirrors the hits (same table)."""
    coords_in = _chk(coords_in, torch.int32, "kmap coords_in")
    coords_out = _chk(coords_out, torch.int32, "kmap coords_out")
    offsets = _chk(offsets, torch.int32, "kmap offsets")
    Vin, Vout, K = coords_in.shape[0], coords_out.shape[0], offsets.shape[0]
    lib = _lib.load()
    ws = _ws(lib.lt_kmap_workspace_bytes(Vin), coords_in.device)
    res = torch.empty((K, Vout), dtype=torch.int32, device=coords_in.device)
    if submanifold:
        if coords_out.data_ptr() != coords_in.data_ptr() or K % 2 != 1:
            raise ValueError("kmap_build(submanifold=True): needs coords_out is coords_in and an odd kernel volume")
        _lib.call("lt_kmap_build_sub", _p(coords_in), Vin, _p(offsets), K, _p(ws), ws.numel(), _p(res), _stream())
        return res
    _lib.call("lt_kmap_build", _p(coords_in), Vin, _p(coords_out), Vout, _p(offsets), K, _p(ws), ws.numel(),
              _p(res), _stream())
    return res


def kmap_pairs(results: torch.Tensor):
    results = _chk(results, torch.int32, "kmap_pairs results")
    K, Vout = results.shape
    lib = _lib.load()
    ws = _ws(lib.lt_kmap_pairs_workspace_bytes(K, Vout), results.device)
    nbmaps = torch.empty((max(K * Vout, 1), 2), dtype=torch.int32, device=results.device)
    nbsizes = torch.empty(K, dtype=torch.int32, device=results.device)
    m = ctypes.c_int64(0)
    _lib.call("lt_kmap_pairs", _p(results), K, Vout, _p(ws), ws.numel(), _p(nbmaps), _p(nbsizes),
              ctypes.byref(m), _stream())
    return nbmaps[:m.value], nbsizes


def kmap_reverse(results: torch.Tensor, Vin: int) -> torch.Tensor:
    results = _chk(results, torch.int32, "kmap_reverse results")
    K, Vout = results.shape
    rev = torch.empty((K, int(Vin)), dtype=torch.int32, device=results.device)
    _lib.call("lt_kmap_reverse", _p(results), K, Vout, int(Vin), _p(rev), _stream())
    return rev


def kmap_order(table: torch.Tensor) -> torch.Tensor:
    """Row processing order of a neighbour table [K,V] for spconv (rows sorted by their neighbour mask)."""
    table = _chk(table, torch.int32, "kmap_order table")
    K, V = table.shape
    lib = _lib.load()
    ws = _ws(lib.lt_kmap_order_workspace_bytes(V), table.device)
    order = torch.empty(V, dtype=torch.int32, device=table.device)
    _lib.call("lt_kmap_order", _p(table), K, V, _p(ws), ws.numel(), _p(order), _stream())
    return order


def kmap_plan(table: torch.Tensor) -> torch.Tensor:
    """Conv plan (opaque uint8 buffer) of a neighbour table [K,V]; see lt_kmap_plan."""
    table = _chk(table, torch.int32, "kmap_plan table")
    K, V = table.shape
    lib = _lib.load()
    ws = _ws(lib.lt_kmap_plan_workspace_bytes(V), table.device)
    plan = _ws(lib.lt_kmap_plan_bytes(K, V), table.device)
    _lib.call("lt_kmap_plan", _p(table), K, V, _p(ws), ws.numel(), _p(plan), plan.numel(), _stream())
    return plan


# ------------------------------------------------------------------ sparse conv
def pack_weights(weight: torch.Tensor, mode: int = 0) -> torch.Tensor:
    """weight [K,Cin,Cout] (or [Cin,Cout]) fp32 -> MFMA-fragment order (see spconv.hip)."""
    w = _chk(weight.detach(), torch.float32, "pack_weights")
    if w.dim() == 2:
        w = w.unsqueeze(0)
    K, Cin, Cout = w.shape
    packed = torch.empty(K * Cin * Cout, dtype=torch.float32, device=w.device)
    _lib.call("lt_spconv_pack_weights", _p(w), K, Cin, Cout, int(mode), _p(packed), _stream())
    return packed


CONV_PRECISION = "f32"


def set_conv_precision(name: str) -> None:
    """'f32' (default) or 'bf16x3' (split-precision bf16 MFMA for the convs with >= 64 input channels): process-wide, see
    lt_conv_set_precision.  Packed weight images made under the other setting are re-made (the cache tag carries it)."""
    global CONV_PRECISION
    mode = {"f32": 0, "fp32": 0, "bf16x3": 1}[name]
    _lib.call("lt_conv_set_precision", mode)
    CONV_PRECISION = "bf16x3" if mode else "f32"


def conv_uses_bf16(cin: int, cout: int) -> bool:
    return CONV_PRECISION != "f32" and bool(_lib.load().lt_conv_uses_bf16(int(cin), int(cout)))


def conv_timer_begin(capacity: int) -> None:
    """bench.py's kernel timer: until conv_timer_end() the library brackets every sparse-conv kernel launch with two HIP
    events on the stream the kernel is launched on (lt_conv_timer_begin)."""
    _lib.call("lt_conv_timer_begin", int(capacity))


def conv_timer_arm(on: bool) -> None:
    """pause / resume the timer between conv_timer_begin and conv_timer_end"""
    _lib.call("lt_conv_timer_arm", int(bool(on)))


def conv_timer_end():
    """-> (ms float32 [n], meta int32 [n,4] = Cin, Cout, K, Vout), one row per conv kernel launch since conv_timer_begin."""
    import numpy as np
    lib = _lib.load()
    cap = 1 << 20
    ms = np.empty(cap, dtype=np.float32)
    meta = np.empty((cap, 4), dtype=np.int32)
    n = lib.lt_conv_timer_end(ms.ctypes.data_as(ctypes.c_void_p), meta.ctypes.data_as(ctypes.c_void_p), cap)
    if n < 0:
        msg = lib.lt_last_error()
        raise _lib.LatteHipError(f"lt_conv_timer_end failed (rc={n}): {msg.decode() if msg else ''}")
    return ms[:n].copy(), meta[:n].copy()


# bench.py's flop counter (untimed pass): when a list, every spconv call appends its pair count (device scalar).
CONV_PAIRS = None


def spconv(feats: torch.Tensor, packed: torch.Tensor, nbr: torch.Tensor, cout: int,
           order: torch.Tensor | None = None) -> torch.Tensor:
    """out[o] = sum_k feats[nbr[k,o]] @ E_k with E packed by pack_weights; nbr int32 [K,Vout]."""
    feats = _chk(feats, torch.float32, "spconv feats")
    nbr = _chk(nbr, torch.int32, "spconv nbr")
    packed = _chk(packed, torch.float32, "spconv packed")
    Vin, Cin = feats.shape
    K, Vout = nbr.shape
    if packed.numel() != K * Cin * cout:
        raise ValueError(f"spconv: packed weights have {packed.numel()} floats, expected {K}*{Cin}*{cout}")
    if order is not None:
        order = _chk(order, torch.int32, "spconv order")
        assert order.numel() == Vout
    out = torch.empty((Vout, cout), dtype=torch.float32, device=feats.device)
    _lib.call("lt_spconv_fwd", _p(feats), _p(packed), _p(nbr), _p(order), _p(out), Vin, Vout, Cin, int(cout), K,
              _stream())
    return out


def spconv_stem(feats: torch.Tensor, weight: torch.Tensor, plan: torch.Tensor, K: int, Vout: int) -> torch.Tensor:
    """the 4-channel first conv through its streaming kernel; weight = the unpacked kernel [K, 4, Cout] (Cout 16 | 32)"""
    feats = _chk(feats, torch.float32, "spconv_stem feats")
    w = _chk(weight.detach(), torch.float32, "spconv_stem weight")
    assert w.dim() == 3 and w.shape[0] == K and w.shape[1] == 4 and feats.shape[1] == 4
    cout = w.shape[2]
    out = torch.empty((Vout, cout), dtype=torch.float32, device=feats.device)
    _lib.call("lt_spconv_stem_fwd", _p(feats), _p(w), _p(plan), _p(out), feats.shape[0], Vout, 4, int(cout), int(K),
              _stream())
    return out


def spconv_bwd_weight(feats, gout, nbmaps, nbsizes, K: int, cin: int, cout: int, transposed: bool = False):
    """dW [K,Cin,Cout] = per offset feats[in rows]^T @ gout[out rows] over the pair list (lt_spconv_bwd_weight)."""
    feats = _chk(feats, torch.float32, "bwd_weight feats")
    gout = _chk(gout, torch.float32, "bwd_weight gout")
    nbmaps = _chk(nbmaps, torch.int32, "bwd_weight nbmaps")
    nbsizes = _chk(nbsizes, torch.int32, "bwd_weight nbsizes")
    M = nbmaps.shape[0]
    assert feats.shape[1] == cin and gout.shape[1] == cout and nbsizes.numel() == K
    lib = _lib.load()
    ws = _ws(lib.lt_spconv_bwd_weight_workspace_bytes(M, K, cin, cout), feats.device)
    dW = torch.empty((K, cin, cout), dtype=torch.float32, device=feats.device)
    _lib.call("lt_spconv_bwd_weight", _p(feats), _p(gout), _p(nbmaps), _p(nbsizes), M, feats.shape[0], gout.shape[0],
              int(cin), int(cout), int(K), int(bool(transposed)), _p(ws), ws.numel(), _p(dW), _stream())
    return dW


CONV_VARIANT = 0  # lt_spconv_plan_fwd_v variant (0 = library default); tuning knob of tools/conv_bench.py


def spconv_plan(feats: torch.Tensor, packed: torch.Tensor, plan: torch.Tensor, K: int, Vout: int, cout: int,
                variant: int | None = None):
    """spconv through a plan of the [K,Vout] table (the hot path; Cin % 16 == 0)."""
    feats = _chk(feats, torch.float32, "spconv feats")
    packed = _chk(packed, torch.float32, "spconv packed")
    Vin, Cin = feats.shape
    if packed.numel() != K * Cin * cout:
        raise ValueError(f"spconv: packed weights have {packed.numel()} floats, expected {K}*{Cin}*{cout}")
    out = torch.empty((Vout, cout), dtype=torch.float32, device=feats.device)
    _lib.call("lt_spconv_plan_fwd_v", _p(feats), _p(packed), _p(plan), _p(out), Vin, Vout, Cin, int(cout), K,
              int(CONV_VARIANT if variant is None else variant), _stream())
    return out


# ------------------------------------------------------------------ batch norm
def bn_stats(x: torch.Tensor):
    x = _chk(x, torch.float32, "bn_stats x")
    V, C = x.shape
    ws = _scratch(_ws_bytes("lt_bn_workspace_bytes", V, C), x.device)
    mv = torch.empty((2, C), dtype=torch.float32, device=x.device)
    mean, var = mv[0], mv[1]
    _lib.call("lt_bn_stats", _p(x), V, C, _p(ws), ws.numel(), _p(mean), _p(var), _stream())
    return mean, var


def bn_apply(x, mean, var, gamma, beta, eps: float, residual=None, relu: bool = False, out=None, post_add: bool = False):
    """y = relu?(bn(x) + residual); post_add (with relu and a residual): y = relu(bn(x)) + residual."""
    x = _chk(x, torch.float32, "bn_apply x")
    V, C = x.shape
    y = torch.empty_like(x) if out is None else out
    if residual is not None:
        residual = _chk(residual, torch.float32, "bn_apply residual")
        assert residual.shape == x.shape
    _lib.call("lt_bn_apply", _p(x), _p(mean), _p(var), _p(_chk(gamma.detach(), torch.float32, "gamma")),
              _p(_chk(beta.detach(), torch.float32, "beta")), float(eps), _p(residual),
              3 if (post_add and relu and residual is not None) else int(bool(relu)), _p(y), V, C, _stream())
    return y


def bn_apply_devox(x, mean, var, gamma, beta, eps: float, vox, idx, w):
    """relu(bn(x)) + devoxelize(vox, idx, w) in one kernel (lt_bn_apply_devox): x [N,C], vox [V,C], idx int32 [N,8], w [N,8]."""
    x = _chk(x, torch.float32, "bn_apply_devox x")
    vox = _chk(vox, torch.float32, "bn_apply_devox vox")
    idx = _chk(idx, torch.int32, "bn_apply_devox idx")
    w = _chk(w, torch.float32, "bn_apply_devox w")
    N, C = x.shape
    assert idx.shape == (N, 8) and w.shape == (N, 8) and vox.shape[1] == C
    y = torch.empty_like(x)
    _lib.call("lt_bn_apply_devox", _p(x), _p(mean), _p(var), _p(_chk(gamma.detach(), torch.float32, "gamma")),
              _p(_chk(beta.detach(), torch.float32, "beta")), float(eps), _p(vox), _p(idx), _p(w), _p(y), N, int(C),
              vox.shape[0], _stream())
    return y


def _relu_mode(relu: bool, y, beta) -> int:
    """0 = no ReLU; 1 = mask from the saved output y; 2 = no y saved (the forward added no residual): the kernels rebuild
    the mask from x, gamma, beta with the forward's own expression"""
    if not relu:
        return 0
    if y is not None:
        return 1
    if beta is None:
        raise ValueError("ReLU backward needs the saved output y, or beta to rebuild the mask from x")
    return 2


def bn_bwd(dy, x, y, mean, var, gamma, eps: float, relu: bool, need_dx=True, need_dres=False, beta=None, sync=None):
    """-> (dx | None, dgamma, dbeta, dres | None).  y None with relu: mask rebuilt from x (needs beta; no residual).
    sync (optional): callable (sum_g, sum_gx) -> (sum_g, sum_gx, scale) applied between the column reduction and the
    elementwise pass -- cross-rank BatchNorm: the sums become the global ones and the elementwise pass, which divides
    by the LOCAL row count, gets them scaled by local rows / global rows."""
    dy = _chk(dy, torch.float32, "bn_bwd dy")
    x = _chk(x, torch.float32, "bn_bwd x")
    V, C = x.shape
    ws = _scratch(_ws_bytes("lt_bn_workspace_bytes", V, C), x.device)
    sums = torch.empty((2, C), dtype=torch.float32, device=x.device)
    sum_g, sum_gx = sums[0], sums[1]
    mode = _relu_mode(relu, y, beta)
    yp = _p(y) if mode == 1 else None
    gp = _p(_chk(gamma.detach(), torch.float32, "gamma"))
    bp = _p(_chk(beta.detach(), torch.float32, "beta")) if beta is not None else None
    _lib.call("lt_bn_bwd_reduce", _p(dy), _p(x), yp, _p(mean), _p(var), gp, bp, float(eps), mode, V, C, _p(ws),
              ws.numel(), _p(sum_g), _p(sum_gx), _stream())
    dx = torch.empty_like(x) if need_dx else None
    dres = torch.empty_like(x) if need_dres else None
    ag, agx = sum_g, sum_gx
    if sync is not None:
        sum_g, sum_gx, scale = sync(sum_g, sum_gx)
        ag, agx = (sum_g * scale).contiguous(), (sum_gx * scale).contiguous()
    if need_dx or need_dres:
        _lib.call("lt_bn_bwd_apply", _p(dy), _p(x), yp, _p(mean), _p(var), gp, bp, float(eps), mode, _p(ag),
                  _p(agx), _p(dx), _p(dres), V, C, _stream())
    return dx, sum_gx, sum_g, dres


# ------------------------------------------------------------------ point <-> voxel
def voxelize_fwd(feats, idx, counts):
    feats = _chk(feats, torch.float32, "voxelize feats")
    idx = _chk(idx, torch.int32, "voxelize idx")
    counts = _chk(counts, torch.int32, "voxelize counts")
    N, C = feats.shape
    V = counts.numel()
    out = torch.empty((V, C), dtype=torch.float32, device=feats.device)
    _lib.call("lt_voxelize_fwd", _p(feats), _p(idx), _p(counts), _p(out), N, C, V, _stream())
    return out


def voxelize_bwd(gout, idx, counts, N: int):
    gout = _chk(gout, torch.float32, "voxelize_bwd gout")
    V, C = gout.shape
    gf = torch.empty((N, C), dtype=torch.float32, device=gout.device)
    _lib.call("lt_voxelize_bwd", _p(gout), _p(idx), _p(counts), _p(gf), N, C, V, _stream())
    return gf


def csr_build(key: torch.Tensor, V: int):
    """CSR of a scatter index: key int32 [n] with values in [0,V) (others dropped) -> (perm int32 [n], off int32 [V+1])."""
    key = _chk(key.reshape(-1), torch.int32, "csr key")
    n = key.numel()
    dev = key.device
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    off = torch.empty(V + 1, dtype=torch.int32, device=dev)
    ws = _ws(_lib.load().lt_csr_workspace_bytes(n), dev)
    _lib.call("lt_csr_build", _p(key), n, int(V), _p(ws), ws.numel(), _p(perm), _p(off), _stream())
    return perm, off


def voxelize_csr_fwd(feats, csr, V: int):
    feats = _chk(feats, torch.float32, "voxelize feats")
    perm, off = csr
    assert off.numel() == V + 1
    C = feats.shape[1]
    out = torch.empty((V, C), dtype=torch.float32, device=feats.device)
    _lib.call("lt_voxelize_csr_fwd", _p(feats), _p(perm), _p(off), _p(out), V, C, _stream())
    return out


def devoxelize_csr_bwd(gout, csr, w, V: int):
    gout = _chk(gout, torch.float32, "devox_bwd gout")
    perm, off = csr
    assert off.numel() == V + 1 and w.numel() == perm.numel()
    C = gout.shape[1]
    gf = torch.empty((V, C), dtype=torch.float32, device=gout.device)
    _lib.call("lt_devoxelize_csr_bwd", _p(gout), _p(perm), _p(off), _p(w), _p(gf), V, C, _stream())
    return gf


def devoxelize_cells(v2p_idx, p2v_csr, V: int):
    """lt_devoxelize_cells: [V, 8, 2] table "CSR row range of the cell that reaches voxel v through corner c"."""
    v2p_idx = _chk(v2p_idx, torch.int32, "devox_cells idx")
    perm, off = p2v_csr
    N = v2p_idx.shape[0]
    assert v2p_idx.shape[1] == 8 and perm.numel() <= N
    cells = torch.empty((V, 8, 2), dtype=torch.int32, device=v2p_idx.device)
    _lib.call("lt_devoxelize_cells", _p(v2p_idx), _p(perm), _p(off), _p(cells), N, off.numel() - 1, V, _stream())
    return cells


def devoxelize_cell_bwd(gout, cells, p2v_csr, p2v_idx, v2p_idx, w):
    gout = _chk(gout, torch.float32, "devox_bwd gout")
    p2v_idx = _chk(p2v_idx, torch.int32, "devox_bwd p2v idx")
    v2p_idx = _chk(v2p_idx, torch.int32, "devox_bwd v2p idx")
    perm, off = p2v_csr
    N, C = gout.shape
    V = cells.shape[0]
    assert p2v_idx.numel() == N and v2p_idx.numel() == 8 * N and w.numel() == 8 * N and w.is_contiguous()
    gf = torch.empty((V, C), dtype=torch.float32, device=gout.device)
    _lib.call("lt_devoxelize_cell_bwd", _p(gout), _p(cells), _p(perm), _p(off), _p(p2v_idx), _p(v2p_idx), _p(w), _p(gf), N, V,
              C, _stream())
    return gf


def calc_ti_weights(pcoords, idx_query, scale: int):
    """pcoords fp32 [N,4]; idx_query int64 [8,N] -> fp32 [8,N]."""
    pcoords = _chk(pcoords, torch.float32, "ti_weights coords")
    idx_query = _chk(idx_query, torch.int64, "ti_weights idx_query")
    assert pcoords.shape[1] == 4 and idx_query.shape[0] == 8
    N = pcoords.shape[0]
    w = torch.empty((8, N), dtype=torch.float32, device=pcoords.device)
    _lib.call("lt_calc_ti_weights", _p(pcoords), _p(idx_query), N, int(scale), _p(w), _stream())
    return w


def devoxelize_fwd(feats, idx, w):
    feats = _chk(feats, torch.float32, "devox feats")
    idx = _chk(idx, torch.int32, "devox idx")
    w = _chk(w, torch.float32, "devox w")
    V, C = feats.shape
    N = idx.shape[0]
    out = torch.empty((N, C), dtype=torch.float32, device=feats.device)
    _lib.call("lt_devoxelize_fwd", _p(feats), _p(idx), _p(w), _p(out), N, C, V, _stream())
    return out


def devoxelize_bwd(gout, idx, w, V: int):
    gout = _chk(gout, torch.float32, "devox_bwd gout")
    N, C = gout.shape
    gf = torch.empty((V, C), dtype=torch.float32, device=gout.device)
    _lib.call("lt_devoxelize_bwd", _p(gout), _p(idx), _p(w), _p(gf), N, C, V, _stream())
    return gf


# ------------------------------------------------------------------ ST-voxel aggregation (include/latte_hip_st.h)
def softmax_entropy(logits: torch.Tensor, want_probs=True, want_ety=True):
    logits = _chk(logits, torch.float32, "softmax_entropy logits")
    P, C = logits.shape
    probs = torch.empty_like(logits) if want_probs else None
    ety = torch.empty(P, dtype=torch.float32, device=logits.device) if want_ety else None
    _lib.call("lt_softmax_entropy", _p(logits), P, C, _p(probs), _p(ety), _stream())
    return probs, ety


def softmax_bwd(probs, gprobs):
    probs = _chk(probs, torch.float32, "softmax_bwd probs")
    gprobs = _chk(gprobs, torch.float32, "softmax_bwd g")
    out = torch.empty_like(probs)
    _lib.call("lt_softmax_bwd", _p(probs), _p(gprobs), probs.shape[0], probs.shape[1], _p(out), _stream())
    return out


def st_coord_bits(max_range, voxel_size) -> int:
    """bits per ST-voxel coordinate: shifted coordinates lie in [0, 2*ceil(R/vs) + 1] (hr_module.py:53,62)"""
    import math
    return max(1, int(math.ceil(2.0 * max_range / voxel_size + 4.0)).bit_length())


def st_keys(points, seg_out_start, seg_src_start, seg_anchor, seg_T, E: int, max_range, voxel_size, min_offset,
            coord_bits: int | None = None):
    """-> (keys int64 [E] = anchor << 3b | x << 2b | y << b | z with b = coord_bits, or INT64_MAX; src int32 [E])"""
    points = _chk(points, torch.float32, "st_keys points")
    S = seg_anchor.numel()
    if coord_bits is None:
        coord_bits = st_coord_bits(max_range, voxel_size)
    keys = torch.empty(E, dtype=torch.int64, device=points.device)
    src = torch.empty(E, dtype=torch.int32, device=points.device)
    _lib.call("lt_st_keys", _p(points), _p(_chk(seg_out_start, torch.int64, "seg_out_start")),
              _p(_chk(seg_src_start, torch.int64, "seg_src_start")), _p(_chk(seg_anchor, torch.int32, "seg_anchor")),
              _p(None if seg_T is None else _chk(seg_T, torch.float64, "seg_T")), S, int(E), float(max_range),
              float(voxel_size), int(min_offset), int(coord_bits), _p(keys), _p(src), _stream())
    return keys, src


def sort_segments(keys: torch.Tensor, key_bits: int = 63):
    """-> (sorted_elem int32 [E], inverse int32 [E], uniq_keys int64 [V], seg_start int32 [V+1]).  key_bits: every
    unmasked key is < 2^key_bits (fewer radix passes)."""
    keys = _chk(keys.reshape(-1), torch.int64, "sort_segments keys")
    E = keys.numel()
    dev = keys.device
    lib = _lib.load()
    ws = _ws(lib.lt_sort_segments_workspace_bytes(E), dev)
    selem = torch.empty(E, dtype=torch.int32, device=dev)
    inv = torch.empty(E, dtype=torch.int32, device=dev)
    uk = torch.empty(max(E, 1), dtype=torch.int64, device=dev)
    seg = torch.zeros(E + 1, dtype=torch.int32, device=dev)
    v = ctypes.c_int64(0)
    _lib.call("lt_sort_segments", _p(keys), E, int(key_bits), _p(ws), ws.numel(), _p(selem), _p(inv), _p(uk), _p(seg),
              ctypes.byref(v), _stream())
    V = v.value
    return selem, inv, uk[:V], seg[:V + 1]


_MODES = {"sum": 0, "add": 0, "mean": 1, "max": 2}


def segment_reduce(src, sorted_elem, seg_start, mode: str, src_index=None, want_argmax=False):
    src = _chk(src, torch.float32, "segment_reduce src")
    squeeze = src.dim() == 1
    s2 = src.reshape(src.shape[0], -1)
    C = s2.shape[1]
    V = seg_start.numel() - 1
    out = torch.empty((V, C), dtype=torch.float32, device=src.device)
    arg = torch.empty((V, C), dtype=torch.int32, device=src.device) if want_argmax else None
    _lib.call("lt_segment_reduce", _p(s2), C, _p(src_index), _p(sorted_elem), _p(seg_start), V, _MODES[mode], _p(out),
              _p(arg), _stream())
    out = out.reshape(V) if squeeze else out.reshape((V,) + tuple(src.shape[1:]))
    return (out, arg) if want_argmax else out


def segment_reduce_bwd(gout, inverse, seg_start, mean: bool):
    gout = _chk(gout, torch.float32, "segment_reduce_bwd g")
    g2 = gout.reshape(gout.shape[0], -1)
    C = g2.shape[1]
    E = inverse.numel()
    ge = torch.empty((E, C), dtype=torch.float32, device=gout.device)
    _lib.call("lt_segment_reduce_bwd", _p(g2), C, _p(inverse), _p(seg_start), E, int(mean), _p(ge), _stream())
    return ge.reshape((E,) + tuple(gout.shape[1:]))


def st_gather_back(voxel_ety, inverse, frame_start, frame_seg, seg_out_start, P: int):
    F, Wmax = frame_seg.shape
    out = torch.empty(P, dtype=torch.float32, device=inverse.device)
    _lib.call("lt_st_gather_back", _p(_chk(voxel_ety, torch.float32, "voxel_ety")), _p(inverse),
              _p(_chk(frame_start, torch.int64, "frame_start")), F, _p(_chk(frame_seg, torch.int32, "frame_seg")), Wmax,
              _p(seg_out_start), int(P), _p(out), _stream())
    return out


def st_gather_back_rows(voxel_ety, inverse, seg_row_start, seg_out_start, P: int):
    """lt_st_gather_back_rows: the write-back over row ranges (windows whose frames are not contiguous in the batch)."""
    S = seg_row_start.numel()
    assert seg_out_start.numel() == S + 1
    out = torch.empty(P, dtype=torch.float32, device=inverse.device)
    _lib.call("lt_st_gather_back_rows", _p(_chk(voxel_ety, torch.float32, "voxel_ety")), _p(inverse),
              _p(_chk(seg_row_start, torch.int64, "seg_row_start")), _p(_chk(seg_out_start, torch.int64, "seg_out_start")),
              S, int(P), _p(out), _stream())
    return out


def st_join(a_keys, t_keys):
    """sorted int64 key arrays -> (a_sel int32 [M], t_sel int32 [M]) matched voxel ids, ascending."""
    a_keys = _chk(a_keys, torch.int64, "st_join a_keys")
    t_keys = _chk(t_keys, torch.int64, "st_join t_keys")
    Va, Vt = a_keys.numel(), t_keys.numel()
    lib = _lib.load()
    ws = _ws(lib.lt_st_join_workspace_bytes(Va, Vt), a_keys.device)
    cap = max(min(Va, Vt), 1)
    a_sel = torch.empty(max(Va, 1), dtype=torch.int32, device=a_keys.device)
    t_sel = torch.empty(max(Vt, 1), dtype=torch.int32, device=a_keys.device)
    m = ctypes.c_int64(0)
    _lib.call("lt_st_join", _p(a_keys), Va, _p(t_keys), Vt, _p(ws), ws.numel(), _p(a_sel), _p(t_sel), ctypes.byref(m),
              _stream())
    del cap
    return a_sel[:m.value], t_sel[:m.value]


def class_quantile_mask(ety, label, num_classes: int, q: float):
    ety = _chk(ety, torch.float32, "class_quantile ety")
    label = _chk(label, torch.int32, "class_quantile label")
    n = ety.numel()
    lib = _lib.load()
    ws = _ws(lib.lt_class_quantile_workspace_bytes(n), ety.device)
    mask = torch.empty(n, dtype=torch.uint8, device=ety.device)
    _lib.call("lt_class_quantile_mask", _p(ety), _p(label), n, int(num_classes), float(q), _p(ws), ws.numel(), _p(mask),
              _stream())
    return mask.bool()


def class_median_mask(vals, label, num_classes: int, cap: float):
    """keep[j] = !(vals[j] < min(median of vals over j's class, cap)) -- refine_pseudo_labels' filter."""
    vals = _chk(vals, torch.float32, "class_median vals")
    label = _chk(label, torch.int32, "class_median label")
    n = vals.numel()
    lib = _lib.load()
    ws = _ws(lib.lt_class_quantile_workspace_bytes(n), vals.device)
    keep = torch.empty(n, dtype=torch.uint8, device=vals.device)
    _lib.call("lt_class_median_mask", _p(vals), _p(label), n, int(num_classes), float(cap), _p(ws), ws.numel(), _p(keep),
              _stream())
    return keep.bool()


def quantile(x: torch.Tensor, q: float) -> torch.Tensor:
    """torch.quantile(x, q) of a 1-D float tensor as a 0-d device tensor (lt_quantile: no sync)."""
    x = _chk(x.reshape(-1), torch.float32, "quantile x")
    n = x.numel()
    lib = _lib.load()
    ws = _ws(lib.lt_quantile_workspace_bytes(n), x.device)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _lib.call("lt_quantile", _p(x), n, float(q), _p(ws), ws.numel(), _p(out), _stream())
    return out[0]


def pseudo_label_fuse(logits_2d, logits_3d, fov_idx, st_ety_2d, st_ety_3d, thr_2d=None, thr_3d=None):
    """-> (pc_2d [Nf], pc_3d [N], ps_conf [N], ps_label int32 [N]); see lt_pseudo_label_fuse.  thr_*: 0-d device tensors."""
    l2 = _chk(logits_2d, torch.float32, "pl logits_2d")
    l3 = _chk(logits_3d, torch.float32, "pl logits_3d")
    fi = _chk(fov_idx, torch.int64, "pl fov_idx")
    s2 = _chk(st_ety_2d, torch.float32, "pl st_ety_2d")
    s3 = _chk(st_ety_3d, torch.float32, "pl st_ety_3d")
    N, C = l3.shape
    Nf = l2.shape[0]
    assert fi.numel() == Nf and s2.numel() == Nf and s3.numel() == N and l2.shape[1] == C
    dev = l3.device
    t2 = None if thr_2d is None else _chk(thr_2d.reshape(1), torch.float32, "pl thr_2d")
    t3 = None if thr_3d is None else _chk(thr_3d.reshape(1), torch.float32, "pl thr_3d")
    probs3 = torch.empty((N, C), dtype=torch.float32, device=dev)
    pc2 = torch.empty(Nf, dtype=torch.float32, device=dev)
    pc3 = torch.empty(N, dtype=torch.float32, device=dev)
    conf = torch.empty(N, dtype=torch.float32, device=dev)
    label = torch.empty(N, dtype=torch.int32, device=dev)
    _lib.call("lt_pseudo_label_fuse", _p(l2), _p(l3), _p(fi), _p(s2), _p(s3), _p(t2), _p(t3), N, Nf, C, _p(probs3),
              _p(pc2), _p(pc3), _p(conf), _p(label), _stream())
    return pc2, pc3, conf, label


def cr_loss_fwd(ac, tr, anchor_start, B: int, keep=None, ac_pc_ety=None, ac_p2v=None, tr_pc_ety=None, tr_p2v=None):
    """-> (loss [1], row_w [M], anchor_stats [B,2]); see lt_cr_loss_fwd.  keep: uint8 / bool [M] or None."""
    ac = _chk(ac, torch.float32, "cr ac")
    tr = _chk(tr, torch.float32, "cr tr")
    M, C = ac.shape
    assert tr.shape == ac.shape and anchor_start.numel() == B + 1
    dev = ac.device
    row_w = torch.empty(M, dtype=torch.float32, device=dev)
    row_loss = torch.empty(M, dtype=torch.float32, device=dev)
    stats = torch.empty((B, 2), dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    if keep is not None:
        keep = _chk(keep.view(torch.uint8) if keep.dtype == torch.bool else keep, torch.uint8, "cr keep")
        assert keep.numel() >= M
    ety = ac_pc_ety is not None
    _lib.call("lt_cr_loss_fwd", _p(ac), _p(tr), M, int(C),
              _p(_chk(ac_pc_ety, torch.float32, "cr ac_pc_ety")) if ety else None,
              _p(_chk(ac_p2v, torch.int64, "cr ac_p2v")) if ety else None,
              _p(_chk(tr_pc_ety, torch.float32, "cr tr_pc_ety")) if ety else None,
              _p(_chk(tr_p2v, torch.int64, "cr tr_p2v")) if ety else None,
              _p(keep), _p(_chk(anchor_start, torch.int32, "cr anchor_start")), int(B), _p(row_w), _p(row_loss), _p(stats),
              _p(loss), _stream())
    return loss, row_w, stats, keep


def cr_loss_bwd(ac, tr, row_w, keep, row_anchor, stats, B: int, gloss):
    M, C = ac.shape
    g_ac = torch.empty_like(ac)
    _lib.call("lt_cr_loss_bwd", _p(ac), _p(tr), M, int(C), _p(row_w), _p(keep),
              _p(_chk(row_anchor, torch.int64, "cr row_anchor")), _p(stats), int(B),
              _p(_chk(gloss.reshape(1), torch.float32, "cr gloss")), _p(g_ac), _stream())
    return g_ac


# ------------------------------------------------------------------ conv -> BN [+ residual] [+ ReLU] composites
def conv_bn_fwd(feats, packed, plan, K: int, Vout: int, cout: int, gamma, beta, eps: float, residual, relu: bool,
                keep_conv_out: bool = True):
    """-> (conv_out, y, mean, var).  One host call (lt_conv_bn_fwd): conv with statistics epilogue, finalize, BN apply.
    keep_conv_out False (no backward will come): BatchNorm is applied in place, conv_out is y."""
    Vin, Cin = feats.shape
    dev = feats.device
    conv_out = torch.empty((Vout, cout), dtype=torch.float32, device=dev)
    y = torch.empty((Vout, cout), dtype=torch.float32, device=dev) if keep_conv_out else conv_out
    stats = torch.empty((2, cout), dtype=torch.float32, device=dev)
    ws = _scratch(_ws_bytes("lt_conv_bn_workspace_bytes", Vout, cout), dev)
    mean, var = stats[0], stats[1]
    from . import functional as LF
    if LF._DIAG_NO_APPLY:   # diagnostics: conv + statistics, no apply pass (y = the raw conv output)
        _lib.call("lt_spconv_plan_fwd_stats", _p(feats), _p(packed), _p(plan), _p(conv_out), Vin, Vout, Cin, int(cout), K, _p(ws),
                  _stream())
        _lib.call("lt_bn_finalize_sums", _p(ws), Vout, int(cout), _p(mean), _p(var), _stream())
        return conv_out, conv_out, mean, var
    _lib.call("lt_conv_bn_fwd", _p(feats), _p(packed), _p(plan), Vin, Vout, Cin, int(cout), K, _p(gamma), _p(beta),
              float(eps), _p(residual), int(relu), _p(ws), ws.numel(), _p(conv_out), _p(y), _p(mean), _p(var), _stream())
    return conv_out, y, mean, var


def conv_bn_bwd(dy, conv_out, y, mean, var, gamma, eps: float, relu: bool, need_dres: bool, packed_bwd, plan_bwd,
                K: int, Vin: int, cin: int, beta=None):
    """-> (din | None, dgamma, dbeta, dres | None); din is computed when packed_bwd is given.  y None with relu: the
    ReLU mask is rebuilt from conv_out (needs beta; the forward added no residual)."""
    Vout, cout = conv_out.shape
    dev = conv_out.device
    ws = _scratch(_ws_bytes("lt_conv_bn_workspace_bytes", Vout, cout), dev)
    dgb = torch.empty((2, cout), dtype=torch.float32, device=dev)
    dgamma, dbeta = dgb[0], dgb[1]
    dres = torch.empty_like(conv_out) if need_dres else None
    need_din = packed_bwd is not None
    dconv = torch.empty_like(conv_out) if need_din else None
    din = torch.empty((Vin, cin), dtype=torch.float32, device=dev) if need_din else None
    mode = _relu_mode(relu, y, beta)
    yp = _p(y) if mode == 1 else None
    _lib.call("lt_conv_bn_bwd", _p(dy), _p(conv_out), yp, _p(mean), _p(var), _p(gamma), _p(beta), float(eps), mode, Vout,
              cout, _p(ws), ws.numel(), _p(dgamma), _p(dbeta), _p(dres), _p(dconv), _p(packed_bwd), _p(plan_bwd),
              Vin, cin, K, _p(din), _stream())
    return din, dgamma, dbeta, dres


# ------------------------------------------------------------------ parameter update
def adam_ema_step(param, grad, exp_avg, exp_avg_sq, ema, lr: float, beta1: float, beta2: float, eps: float,
                  weight_decay: float, step: int, ema_alpha: float, grad_scale: float = 1.0) -> None:
    """In place over flat fp32 buffers: torch.optim.Adam's step, then ema = alpha*ema + (1-alpha)*param (lt_adam_ema_step)."""
    n = param.numel()
    for name, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq), ("ema", ema)):
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
            raise ValueError(f"adam_ema_step: {name} must be a contiguous fp32 GPU buffer of {n} elements")
    _lib.call("lt_adam_ema_step", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(ema), n, float(lr), float(beta1),
              float(beta2), float(eps), float(weight_decay), int(step), float(ema_alpha), float(grad_scale), _stream())


# ------------------------------------------------------------------ heads + inverse_to_all
def head_gather_fwd(feats, idx, packed, bias, c1: int, c2: int):
    """-> (out1 [N,c1], out2 [N,c2] | None): both linear heads at the rows feats[idx] (lt_head_gather_fwd)."""
    feats = _chk(feats, torch.float32, "head feats")
    V, Cin = feats.shape
    if idx is not None:
        idx = _chk(idx, torch.int64, "head idx")
        N = idx.numel()
    else:
        N = V
    out1 = torch.empty((N, c1), dtype=torch.float32, device=feats.device)
    out2 = torch.empty((N, c2), dtype=torch.float32, device=feats.device) if c2 > 0 else None
    _lib.call("lt_head_gather_fwd", _p(feats), _p(idx), _p(_chk(packed, torch.float32, "head packed")),
              _p(_chk(bias, torch.float32, "head bias")), N, V, int(Cin), int(c1), int(c2), _p(out1), _p(out2), _stream())
    return out1, out2
