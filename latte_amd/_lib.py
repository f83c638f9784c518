"""ctypes binding of liblatte_hip.so (the C-ABI declared in include/latte_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LATTE_HIP_LIB: another build of the same library (A/B measurements of kernel changes on one GPU box)
LIB_PATH = os.environ.get("LATTE_HIP_LIB") or os.path.join(_HERE, "liblatte_hip.so")

P = ctypes.c_void_p
I64 = ctypes.c_int64
I32 = ctypes.c_int32
F32 = ctypes.c_float
F64 = ctypes.c_double
PI64 = ctypes.POINTER(ctypes.c_int64)

# name -> (restype, argtypes); must list every symbol include/latte_hip.h declares
SIGNATURES = {
    "lt_last_error": (ctypes.c_char_p, []),
    "lt_version": (ctypes.c_int, []),
    "lt_sphash": (ctypes.c_int, [P, I64, P, I32, P, P]),
    "lt_table_bytes": (I64, [I64]),
    "lt_hash_build": (ctypes.c_int, [P, I64, P, I64, P]),
    "lt_hash_query": (ctypes.c_int, [P, I64, P, I64, P, P]),
    "lt_spcount": (ctypes.c_int, [P, I64, P, I64, P]),
    "lt_unique_workspace_bytes": (I64, [I64]),
    "lt_unique_sorted": (ctypes.c_int, [P, I64, I32, P, I64, P, P, P, PI64, P]),
    "lt_pack_xyz_keys": (ctypes.c_int, [P, I64, P, P]),
    "lt_downsample_workspace_bytes": (I64, [I64]),
    "lt_downsample_coords": (ctypes.c_int, [P, I64, I32, P, I64, P, PI64, P]),
    "lt_kmap_workspace_bytes": (I64, [I64]),
    "lt_kmap_build": (ctypes.c_int, [P, I64, P, I64, P, I32, P, I64, P, P]),
    "lt_kmap_build_sub": (ctypes.c_int, [P, I64, P, I32, P, I64, P, P]),
    "lt_kmap_pairs_workspace_bytes": (I64, [I64, I64]),
    "lt_kmap_pairs": (ctypes.c_int, [P, I32, I64, P, I64, P, P, PI64, P]),
    "lt_kmap_reverse": (ctypes.c_int, [P, I32, I64, I64, P, P]),
    "lt_kmap_order_workspace_bytes": (I64, [I64]),
    "lt_kmap_order": (ctypes.c_int, [P, I32, I64, P, I64, P, P]),
    "lt_kmap_plan_bytes": (I64, [I32, I64]),
    "lt_kmap_plan_workspace_bytes": (I64, [I64]),
    "lt_kmap_plan": (ctypes.c_int, [P, I32, I64, P, I64, P, I64, P]),
    "lt_spconv_plan_fwd": (ctypes.c_int, [P, P, P, P, I64, I64, I32, I32, I32, P]),
    "lt_spconv_plan_fwd_v": (ctypes.c_int, [P, P, P, P, I64, I64, I32, I32, I32, I32, P]),
    "lt_spconv_stat_blocks": (I64, [I64, I32]),
    "lt_conv_set_precision": (ctypes.c_int, [I32]),
    "lt_conv_uses_bf16": (ctypes.c_int, [I32, I32]),
    "lt_conv_timer_begin": (ctypes.c_int, [I32]),
    "lt_conv_timer_arm": (ctypes.c_int, [I32]),
    "lt_conv_timer_end": (I64, [P, P, I64]),
    "lt_conv_debug_stamps": (ctypes.c_int, [P, I64]),
    "lt_spconv_plan_fwd_stats": (ctypes.c_int, [P, P, P, P, I64, I64, I32, I32, I32, P, P]),
    "lt_bn_finalize_sums": (ctypes.c_int, [P, I64, I32, P, P, P]),
    "lt_spconv_stats_bytes": (I64, [I64, I32]),
    "lt_conv_bn_workspace_bytes": (I64, [I64, I32]),
    "lt_conv_bn_fwd": (ctypes.c_int, [P, P, P, I64, I64, I32, I32, I32, P, P, F32, P, I32, P, I64, P, P, P, P, P]),
    "lt_conv_bn_bwd": (ctypes.c_int, [P, P, P, P, P, P, P, F32, I32, I64, I32, P, I64, P, P, P, P, P, P, I64, I32, I32, P,
                                       P]),
    "lt_spconv_packed_floats": (I64, [I32, I32, I32]),
    "lt_spconv_pack_weights": (ctypes.c_int, [P, I32, I32, I32, I32, P, P]),
    "lt_spconv_fwd": (ctypes.c_int, [P, P, P, P, P, I64, I64, I32, I32, I32, P]),
    "lt_spconv_stem_fwd": (ctypes.c_int, [P, P, P, P, I64, I64, I32, I32, I32, P]),
    "lt_spconv_bwd_weight_workspace_bytes": (I64, [I64, I32, I32, I32]),
    "lt_spconv_bwd_weight": (ctypes.c_int, [P, P, P, P, I64, I64, I64, I32, I32, I32, I32, P, I64, P, P]),
    "lt_bn_workspace_bytes": (I64, [I64, I32]),
    "lt_bn_stats": (ctypes.c_int, [P, I64, I32, P, I64, P, P, P]),
    "lt_bn_stats_static": (ctypes.c_int, [P, I64, I32, P, P, P]),
    "lt_bn_apply": (ctypes.c_int, [P, P, P, P, P, F32, P, I32, P, I64, I32, P]),
    "lt_bn_apply_devox": (ctypes.c_int, [P, P, P, P, P, F32, P, P, P, P, I64, I32, I64, P]),
    "lt_bn_bwd_reduce": (ctypes.c_int, [P, P, P, P, P, P, P, F32, I32, I64, I32, P, I64, P, P, P]),
    "lt_bn_bwd_apply": (ctypes.c_int, [P, P, P, P, P, P, P, F32, I32, P, P, P, P, I64, I32, P]),
    "lt_head_gather_fwd": (ctypes.c_int, [P, P, P, P, I64, I64, I32, I32, I32, P, P, P]),
    "lt_adam_ema_step": (ctypes.c_int, [P, P, P, P, P, I64, F32, F32, F32, F32, F32, I64, F32, F32, P]),
    "lt_voxelize_fwd": (ctypes.c_int, [P, P, P, P, I64, I32, I64, P]),
    "lt_voxelize_bwd": (ctypes.c_int, [P, P, P, P, I64, I32, I64, P]),
    "lt_calc_ti_weights": (ctypes.c_int, [P, P, I64, I32, P, P]),
    "lt_devoxelize_fwd": (ctypes.c_int, [P, P, P, P, I64, I32, I64, P]),
    "lt_devoxelize_bwd": (ctypes.c_int, [P, P, P, P, I64, I32, I64, P]),
    "lt_csr_workspace_bytes": (I64, [I64]),
    "lt_csr_build": (ctypes.c_int, [P, I64, I64, P, I64, P, P, P]),
    "lt_voxelize_csr_fwd": (ctypes.c_int, [P, P, P, P, I64, I32, P]),
    "lt_devoxelize_csr_bwd": (ctypes.c_int, [P, P, P, P, P, I64, I32, P]),
    "lt_devoxelize_cells": (ctypes.c_int, [P, P, P, P, I64, I64, I64, P]),
    "lt_devoxelize_cell_bwd": (ctypes.c_int, [P, P, P, P, P, P, P, P, I64, I64, I32, P]),
    # include/latte_hip_st.h
    "lt_softmax_entropy": (ctypes.c_int, [P, I64, I32, P, P, P]),
    "lt_softmax_bwd": (ctypes.c_int, [P, P, I64, I32, P, P]),
    "lt_st_keys": (ctypes.c_int, [P, P, P, P, P, I32, I64, F64, F64, I32, I32, P, P, P]),
    "lt_sort_segments_workspace_bytes": (I64, [I64]),
    "lt_sort_segments": (ctypes.c_int, [P, I64, I32, P, I64, P, P, P, P, PI64, P]),
    "lt_segment_reduce": (ctypes.c_int, [P, I32, P, P, P, I64, I32, P, P, P]),
    "lt_segment_reduce_bwd": (ctypes.c_int, [P, I32, P, P, I64, I32, P, P]),
    "lt_st_gather_back": (ctypes.c_int, [P, P, P, I32, P, I32, P, I64, P, P]),
    "lt_st_gather_back_rows": (ctypes.c_int, [P, P, P, P, I32, I64, P, P]),
    "lt_st_join_workspace_bytes": (I64, [I64, I64]),
    "lt_st_join": (ctypes.c_int, [P, I64, P, I64, P, I64, P, P, PI64, P]),
    "lt_class_quantile_workspace_bytes": (I64, [I64]),
    "lt_class_quantile_mask": (ctypes.c_int, [P, P, I64, I32, F32, P, I64, P, P]),
    "lt_class_median_mask": (ctypes.c_int, [P, P, I64, I32, F32, P, I64, P, P]),
    "lt_quantile_workspace_bytes": (I64, [I64]),
    "lt_quantile": (ctypes.c_int, [P, I64, F32, P, I64, P, P]),
    "lt_pseudo_label_fuse": (ctypes.c_int, [P, P, P, P, P, P, P, I64, I64, I32, P, P, P, P, P, P]),
    "lt_cr_loss_fwd": (ctypes.c_int, [P, P, I64, I32, P, P, P, P, P, P, I32, P, P, P, P, P]),
    "lt_cr_loss_bwd": (ctypes.c_int, [P, P, I64, I32, P, P, P, P, I32, P, P, P]),
}

# include/latte_hip_net.h: the backbone pass as one call (host-side structs, see latte_amd/netprog.py)
class NetOp(ctypes.Structure):
    _fields_ = [("kind", I32), ("in0", I32), ("in1", I32), ("out", I32), ("layer", I32), ("map", I32), ("level", I32),
                ("relu", I32)]


class NetBuf(ctypes.Structure):
    _fields_ = [("rows", I32), ("channels", I32)]


class NetLayer(ctypes.Structure):
    _fields_ = [("w_fwd", P), ("w_bwd", P), ("gamma", P), ("beta", P), ("eps", F32), ("cin", I32), ("cout", I32),
                ("K", I32), ("dgamma_off", I64), ("dbeta_off", I64)]


class NetProg(ctypes.Structure):
    _fields_ = [("ops", ctypes.POINTER(NetOp)), ("n_ops", I32), ("bufs", ctypes.POINTER(NetBuf)), ("n_bufs", I32),
                ("layers", ctypes.POINTER(NetLayer)), ("n_layers", I32), ("out_buf", I32)]


class NetMaps(ctypes.Structure):
    _fields_ = [("n_points", I64), ("n_vox", I64 * 5), ("plan_k3", P * 5), ("plan_down", P * 4), ("plan_up", P * 4),
                ("plan_dense", P * 6), ("p2v_perm", P * 5), ("p2v_off", P * 5), ("p2v_idx", P * 5), ("p2v_counts", P * 5),
                ("v2p_idx", P * 5), ("v2p_w", P * 5), ("v2p_cells", P * 5)]


_SIGS = {
    "lt_kmap_plan_identity": (ctypes.c_int, [I64, P, I64, P]),
    "lt_net_arena_floats": (I64, [ctypes.POINTER(NetProg), ctypes.POINTER(NetMaps), P, I32]),
    "lt_net_forward": (ctypes.c_int, [ctypes.POINTER(NetProg), ctypes.POINTER(NetMaps), P, P, I64, I32, P, P]),
    "lt_net_grad_arena_floats": (I64, [ctypes.POINTER(NetProg), ctypes.POINTER(NetMaps), P]),
    "lt_net_backward": (ctypes.c_int, [ctypes.POINTER(NetProg), ctypes.POINTER(NetMaps), P, P, I64, P, P, P, I64, P]),
}
SIGNATURES.update(_SIGS)


_lib = None


class LatteHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load liblatte_hip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LatteHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C latte_amd/csrc). There is no CPU fallback for the product path.")
    # torch ships its own libamdhip64.so.7; it must be the HIP runtime of the process BEFORE this library is mapped,
    # so that both resolve the same SONAME to ONE runtime (streams and device pointers are shared with torch).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    """Call an int-returning entry point and raise with lt_last_error() on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.lt_last_error()
        raise LatteHipError(f"{name} failed (rc={rc}): {msg.decode() if msg else ''}")
