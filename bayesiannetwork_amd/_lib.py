"""ctypes loader for libbn_mi355x.so (the C ABI declared in include/bn_mi355x.h).

The library is the product: if it is missing or cannot be loaded this module raises --
there is no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# BN_MI355X_LIB selects another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("BN_MI355X_LIB") or os.path.join(_HERE, "libbn_mi355x.so")
_LIB = None

BN_OK, BN_ERR_ARG, BN_ERR_HIP, BN_ERR_NO_DEVICE, BN_ERR_ALLOC, BN_ERR_COMM, BN_ERR_STATE = 0, -1, -2, -3, -4, -5, -6
BN_DEVICE_HOST_ONLY, BN_DEVICE_CURRENT = -2, -1
BN_MAX_BATCH_SETS = 256  # include/bn_mi355x.h

i32p = ctypes.POINTER(ctypes.c_int32)
i64p = ctypes.POINTER(ctypes.c_int64)
f64p = ctypes.POINTER(ctypes.c_double)
u8p = ctypes.POINTER(ctypes.c_uint8)


class ModelDesc(ctypes.Structure):
    _fields_ = [("n_nodes", ctypes.c_int32), ("k", i32p), ("in_ptr", i32p), ("in_idx", i32p),
                ("cpt_off", i64p), ("cpt", f64p), ("device", ctypes.c_int32), ("lanes_per_node", ctypes.c_int32)]


class BpStats(ctypes.Structure):
    _fields_ = [("sweeps", ctypes.c_int32), ("sweep_launches", ctypes.c_int32), ("sweep_kernel_ms", ctypes.c_float),
                ("total_ms", ctypes.c_float), ("algorithmic_bytes_per_sweep", ctypes.c_int64),
                ("layout_bytes_per_sweep", ctypes.c_int64), ("messages_per_sweep", ctypes.c_int64),
                ("sweep_devclock_ms", ctypes.c_float), ("resident_aborts", ctypes.c_int32)]


class LayoutInfo(ctypes.Structure):
    _fields_ = [("n_nodes", ctypes.c_int32), ("n_edges", ctypes.c_int32), ("n_classes", ctypes.c_int32),
                ("n_tiles", ctypes.c_int32), ("lanes_per_node_max", ctypes.c_int32),
                ("cpt_doubles", ctypes.c_int64), ("rec_doubles", ctypes.c_int64), ("node_doubles", ctypes.c_int64),
                ("algorithmic_bytes_per_sweep", ctypes.c_int64), ("layout_bytes_per_sweep", ctypes.c_int64),
                ("messages_per_sweep", ctypes.c_int64), ("rank", ctypes.c_int32), ("nranks", ctypes.c_int32),
                ("n_owned", ctypes.c_int32), ("n_interior_tiles", ctypes.c_int32), ("n_cut_edges", ctypes.c_int64),
                ("segment_bytes", ctypes.c_int64), ("segment_used_bytes", ctypes.c_int64),
                ("exchange_base", ctypes.c_int64)]


# every symbol include/bn_mi355x.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("bn_create", ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.POINTER(ctypes.c_void_p)]),
    ("bn_create_sharded", ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_int32, ctypes.c_int32, i32p,
                                         ctypes.POINTER(ctypes.c_void_p)]),
    ("bn_peer_blob_size", ctypes.c_int64, [ctypes.c_void_p]),
    ("bn_peer_export", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    ("bn_peer_import", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), i64p, ctypes.c_int32]),
    ("bn_layout_flow", ctypes.c_int, [ctypes.c_void_p, i32p, ctypes.POINTER(ctypes.c_uint32)]),
    ("bn_mid_plan_get", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, i32p, ctypes.POINTER(ctypes.c_uint32), f64p, ctypes.POINTER(ctypes.c_uint32),
                                       ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), f64p]),
    ("bn_small_plan_get", ctypes.c_int, [ctypes.c_void_p, i32p, ctypes.POINTER(ctypes.c_uint32), f64p, ctypes.POINTER(ctypes.c_uint32),
                                         ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                                         f64p]),
    ("bn_reload_cpt", ctypes.c_int, [ctypes.c_void_p, f64p, ctypes.c_int64]),
    ("bn_dag_plan_get", ctypes.c_int, [ctypes.c_void_p, i32p, i32p, i32p, i32p, i32p, i32p, f64p, f64p]),
    ("bn_comm_unique_id", ctypes.c_int, [ctypes.c_void_p]),
    ("bn_comm_init", ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    ("bn_destroy", None, [ctypes.c_void_p]),
    ("bn_last_error", ctypes.c_char_p, []),
    ("bn_version", ctypes.c_char_p, []),
    ("bn_bp_run", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, i32p, i32p, f64p, ctypes.c_double, ctypes.c_int32,
                                 f64p, i32p, f64p]),
    ("bn_bp_run_view", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, i32p, i32p, f64p, ctypes.c_double, ctypes.c_int32,
                                      ctypes.POINTER(f64p), i32p, f64p]),
    ("bn_bp_set_evidence", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, i32p, i32p, f64p]),
    ("bn_bp_run_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_int32, i32p, f64p]),
    ("bn_bp_beliefs_device", ctypes.c_void_p, [ctypes.c_void_p]),
    ("bn_bp_copy_beliefs", ctypes.c_int, [ctypes.c_void_p, f64p]),
    ("bn_bp_residual_history", ctypes.c_int, [ctypes.c_void_p, f64p, ctypes.c_int32]),
    ("bn_bp_run_batch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, i32p, i32p, i32p, f64p, ctypes.c_double, ctypes.c_int32,
                                       f64p, i32p, f64p]),
    ("bn_bp_set_evidence_batch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, i32p, i32p, i32p, f64p]),
    ("bn_bp_run_batch_device", ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_int32, i32p, f64p]),
    ("bn_bp_copy_beliefs_batch", ctypes.c_int, [ctypes.c_void_p, f64p]),
    ("bn_bp_residual_history_batch", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, f64p, ctypes.c_int32]),
    ("bn_bp_messages", ctypes.c_int, [ctypes.c_void_p, f64p, f64p]),
    ("bn_set_option", ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int32]),
    ("bn_bp_last_path", ctypes.c_int, [ctypes.c_void_p]),
    ("bn_get_info", ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p]),
    ("bn_bp_step_begin", ctypes.c_int, [ctypes.c_void_p]),
    ("bn_bp_step_sweep", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_double]),
    ("bn_bp_step_sweep_part", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_double, ctypes.c_int32]),
    ("bn_bp_step_finish", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, i32p, i32p,
                                         f64p]),
    ("bn_debug_allgather", ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.c_int32]),
    ("bn_debug_stream", ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, f64p]),
    ("bn_bp_last_stats", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(BpStats)]),
    ("bn_lw_run", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, i32p, i32p, ctypes.c_uint64, ctypes.c_uint64,
                                 ctypes.c_uint64, f64p]),
    ("bn_lw_run_allreduce", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, i32p, i32p, ctypes.c_uint64, ctypes.c_uint64,
                                           ctypes.c_uint64, f64p]),
    ("bn_rs_run", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, i32p, i32p, ctypes.c_uint64, ctypes.c_uint64,
                                 ctypes.c_uint64, ctypes.c_uint64, f64p, ctypes.POINTER(ctypes.c_uint64),
                                 ctypes.POINTER(ctypes.c_uint64)]),
    ("bn_fit_cpt", ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_int64, u8p, ctypes.POINTER(ctypes.c_uint64), f64p]),
    ("bn_lw_states", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, u8p, f64p]),
    ("bn_layout_get", ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(LayoutInfo)]),
    ("bn_layout_edge_refs", ctypes.c_int, [ctypes.c_void_p, i32p, i32p]),
    ("bn_layout_node_slots", ctypes.c_int, [ctypes.c_void_p, i32p]),
    ("bn_layout_node_tiles", ctypes.c_int, [ctypes.c_void_p, i32p]),
    ("bn_layout_class", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, i32p, i32p, i32p, i32p, i32p]),
]


def build(quiet: bool = True) -> str:
    """Compile every HIP source for gfx950 into the in-tree shared library."""
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4", "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError = the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


class BnError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"bn_mi355x error {code}: {msg}")
        self.code = code


def check(rc: int) -> int:
    if rc < 0:
        raise BnError(rc, lib().bn_last_error().decode("utf-8", "replace"))
    return rc
