"""ctypes loader for librlap_hip.so (the C ABI in include/rlap_hip.h).

The product path has no CPU fallback: if the HIP library is missing, or no GPU is
visible, calls raise instead of silently computing somewhere else.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RLAP_AMD_LIB") or os.path.join(_HERE, "librlap_hip.so")   # (override: kernel-variant experiments)

_lib = None

# every symbol include/rlap_hip.h declares
EXPORTS = [
    "rlap_create", "rlap_destroy", "rlap_set_stream", "rlap_set_timing", "rlap_status_string",
    "rlap_identity", "rlap_unpack_edge_info", "rlap_approx_chol", "rlap_approx_chol_batched",
    "rlap_rng_uniforms", "rlap_util_ba_graph", "rlap_debug_wave_sort",
    "rlap_approx_chol_from_edges", "rlap_debug_set_limits", "rlap_pack_rows", "rlap_unpack_rows",
    "rlap_workspace_bytes", "rlap_workspace_query", "rlap_set_workspace", "rlap_workspace_needed", "rlap_debug_set_poison", "rlap_debug_set_jitter",
    "rlap_set_rng_mode",
]

E_WORKSPACE = 11   # RLAP_E_WORKSPACE


class Stats(ctypes.Structure):
    _fields_ = [
        ("nnz", ctypes.c_int64), ("n_eliminated", ctypes.c_int64), ("n_draws", ctypes.c_int64),
        ("out_rows", ctypes.c_int64), ("live_entries", ctypes.c_int64),
        ("ms_setup", ctypes.c_float), ("ms_elim", ctypes.c_float), ("ms_output", ctypes.c_float),
        ("ms_sc_merge", ctypes.c_float), ("ms_sc_compact", ctypes.c_float), ("ms_total", ctypes.c_float),
        ("n_retries", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("n_rounds", ctypes.c_int64), ("n_singles", ctypes.c_int64),
    ]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def load():
    """dlopen the library and declare prototypes. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"rlap_amd: {LIB_PATH} is missing. Build it with `make -C rlap_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64, u64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int
    lib.rlap_create.restype = ci
    lib.rlap_create.argtypes = [ctypes.POINTER(vp)]
    lib.rlap_destroy.restype = ci
    lib.rlap_destroy.argtypes = [vp]
    lib.rlap_set_stream.restype = ci
    lib.rlap_set_stream.argtypes = [vp, vp]
    lib.rlap_set_timing.restype = ci
    lib.rlap_set_timing.argtypes = [vp, ci]
    lib.rlap_status_string.restype = ctypes.c_char_p
    lib.rlap_status_string.argtypes = [ci]
    lib.rlap_identity.restype = ci
    lib.rlap_identity.argtypes = [vp, vp, vp, vp, i64, i64]
    lib.rlap_unpack_edge_info.restype = ci
    lib.rlap_unpack_edge_info.argtypes = [vp, vp, i64, vp, vp, vp]
    lib.rlap_approx_chol.restype = ci
    lib.rlap_approx_chol.argtypes = [vp, vp, vp, vp, i64, i64, i64, ci, ci, vp, u64, vp, i64,
                                     ctypes.POINTER(i64), ctypes.POINTER(Stats)]
    lib.rlap_approx_chol_batched.restype = ci
    lib.rlap_approx_chol_batched.argtypes = [vp, vp, vp, vp, i64, i64, vp, vp, ci, ci, vp, u64, vp, i64, vp,
                                             ctypes.POINTER(Stats)]
    lib.rlap_approx_chol_from_edges.restype = ci
    lib.rlap_approx_chol_from_edges.argtypes = [vp, vp, vp, vp, i64, i64, i64, ctypes.c_double, ci, ci, ci, vp, u64, vp, i64,
                                                ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(Stats)]
    lib.rlap_debug_set_limits.restype = ci
    lib.rlap_debug_set_limits.argtypes = [vp, ctypes.c_double, ctypes.c_double, i64, i64]
    lib.rlap_pack_rows.restype = ci
    lib.rlap_pack_rows.argtypes = [vp, vp, i64, vp]
    lib.rlap_unpack_rows.restype = ci
    lib.rlap_unpack_rows.argtypes = [vp, vp, i64, vp]
    lib.rlap_rng_uniforms.restype = ci
    lib.rlap_rng_uniforms.argtypes = [vp, i64, vp]
    lib.rlap_set_rng_mode.restype = ci
    lib.rlap_set_rng_mode.argtypes = [vp, ci]
    lib.rlap_debug_wave_sort.restype = ci
    lib.rlap_debug_wave_sort.argtypes = [vp, vp, vp, ctypes.c_int32, ctypes.c_int32, vp]
    sz = ctypes.c_size_t
    lib.rlap_workspace_bytes.restype = ci
    lib.rlap_workspace_bytes.argtypes = [i64, i64, i64, ci, ctypes.POINTER(sz), ctypes.POINTER(i64)]
    lib.rlap_workspace_query.restype = ci
    lib.rlap_workspace_query.argtypes = [vp, i64, i64, i64, ci, ctypes.POINTER(sz), ctypes.POINTER(i64)]
    lib.rlap_set_workspace.restype = ci
    lib.rlap_set_workspace.argtypes = [vp, vp, sz, vp, i64]
    lib.rlap_workspace_needed.restype = ci
    lib.rlap_workspace_needed.argtypes = [vp, ctypes.POINTER(sz), ctypes.POINTER(i64)]
    lib.rlap_debug_set_jitter.restype = ci
    lib.rlap_debug_set_jitter.argtypes = [vp, ci]
    lib.rlap_debug_set_poison.restype = ci
    lib.rlap_debug_set_poison.argtypes = [vp, ci]
    lib.rlap_util_ba_graph.restype = i64
    lib.rlap_util_ba_graph.argtypes = [i64, i64, u64, vp, vp]
    _lib = lib
    return lib


def status_string(code):
    return load().rlap_status_string(int(code)).decode()
