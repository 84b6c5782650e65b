"""ctypes binding of oracle/librlap_oracle.so (test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librlap_oracle.so")
_lib = None

O_V = {"random": 0, "degree": 1, "coarsen": 2}
O_N = {"asc": 0, "desc": 1, "random": 2}


class Stats(ctypes.Structure):
    _fields_ = [
        ("n_eliminated", ctypes.c_int64),
        ("n_draws", ctypes.c_int64),
        ("n_pq_moves", ctypes.c_int64),
        ("nnz", ctypes.c_int64),
        ("t_setup", ctypes.c_double),
        ("t_elim", ctypes.c_double),
        ("t_output", ctypes.c_double),
        ("t_total", ctypes.c_double),
    ]


def build(force=False):
    src = os.path.join(_HERE, "rlap_oracle.cc")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "librlap_oracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        lib.rlap_oracle_approx_chol.restype = ctypes.c_int
        lib.rlap_oracle_approx_chol.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
            ctypes.POINTER(ctypes.POINTER(ctypes.c_double)), ctypes.POINTER(ctypes.c_int64),
            ctypes.c_void_p, ctypes.POINTER(Stats),
        ]
        lib.rlap_oracle_free.argtypes = [ctypes.POINTER(ctypes.c_double)]
        lib.rlap_oracle_uniforms.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        lib.rlap_oracle_stdsort_perm.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
        lib.rlap_oracle_heapsort_perm.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
        lib.rlap_oracle_keyed_keys.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        _lib = lib
    return _lib


def approximate_cholesky(edge_index, edge_weights, num_nodes, num_remove, o_v, o_n, *, perm=None,
                         shuffle_seed=0, sort="libstdcxx", faithful=True, return_order=False,
                         return_stats=False, mode="exact"):
    """Oracle with the reference's signature (rlap/ops.py:7-14) on numpy arrays.

    edge_index (2,E) integer, edge_weights (1,E)/(E,)/None. Returns (m,3) float64.
    """
    lib = _load()
    ei = np.asarray(edge_index)
    assert ei.shape[0] == 2
    E = ei.shape[1]
    w = np.ones(E, dtype=np.float64) if edge_weights is None else np.asarray(edge_weights, dtype=np.float64).reshape(-1)
    info = np.empty((E, 3), dtype=np.float64)
    info[:, 0] = ei[0]
    info[:, 1] = ei[1]
    info[:, 2] = w
    assert o_v in O_V and o_n in O_N
    perm_arr = None
    if perm is not None:
        perm_arr = np.ascontiguousarray(np.asarray(perm, dtype=np.int64))
        assert perm_arr.shape[0] == num_nodes
    out = ctypes.POINTER(ctypes.c_double)()
    rows = ctypes.c_int64(0)
    st = Stats()
    order = np.full(max(int(num_nodes), 1), -1, dtype=np.int64)
    rc = lib.rlap_oracle_approx_chol(
        info.ctypes.data, E, int(num_nodes), int(num_remove), O_V[o_v], O_N[o_n],
        perm_arr.ctypes.data if perm_arr is not None else None, int(shuffle_seed) & (2**64 - 1),
        (0 if sort == "libstdcxx" else 1) | (16 if mode == "frontier" else 0), 1 if faithful else 0,
        ctypes.byref(out), ctypes.byref(rows), order.ctypes.data, ctypes.byref(st))
    if rc == 1:
        raise ValueError("adjacency matrix is not symmetric")
    if rc != 0:
        raise ValueError(f"oracle error {rc}")
    m = rows.value
    res = np.ctypeslib.as_array(out, shape=(max(m, 1) * 3,))[: 3 * m].copy().reshape(m, 3)
    lib.rlap_oracle_free(out)
    ret = [res]
    if return_order:
        ret.append(order[: int(num_nodes)])
    if return_stats:
        ret.append({f: getattr(st, f) for f, _ in Stats._fields_})
    return ret[0] if len(ret) == 1 else tuple(ret)


def uniforms(count):
    lib = _load()
    u = np.empty(count, dtype=np.float64)
    raw = np.empty(count, dtype=np.uint64)
    lib.rlap_oracle_uniforms(count, u.ctypes.data, raw.ctypes.data)
    return u, raw


def stdsort_perm(keys, desc=False):
    lib = _load()
    k = np.ascontiguousarray(np.asarray(keys, dtype=np.float64))
    p = np.empty(k.shape[0], dtype=np.int64)
    lib.rlap_oracle_stdsort_perm(k.ctypes.data, k.shape[0], 1 if desc else 0, p.ctypes.data)
    return p


def heapsort_perm(keys, desc=False):
    lib = _load()
    k = np.ascontiguousarray(np.asarray(keys, dtype=np.float64))
    p = np.empty(k.shape[0], dtype=np.int64)
    lib.rlap_oracle_heapsort_perm(k.ctypes.data, k.shape[0], 1 if desc else 0, p.ctypes.data)
    return p


def keyed_order(seed, vertex, phase, nbrs):
    """Positions of the injected neighbour order: returns the neighbours' indices in the order they are visited."""
    lib = _load()
    nb = np.ascontiguousarray(np.asarray(nbrs, dtype=np.int64))
    k = np.empty(nb.shape[0], dtype=np.float64)
    lib.rlap_oracle_keyed_keys(int(seed) & (2**64 - 1), int(vertex), int(phase), nb.ctypes.data, nb.shape[0], k.ctypes.data)
    return stdsort_perm(k), k
