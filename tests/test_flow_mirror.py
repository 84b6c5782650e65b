"""CPU checks of the multi-CU ("dataflow") elimination for o_v = "random" (rlap_amd/csrc/rlap_flow.h): the protocol --
pending counters instead of program order, look-back uniform offsets, tagged out-of-order appends, tag order restored where
ids repeat and in the surviving columns -- run by randomly interleaved virtual waves on the host mirror, bit-exact against the
oracle (row order, indices, weights)."""
import ctypes

import numpy as np
import pytest

import oracle
from util import ba_graph, clique, grid2d, path, star, sym_weights


def mirror_flow(lib, ei, w, n, t, o_n, perm, seed=0, nwaves=8, sched_seed=1):
    E = ei.shape[1]
    row = np.ascontiguousarray(ei[0])
    col = np.ascontiguousarray(ei[1])
    w = np.ones(E) if w is None else np.ascontiguousarray(w, dtype=np.float64)
    out = ctypes.POINTER(ctypes.c_double)()
    rows = ctypes.c_int64()
    stats = np.zeros(8, dtype=np.int64)
    p = np.ascontiguousarray(perm, dtype=np.int64)
    lib.mirror_flow_chol.restype = ctypes.c_int
    rc = lib.mirror_flow_chol(
        ctypes.c_void_p(row.ctypes.data), ctypes.c_void_p(col.ctypes.data), ctypes.c_void_p(w.ctypes.data),
        ctypes.c_int64(E), ctypes.c_int64(n), ctypes.c_int64(t), oracle.O_N[o_n], ctypes.c_void_p(p.ctypes.data),
        ctypes.c_uint64(seed), ctypes.c_int32(6 * E + 64 * n + 64), ctypes.c_int32(nwaves), ctypes.c_uint64(sched_seed),
        ctypes.byref(out), ctypes.byref(rows), ctypes.c_void_p(stats.ctypes.data))
    assert rc == 0, rc
    m = rows.value
    res = np.ctypeslib.as_array(out, shape=(max(m, 1) * 3,))[: 3 * m].copy().reshape(m, 3)
    lib.mirror_free(out)
    return res, stats


GRAPHS = [("K6", clique(6), 6), ("K40", clique(40), 40), ("P9", path(9), 9), ("star7", star(7), 7), ("star300", star(300), 300),
          ("grid5x6", grid2d(5, 6), 30), ("BA100_50", ba_graph(100, 50, 0), 100), ("BA500_3", ba_graph(500, 3, 1), 500),
          ("BA1500_8", ba_graph(1500, 8, 3), 1500)]


@pytest.mark.parametrize("o_n", ["asc", "desc", "random"])
def test_flow_protocol_matches_oracle(host_mirror, o_n):
    dup = unsorted = 0
    for name, ei, n in GRAPHS:
        perm = np.random.RandomState(7).permutation(n)
        for t in sorted({0, 1, n // 2, n - 1, n + 5}):
            for wts in (None, sym_weights(ei, n, 5)):
                a = oracle.approximate_cholesky(ei, wts, n, t, "random", o_n, perm=perm, shuffle_seed=3)
                for nwaves, ss in ((1, 1), (3, 2), (16, 3), (64, 4)):
                    b, st = mirror_flow(host_mirror, ei, wts, n, t, o_n, perm, seed=3, nwaves=nwaves, sched_seed=ss)
                    assert a.shape == b.shape and np.array_equal(a, b), (name, t, nwaves)
                    dup += int(st[0]); unsorted += int(st[1])
    assert dup > 0 and unsorted > 0, "neither equal ids in a gather nor out-of-order appends were exercised"


def test_flow_protocol_dense_and_hubs(host_mirror):
    """Dense graphs (multi-edges at once, columns beyond the inline chunk directory) and a hub-and-spoke graph whose hub column
    takes thousands of appended entries from concurrent eliminations."""
    rng = np.random.RandomState(5)
    cases = [(400, 40, 200), (150, 70, 75)]
    for n, m, t in cases:
        ei = ba_graph(n, m, 5)
        perm = rng.permutation(n)
        for wts in (None, sym_weights(ei, n, 5)):
            for o_n in ("asc", "desc"):
                a = oracle.approximate_cholesky(ei, wts, n, t, "random", o_n, perm=perm, shuffle_seed=3)
                b, st = mirror_flow(host_mirror, ei, wts, n, t, o_n, perm, seed=3, nwaves=32, sched_seed=n)
                assert a.shape == b.shape and np.array_equal(a, b), (n, m, o_n)
    # two hubs joined to everything, spokes joined in a ring: the hubs are eliminated last
    n = 3000
    a0 = np.concatenate([np.zeros(n - 2, dtype=np.int64), np.ones(n - 2, dtype=np.int64), np.arange(2, n - 1)])
    b0 = np.concatenate([np.arange(2, n), np.arange(2, n), np.arange(3, n)])
    from util import symmetrize
    ei = symmetrize(a0, b0, n)
    perm = np.concatenate([[0, 1], 2 + rng.permutation(n - 2)])   # popped from the back: hubs last
    a = oracle.approximate_cholesky(ei, None, n, n - 3, "random", "asc", perm=perm, shuffle_seed=3)
    b, st = mirror_flow(host_mirror, ei, None, n, n - 3, "asc", perm, seed=3, nwaves=48, sched_seed=9)
    assert a.shape == b.shape and np.array_equal(a, b)
    assert st[4] > 0, "no column went beyond the inline chunk directory"
