"""CPU checks of the code the HIP kernels share (rlap_amd/csrc/rlap_core.h):
  * std::sort emulation == libstdc++ std::sort, permutation for permutation;
  * chunked columns + lazy bucket-stack PQ + sequential elimination == oracle, bit-exact."""
import ctypes

import numpy as np
import pytest

import oracle
from util import ba_graph, clique, grid2d, path, star, sym_weights


def _mperm(lib, fn, keys, desc):
    k = np.ascontiguousarray(keys, dtype=np.float64)
    p = np.empty(len(k), dtype=np.int64)
    getattr(lib, fn)(ctypes.c_void_p(k.ctypes.data), ctypes.c_int64(len(k)), ctypes.c_int(desc), ctypes.c_void_p(p.ctypes.data))
    return p


def test_std_sort_emulation_matches_libstdcxx(host_mirror):
    rng = np.random.RandomState(0)
    for trial in range(600):
        n = int(rng.choice([1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 40, 64, 65, 100, 257, 1000, 3000]))
        kind = trial % 5
        if kind == 0:
            k = np.ones(n)
        elif kind == 1:
            k = rng.randint(0, 3, size=n).astype(float)
        elif kind == 2:
            k = rng.rand(n)
        elif kind == 3:
            k = np.sort(rng.randint(0, n // 4 + 1, size=n)).astype(float)
        else:
            k = np.sort(rng.randint(0, n // 4 + 1, size=n))[::-1].astype(float)
        for desc in (0, 1):
            assert np.array_equal(oracle.stdsort_perm(k, bool(desc)), _mperm(host_mirror, "mirror_sort_perm", k, desc))


def test_depth_limit_branch_matches_libstdcxx(host_mirror):
    """Adversarial keys (util.introsort_killer) push std::sort into its heap-sort branch: the restatement follows."""
    from util import introsort_killer
    for n in (40, 64, 100, 200, 384, 512, 3000):
        k, hit = introsort_killer(n)
        assert hit
        for keys in (k, -k, np.concatenate([k, k[: n // 3]])):   # distinct, mirrored for the descending comparator, with ties
            for desc in (0, 1):
                assert np.array_equal(oracle.stdsort_perm(keys, bool(desc)), _mperm(host_mirror, "mirror_sort_perm", keys, desc)), (n, desc)


def test_small_stack_free_sort_matches_libstdcxx(host_mirror):
    rng = np.random.RandomState(2)
    for trial in range(3000):
        n = int(rng.randint(1, 33))
        kind = trial % 4
        k = [np.ones(n), rng.randint(0, 3, size=n).astype(float), rng.rand(n), np.sort(rng.randint(0, 5, size=n))[::-1].astype(float)][kind]
        for desc in (0, 1):
            assert np.array_equal(oracle.stdsort_perm(k, bool(desc)), _mperm(host_mirror, "mirror_sort_perm_small", k, desc))


def test_heap_sort_fallback_matches_partial_sort(host_mirror):
    rng = np.random.RandomState(1)
    for t in range(400):
        n = int(rng.choice([2, 3, 4, 5, 17, 18, 33, 100, 1001]))
        k = rng.randint(0, max(2, n // 3), size=n).astype(float) if t % 2 else rng.rand(n)
        for d in (0, 1):
            assert np.array_equal(oracle.heapsort_perm(k, bool(d)), _mperm(host_mirror, "mirror_heapsort_perm", k, d))


def _mirror(lib, ei, w, n, t, o_v, o_n, perm=None, seed=0):
    E = ei.shape[1]
    row = np.ascontiguousarray(ei[0])
    col = np.ascontiguousarray(ei[1])
    w = np.ones(E) if w is None else np.ascontiguousarray(w, dtype=np.float64)
    out = ctypes.POINTER(ctypes.c_double)()
    rows = ctypes.c_int64()
    order = np.full(max(n, 1), -1, dtype=np.int64)
    p = np.ascontiguousarray(perm, dtype=np.int64) if perm is not None else None
    rc = lib.mirror_approx_chol(
        ctypes.c_void_p(row.ctypes.data), ctypes.c_void_p(col.ctypes.data), ctypes.c_void_p(w.ctypes.data),
        ctypes.c_int64(E), ctypes.c_int64(n), ctypes.c_int64(t), oracle.O_V[o_v], oracle.O_N[o_n],
        ctypes.c_void_p(p.ctypes.data) if p is not None else None, ctypes.c_uint64(seed), ctypes.c_int32(4 * E + 64),
        ctypes.byref(out), ctypes.byref(rows), ctypes.c_void_p(order.ctypes.data))
    assert rc == 0
    m = rows.value
    res = np.ctypeslib.as_array(out, shape=(max(m, 1) * 3,))[: 3 * m].copy().reshape(m, 3)
    lib.mirror_free(out)
    return res, order[:n]


GRAPHS = [("K6", clique(6), 6), ("P9", path(9), 9), ("star7", star(7), 7), ("grid5x6", grid2d(5, 6), 30),
          ("BA100_50", ba_graph(100, 50, 0), 100), ("BA500_3", ba_graph(500, 3, 1), 500)]


@pytest.mark.parametrize("o_v", ["degree", "random", "coarsen"])
@pytest.mark.parametrize("o_n", ["asc", "desc", "random"])
def test_device_data_structures_match_oracle(host_mirror, o_v, o_n):
    for name, ei, n in GRAPHS:
        perm = np.random.RandomState(7).permutation(n) if o_v == "random" else None
        for t in sorted({0, 1, n // 2, n - 1, n + 5}):
            a, oa = oracle.approximate_cholesky(ei, None, n, t, o_v, o_n, perm=perm, shuffle_seed=3, return_order=True)
            b, ob = _mirror(host_mirror, ei, None, n, t, o_v, o_n, perm=perm, seed=3)
            assert np.array_equal(oa, ob), (name, t)
            assert a.shape == b.shape and np.array_equal(a, b), (name, t)
        w = sym_weights(ei, n, 5)
        a = oracle.approximate_cholesky(ei, w, n, n // 2, o_v, o_n, perm=perm, shuffle_seed=4)
        b, _ = _mirror(host_mirror, ei, w, n, n // 2, o_v, o_n, perm=perm, seed=4)
        assert np.array_equal(a, b), name


def _mirror_batch(lib, ei, w, n, t, o_v, o_n, B, perm=None, seed=0, bc=32):
    E = ei.shape[1]
    row = np.ascontiguousarray(ei[0])
    col = np.ascontiguousarray(ei[1])
    w = np.ones(E) if w is None else np.ascontiguousarray(w, dtype=np.float64)
    out = ctypes.POINTER(ctypes.c_double)()
    rows = ctypes.c_int64()
    order = np.full(max(n, 1), -1, dtype=np.int64)
    stats = np.zeros(24, dtype=np.int64)
    p = np.ascontiguousarray(perm, dtype=np.int64) if perm is not None else None
    lib.mirror_approx_chol_batch_bc.restype = ctypes.c_int
    rc = lib.mirror_approx_chol_batch_bc(
        ctypes.c_void_p(row.ctypes.data), ctypes.c_void_p(col.ctypes.data), ctypes.c_void_p(w.ctypes.data),
        ctypes.c_int64(E), ctypes.c_int64(n), ctypes.c_int64(t), oracle.O_V[o_v], oracle.O_N[o_n],
        ctypes.c_void_p(p.ctypes.data) if p is not None else None, ctypes.c_uint64(seed), ctypes.c_int32(4 * E + 64),
        ctypes.c_int32(B), ctypes.c_int32(bc), ctypes.byref(out), ctypes.byref(rows), ctypes.c_void_p(order.ctypes.data),
        ctypes.c_void_p(stats.ctypes.data))
    assert rc == 0
    m = rows.value
    res = np.ctypeslib.as_array(out, shape=(max(m, 1) * 3,))[: 3 * m].copy().reshape(m, 3)
    lib.mirror_free(out)
    return res, order[:n], stats


@pytest.mark.parametrize("o_v", ["degree", "random", "coarsen"])
@pytest.mark.parametrize("o_n", ["asc", "desc", "random"])
def test_batch_rounds_equal_sequential_order(host_mirror, o_v, o_n):
    """The frontier kernel's round rules (predict B pops, commit the longest independent,
    un-pre-empted prefix, RNG offsets by prefix sum, shared targets in candidate order)
    replayed on the CPU: bit-exact against the oracle for any batch size."""
    for name, ei, n in GRAPHS + [("BA1500_8", ba_graph(1500, 8, 3), 1500)]:
        perm = np.random.RandomState(10).permutation(n) if o_v == "random" else None
        for t in sorted({1, n // 2, n - 1}):
            for wts in (None, sym_weights(ei, n, 5)):
                a, oa = oracle.approximate_cholesky(ei, wts, n, t, o_v, o_n, perm=perm, shuffle_seed=3, return_order=True)
                for B, bc in ((1, 32), (7, 32), (128, 32), (64, 64), (32, 128)):   # 64- / 128-slot candidates: the o_v="random" kernel variants
                    b, ob, _ = _mirror_batch(host_mirror, ei, wts, n, t, o_v, o_n, B, perm=perm, seed=3, bc=bc)
                    assert np.array_equal(oa, ob), (name, t, B, bc)
                    assert a.shape == b.shape and np.array_equal(a, b), (name, t, B, bc)


@pytest.mark.parametrize("o_n", ["asc", "desc", "random"])
def test_wide_candidates_equal_sequential_order(host_mirror, o_n):
    """128-slot candidates (o_v="random" on graphs whose columns run long): dense graphs, where most columns have 65..128
    entries and multi-edges appear at once, unit and tie-free weights."""
    for n, m in ((400, 40), (900, 20), (150, 70)):
        ei = ba_graph(n, m, 5)
        perm = np.random.RandomState(11).permutation(n)
        for wts in (None, sym_weights(ei, n, 5)):
            a, oa = oracle.approximate_cholesky(ei, wts, n, n // 2, "random", o_n, perm=perm, shuffle_seed=3, return_order=True)
            b, ob, st = _mirror_batch(host_mirror, ei, wts, n, n // 2, "random", o_n, 32, perm=perm, seed=3, bc=128)
            assert np.array_equal(oa, ob), (n, m)
            assert a.shape == b.shape and np.array_equal(a, b), (n, m)
            b64, _, st64 = _mirror_batch(host_mirror, ei, wts, n, n // 2, "random", o_n, 64, perm=perm, seed=3, bc=64)
            assert st[1] < st64[1], "fewer single-vertex fallbacks than with 64 slots"


@pytest.mark.parametrize("o_v", ["degree", "random"])
@pytest.mark.parametrize("o_n", ["asc", "desc", "random"])
def test_patched_dependent_candidates_equal_sequential_order(host_mirror, o_v, o_n):
    """Round 2 rule (rlap_core.h::cand_patch): a candidate adjacent to an earlier candidate of its round is patched from
    that one's sampled record instead of cutting the round.  Graphs large enough for the rule to fire hundreds of times,
    unit and tie-free weights, both candidate widths; bit-exact against the oracle, and the rule really is exercised."""
    for n, m in ((8000, 10), (3000, 3)):
        ei = ba_graph(n, m, 7)
        perm = np.random.RandomState(10).permutation(n) if o_v == "random" else None
        for wts in (None, sym_weights(ei, n, 5)):
            a, oa = oracle.approximate_cholesky(ei, wts, n, n // 2, o_v, o_n, perm=perm, shuffle_seed=3, return_order=True)
            for B, bc in ((128, 32), (64, 64)):
                b, ob, st = _mirror_batch(host_mirror, ei, wts, n, n // 2, o_v, o_n, B, perm=perm, seed=3, bc=bc)
                assert np.array_equal(oa, ob), (n, m, B, bc)
                assert a.shape == b.shape and np.array_equal(a, b), (n, m, B, bc)
                if o_v == "degree" and bc == 32 and m == 10:
                    assert st[18] > 20, "the patch rule was not exercised"


def test_patch_refused_when_the_new_weight_is_not_positive(host_mirror):
    """Found by the soak run (tests/tools/soak.py, case 3023): late in a long column of a `desc` elimination rounding leaves
    f > 1, the new weight f(1-f)wdeg is negative (-5e-31) and the rewritten entry is dead for getColumnLength
    (preconditioner.cc:252, `val > 0`).  A dependent candidate patched with that entry kept a neighbour the reference no longer
    sees; cand_patch now refuses, the candidate is gathered afresh.  Same graphs, same call as the failing case."""
    from rlap_amd import graphs
    n, t, seed = 31749, 28574, 690086160
    dead = 0
    for g in (5, 8):
        ei = graphs.barabasi_albert(n, 5, seed + g).numpy()
        a, oa = oracle.approximate_cholesky(ei, None, n, t, "degree", "desc", shuffle_seed=seed + g, return_order=True)
        for B in (4, 32, 128):
            b, ob, st = _mirror_batch(host_mirror, ei, None, n, t, "degree", "desc", B, seed=seed + g)
            assert np.array_equal(oa, ob), (g, B)
            assert a.shape == b.shape and np.array_equal(a, b), (g, B)
            dead += int(st[19])
    assert dead > 0, "the case no longer produces a non-positive new weight in front of a dependent candidate"
