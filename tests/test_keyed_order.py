"""Statistical evidence for the injected neighbour order that stands in for the reference's
std::shuffle(std::random_device) (preconditioner.cc:303-307,340-342): over seeds (and over vertices) every
neighbour must be equally likely at every position, and all orders of a short list equally likely.
Chi-square tests at p = 1e-4 on fixed seeds (deterministic: no flakiness)."""
import itertools

import numpy as np
from scipy import stats

import oracle


def _counts(seeds, vertex, phase, nbrs):
    n = len(nbrs)
    c = np.zeros((n, n), dtype=np.int64)          # c[position, neighbour]
    for s in seeds:
        order, keys = oracle.keyed_order(s, vertex, phase, nbrs)
        assert len(set(keys.tolist())) == n        # 52-bit keys: ties would make std::sort's tie rule matter
        c[np.arange(n), order] += 1
    return c


def _chi2_uniform(table):
    exp = table.sum() / table.size
    return float(((table - exp) ** 2 / exp).sum())


def test_position_by_neighbour_uniform_over_seeds():
    nbrs = [3, 17, 18, 250, 251, 4095, 70000, 999999]        # neighbouring ids, sparse ids, large ids
    for vertex, phase in ((0, 0), (12345, 0), (12345, 1)):
        table = _counts(range(20000), vertex, phase, nbrs)
        chi2 = _chi2_uniform(table)
        df = (len(nbrs) - 1) ** 2                            # rows and columns are fixed: a doubly stochastic table
        assert chi2 < stats.chi2.ppf(1 - 1e-4, df), (vertex, phase, chi2)


def test_position_by_neighbour_uniform_over_vertices():
    # one seed, many eliminated vertices with the same neighbour set (the hash must decorrelate vertices too)
    nbrs = list(range(100, 112))
    n = len(nbrs)
    c = np.zeros((n, n), dtype=np.int64)
    for v in range(15000):
        order, _ = oracle.keyed_order(2024, v, 0, nbrs)
        c[np.arange(n), order] += 1
    assert _chi2_uniform(c) < stats.chi2.ppf(1 - 1e-4, (n - 1) ** 2)


def test_all_orders_of_four_equally_likely():
    nbrs = [5, 6, 7, 1000]
    perms = {p: 0 for p in itertools.permutations(range(4))}
    for s in range(24000):
        order, _ = oracle.keyed_order(s * 7919 + 13, 42, 0, nbrs)
        perms[tuple(int(x) for x in order)] += 1
    obs = np.array(list(perms.values()), dtype=np.float64)
    chi2 = float(((obs - obs.mean()) ** 2 / obs.mean()).sum())
    assert chi2 < stats.chi2.ppf(1 - 1e-4, 23), chi2


def test_elimination_and_output_orders_are_independent():
    # phase 0 (elimination) and phase 1 (output) of the same vertex must not be the same order
    nbrs = list(range(20))
    same = sum(np.array_equal(oracle.keyed_order(s, 9, 0, nbrs)[0], oracle.keyed_order(s, 9, 1, nbrs)[0]) for s in range(300))
    assert same == 0


def test_device_definition_matches(host_mirror):
    """rlap_core.h (what the kernels compile) and the oracle define the same keys."""
    import ctypes
    if not hasattr(host_mirror, "mirror_keyed_key"):
        import pytest
        pytest.skip("host mirror without the hook")
    host_mirror.mirror_keyed_key.restype = ctypes.c_double
    host_mirror.mirror_keyed_key.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64]
    for seed, v, ph, nb in ((0, 0, 0, 0), (5, 17, 1, 99), (2**63 + 11, 123456, 0, 7), (77, 3, 1, 2**20)):
        _, k = oracle.keyed_order(seed, v, ph, [nb])
        assert host_mirror.mirror_keyed_key(seed, v, ph, nb) == k[0]
