"""GPU parity of the multi-CU ("dataflow") elimination for o_v = "random" (rlap_amd/csrc/rlap_flow.hip, RLAP_FLOW=1) against the CPU
oracle: row order, indices and weights bit for bit; its long-column sorts against libstdc++'s std::sort itself; schedule jitter and
poisoned memory; the same rows as the round kernel."""
import os

import numpy as np
import pytest
import torch

import oracle
from util import ba_graph, clique, grid2d, introsort_killer, path, star, sym_weights, symmetrize

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from rlap_amd import ops as _ops
    return _ops


@pytest.fixture()
def flow_env():
    old = {k: os.environ.get(k) for k in ("RLAP_FLOW", "RLAP_FLOW_SHAPE", "RLAP_FLOW_WAVES")}
    os.environ["RLAP_FLOW"] = "1"
    yield os.environ
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def call(ops, ei, w, n, t, o_n, perm, seed=3):
    out = ops.approximate_cholesky(torch.from_numpy(np.ascontiguousarray(ei)).cuda(), None if w is None else torch.from_numpy(w).cuda(), n, t,
                                   "random", o_n, perm=torch.from_numpy(np.asarray(perm, dtype=np.int64)), seed=seed)
    return out.numpy()


def check(ops, ei, w, n, t, o_n, perm, what):
    a = oracle.approximate_cholesky(ei, w, n, t, "random", o_n, perm=perm, shuffle_seed=3)
    b = call(ops, ei, w, n, t, o_n, perm)
    assert a.shape == b.shape, f"{what}: rows {b.shape} vs {a.shape}"
    assert np.array_equal(a, b), f"{what}: {int((a != b).any(axis=1).sum())} rows differ"


def test_flow_long_column_sorts_match_libstdcxx(ops):
    """The sort forms a long column goes through (records in LDS: level-synchronous up to 512 keys per segment, partitions on top of it
    beyond, on two waves; 16-bit indices in LDS; records in global memory, partitioned there and sorted through in LDS segment by
    segment) against std::sort itself: ties, runs, killers."""
    rng = np.random.RandomState(1)
    arrays = []
    for trial in range(160):
        n = int(rng.choice([1, 2, 16, 17, 64, 65, 200, 897, 1000, 1024, 1025, 1100, 1500, 2047, 2300, 3000, 3097, 3500, 3600, 5000, 7000, 9000, 12000, 30000]))
        kind = trial % 6
        if kind == 0:
            k = np.ones(n)
        elif kind == 1:
            k = rng.randint(0, 3, size=n).astype(float)
        elif kind == 2:
            k = rng.rand(n)
        elif kind == 3:
            k = np.sort(rng.randint(0, n // 4 + 1, size=n)).astype(float)
        elif kind == 4:
            k = np.sort(rng.randint(0, n // 4 + 1, size=n))[::-1].astype(float)
        else:
            k = np.concatenate([np.ones(n // 2), rng.rand(n - n // 2)])[rng.permutation(n)]
        arrays.append(k)
    for n in (200, 900, 1024, 2000, 3000, 5000, 9000):
        k, hit = introsort_killer(n)
        assert hit
        arrays += [k, -k, np.concatenate([k, k[: n // 3]])]
    offs = np.zeros(len(arrays) + 1, dtype=np.int32)
    offs[1:] = np.cumsum([len(a) for a in arrays])
    keys = torch.from_numpy(np.concatenate(arrays)).cuda()
    offs_t = torch.from_numpy(offs).cuda()
    lib, h = ops._handle(torch.device("cuda", 0))
    for desc in (32, 33, 96, 97, 160, 161):   # bit 5: the dataflow kernel's sort; bit 0: descending; bit 6: records in global memory; bit 7: index sort in LDS
        out = torch.full((int(offs[-1]),), -7, dtype=torch.int32, device="cuda")
        assert lib.rlap_debug_wave_sort(h, keys.data_ptr(), offs_t.data_ptr(), len(arrays), desc, out.data_ptr()) == 0
        got = out.cpu().numpy()
        for a_i, k in enumerate(arrays):
            exp = oracle.stdsort_perm(k, bool(desc & 1))
            assert np.array_equal(got[offs[a_i]:offs[a_i + 1]], exp), (a_i, len(k), desc)


def test_flow_radix_form_for_distinct_keys(ops):
    """Where keys cannot repeat the elimination orders them with a radix sort (ids of a column without multi-edges, tags): the hook's
    bit 9 runs that form in every place the records can live (short-column LDS view is exercised by the parity tests; here records in
    LDS, 16-bit indices in LDS, records in global memory with the radix buffers in LDS, and beyond that the fall-back to the std::sort
    form) -- distinct keys have one sorted order."""
    rng = np.random.RandomState(5)
    arrays = []
    for n in (1, 2, 64, 65, 100, 512, 897, 1500, 3000, 3096, 3097, 4000, 4184, 4185, 5000, 6100, 6300, 9000, 20000):
        for hi in (n, 1 << 20, (1 << 31) - 1):
            if hi <= 4 * n:
                k = rng.permutation(max(hi, n))[:n]
            else:   # (distinct draws without materialising the population)
                k = rng.permutation(np.unique(rng.randint(0, hi, size=3 * n + 8)))[:n]
            assert len(k) == n and len(np.unique(k)) == n
            arrays.append(k.astype(np.float64))
    offs = np.zeros(len(arrays) + 1, dtype=np.int32)
    offs[1:] = np.cumsum([len(a) for a in arrays])
    keys = torch.from_numpy(np.concatenate(arrays)).cuda()
    offs_t = torch.from_numpy(offs).cuda()
    lib, h = ops._handle(torch.device("cuda", 0))
    for desc in (32 | 512, 33 | 512, 96 | 512, 97 | 512, 160 | 512, 161 | 512):
        out = torch.full((int(offs[-1]),), -7, dtype=torch.int32, device="cuda")
        assert lib.rlap_debug_wave_sort(h, keys.data_ptr(), offs_t.data_ptr(), len(arrays), desc, out.data_ptr()) == 0
        got = out.cpu().numpy()
        for a_i, k in enumerate(arrays):
            exp = np.argsort(k, kind="stable")
            if desc & 1:
                exp = exp[::-1]
            assert np.array_equal(got[offs[a_i]:offs[a_i + 1]], exp), (a_i, len(k), desc)


GRAPHS = [("K4", clique(4), 4), ("K6", clique(6), 6), ("P9", path(9), 9), ("star7", star(7), 7), ("K40", clique(40), 40),
          ("grid5x6", grid2d(5, 6), 30), ("BA100_50", ba_graph(100, 50, 0), 100), ("BA500_3", ba_graph(500, 3, 1), 500),
          ("BA3000_10", ba_graph(3000, 10, 2), 3000), ("BA400_40", ba_graph(400, 40, 5), 400)]


@pytest.mark.parametrize("shape", ["1", "2", "3"])
@pytest.mark.parametrize("o_n", ["asc", "desc", "random"])
def test_flow_matches_oracle(ops, flow_env, o_n, shape):
    """All graph families x weights x num_remove, both workgroup shapes (large LDS block / small blocks)."""
    flow_env["RLAP_FLOW_SHAPE"] = shape
    rng = np.random.RandomState(7)
    for nm, ei, n in GRAPHS:
        perm = rng.permutation(n)
        for wts in (None, sym_weights(ei, n, 5)):
            for t in sorted({0, 1, n // 2, n - 1, n + 5}):
                check(ops, ei, wts, n, t, o_n, perm, (nm, o_n, t, wts is not None))


def test_flow_long_columns_and_hubs(ops, flow_env):
    """Columns beyond the wave's LDS block (dense graph), beyond the inline chunk directory (hubs taking thousands of appended
    entries from concurrent eliminations), a star whose centre goes first."""
    rng = np.random.RandomState(5)
    ei = ba_graph(1200, 200, 4)
    perm = rng.permutation(1200)
    for t in (600, 1199):
        check(ops, ei, None, 1200, t, "asc", perm, ("BA1200_200", t))
    check(ops, ei, sym_weights(ei, 1200, 5), 1200, 900, "desc", perm, "BA1200_200 w")
    n = 3000
    a0 = np.concatenate([np.zeros(n - 2, dtype=np.int64), np.ones(n - 2, dtype=np.int64), np.arange(2, n - 1)])
    b0 = np.concatenate([np.arange(2, n), np.arange(2, n), np.arange(3, n)])
    ei = symmetrize(a0, b0, n)
    for hubs_last in (True, False):
        perm = np.concatenate([[0, 1], 2 + rng.permutation(n - 2)]) if hubs_last else np.concatenate([2 + rng.permutation(n - 2), [1, 0]])
        for t in (n - 3, n - 1, n // 2):
            check(ops, ei, None, n, t, "asc", perm, ("hubs", hubs_last, t))
    ei = star(5000)
    perm = np.concatenate([1 + rng.permutation(4999), [0]])   # the centre is popped first: one 4999-entry column
    check(ops, ei, None, 5000, 2500, "asc", perm, "star5000")
    check(ops, ei, sym_weights(ei, 5000, 2), 5000, 2500, "desc", perm, "star5000 w")
    # beyond the 16-bit stop lists of the wave sorts (65,000 entries): one lane sorts, the others must not call it a stall
    n = 70001
    ei = star(n)
    perm = np.concatenate([1 + rng.permutation(n - 1), [0]])
    check(ops, ei, None, n, 300, "asc", perm, "star70001")


def test_flow_equals_round_kernel_on_a_medium_graph(ops, flow_env):
    from rlap_amd import graphs
    n = 120000
    ei = graphs.barabasi_albert(n, 10, 5).cuda()
    perm = torch.from_numpy(np.random.RandomState(3).permutation(n))
    flow_env["RLAP_FLOW"] = "0"
    a = ops.approximate_cholesky(ei, None, n, n // 2, "random", "asc", perm=perm, return_device="same")
    assert ops.last_stats["n_rounds"] > 0
    flow_env["RLAP_FLOW"] = "1"
    b = ops.approximate_cholesky(ei, None, n, n // 2, "random", "asc", perm=perm, return_device="same")
    assert ops.last_stats["n_rounds"] == 0, "the dataflow kernel did not run"
    assert a.shape == b.shape and torch.equal(a, b)
    for waves in ("7", "300"):   # any number of waves in flight gives the same rows
        flow_env["RLAP_FLOW_WAVES"] = waves
        c = ops.approximate_cholesky(ei, None, n, n // 2, "random", "asc", perm=perm, return_device="same")
        assert torch.equal(a, c), waves
    flow_env.pop("RLAP_FLOW_WAVES", None)
    # one wave: the sequential order itself (a smaller graph: one wave is slow)
    n2 = 4000
    ei2 = ba_graph(n2, 6, 9)
    perm2 = np.random.RandomState(4).permutation(n2)
    flow_env["RLAP_FLOW_WAVES"] = "1"
    check(ops, ei2, None, n2, n2 // 2, "asc", perm2, "one wave")


def test_flow_batched_equals_single_calls(ops, flow_env):
    """A batch: every graph has its own uniform stream (look-back stops at the graph's sentinel) and its own seed."""
    from rlap_amd import graphs
    rng = np.random.RandomState(2)
    sizes = [300, 1, 0, 777, 64, 2048, 5]
    eis = [graphs.barabasi_albert(s, 4, 100 + i) if s > 8 else torch.from_numpy(clique(s) if s > 1 else np.zeros((2, 0), dtype=np.int64)) for i, s in enumerate(sizes)]
    big, node_ptr = graphs.batch_disjoint(eis, sizes)
    perm = np.concatenate([rng.permutation(s) for s in sizes]).astype(np.int64)
    ts = [s // 2 for s in sizes]
    for o_n in ("asc", "random"):
        sc, rp = ops.approximate_cholesky_batched(big.cuda(), None, node_ptr, ts, "random", o_n, perm=torch.from_numpy(perm), seed=5)
        sc = sc.cpu().numpy()
        off = 0
        for g, s in enumerate(sizes):
            a = oracle.approximate_cholesky(eis[g].numpy(), None, s, ts[g], "random", o_n, perm=perm[off:off + s], shuffle_seed=5 + g)
            b = sc[rp[g]:rp[g + 1]].copy()
            b[:, :2] -= off
            assert a.shape == b.shape and np.array_equal(a, b), (g, s, o_n)
            off += s


def test_flow_under_jitter_and_poison(ops, flow_env):
    """Schedule perturbation (waves sleep at the phase boundaries and between their claims and stores) and poisoned arena / LDS:
    the rows do not move."""
    n = 6000
    ei = ba_graph(n, 8, 11)
    perm = np.random.RandomState(6).permutation(n)
    a = oracle.approximate_cholesky(ei, None, n, n // 2, "random", "asc", perm=perm, shuffle_seed=3)
    try:
        for jit, poison in ((3, -1), (9, 170), (0, 255), (5, 0)):
            ops.debug_set_jitter(jit)
            ops.debug_set_poison(poison)
            for shape in ("1", "2", "3"):
                flow_env["RLAP_FLOW_SHAPE"] = shape
                b = call(ops, ei, None, n, n // 2, "asc", perm)
                assert a.shape == b.shape and np.array_equal(a, b), (jit, poison, shape)
    finally:
        ops.debug_set_jitter(0)
        ops.debug_set_poison(-1)


def test_flow_growth_retries(ops, flow_env):
    """Tiny pool / uniform table / long-column scratch: the call repeats itself with more and returns the same rows."""
    n = 3000
    ei = ba_graph(n, 10, 2)
    perm = np.random.RandomState(8).permutation(n)
    a = oracle.approximate_cholesky(ei, None, n, n - 1, "random", "asc", perm=perm, shuffle_seed=3)
    ops.debug_set_limits(pool_factor=0.05, log_factor=0.05, rng_len=1000, scratch_entries=16)
    b = call(ops, ei, None, n, n - 1, "asc", perm)
    assert ops.last_stats["n_retries"] >= 1
    assert a.shape == b.shape and np.array_equal(a, b)


def test_abandoned_launch_with_appends_in_flight(ops, flow_env):
    """A launch given up while hub columns are being appended to (pool and scratch limits of a fresh handle's size on a batch of hub
    graphs, all vertices eliminated, waves delayed): append counts are then ahead of their chunks (directory words EMPTY / BUSY / FAIL),
    and the kernels that follow must not build or walk chains from them.  (Soak case 41/33 of round 4: `k_flow_finish` followed such a
    directory word into a memory fault.)  The call repeats itself and returns the oracle's rows."""
    from rlap_amd import graphs
    rng = np.random.RandomState(305289331 % (1 << 31))
    n, G = 9000, 6
    eis = []
    for g in range(G):
        ei = ba_graph(n, 6, 700 + g)
        h = int(rng.randint(n))
        others = np.flatnonzero(rng.rand(n) < 0.33)
        others = others[others != h].astype(np.int64)
        eis.append(symmetrize(np.concatenate([ei[0], np.full(others.size, h, dtype=np.int64)]), np.concatenate([ei[1], others]), n))
    big, node_ptr = graphs.batch_disjoint([torch.from_numpy(e) for e in eis], [n] * G)
    perm = np.concatenate([rng.permutation(n) for _ in range(G)]).astype(np.int64)
    try:
        ops.debug_set_jitter(9)
        for limits in (dict(pool_factor=0.3), dict(pool_factor=0.6, scratch_entries=20000), dict()):
            flow_env["RLAP_FLOW_SHAPE"] = "1"
            if limits:
                ops.debug_set_limits(**limits)
            sc, rp = ops.approximate_cholesky_batched(big.cuda(), None, node_ptr, [n] * G, "random", "random", perm=torch.from_numpy(perm), seed=21)
            sc = sc.cpu().numpy()
            if limits:
                assert ops.last_stats["n_retries"] >= 1, limits
            for g in range(G):
                a = oracle.approximate_cholesky(eis[g], None, n, n, "random", "random", perm=perm[g * n:(g + 1) * n], shuffle_seed=21 + g)
                b = sc[rp[g]:rp[g + 1]].copy()
                b[:, :2] -= g * n
                assert a.shape == b.shape and np.array_equal(a, b), (limits, g)
    finally:
        ops.debug_set_jitter(0)


def test_a_flow_launch_that_gives_up_is_repeated_on_the_round_kernel(ops, flow_env):
    """The stall watchdog set to 1 ms on a star of 16,001 vertices (its centre's column of 13,000 entries keeps one wave busy for
    milliseconds while every later leaf waits): the dataflow launch ends with ST_INTERNAL, the call runs once more on the round kernel,
    the rows are the oracle's."""
    n = 16001
    ei = star(n)
    perm = np.random.RandomState(4).permutation(n)
    perm = np.concatenate([perm[perm != 0][:3000], [0], perm[perm != 0][3000:]])   # the centre is eliminated with 13,000 leaves still there
    a = oracle.approximate_cholesky(ei, None, n, n - 1, "random", "asc", perm=perm, shuffle_seed=3)
    old = os.environ.get("RLAP_FLOW_STALL_MS")
    os.environ["RLAP_FLOW_STALL_MS"] = "1"
    try:
        b = call(ops, ei, None, n, n - 1, "asc", perm)
        assert ops.last_stats["n_retries"] >= 1 and ops.last_stats["n_rounds"] > 0   # (rounds: the round kernel produced the result)
    finally:
        if old is None:
            os.environ.pop("RLAP_FLOW_STALL_MS", None)
        else:
            os.environ["RLAP_FLOW_STALL_MS"] = old
    assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("o_v", ["random", "degree", "coarsen"])
def test_frontier_mode_matches_the_oracle_in_that_mode(ops, flow_env, o_v):
    """mode="frontier" (SURVEY section 7 step 7 / 8(b)): counter-based uniforms keyed by (seed, vertex, position) -- bit-exact
    against the oracle run in the same mode, for every order (the round kernels draw the same numbers), different from mode="exact",
    and mode="exact" is what a later call without the keyword gets."""
    rng = np.random.RandomState(3)
    for nm, ei, n in GRAPHS + [("BA6000_8", ba_graph(6000, 8, 11), 6000), ("BA1200_200", ba_graph(1200, 200, 4), 1200)]:
        perm = rng.permutation(n)
        for o_n in ("asc", "random"):
            for wts in (None, sym_weights(ei, n, 5)):
                t = n // 2
                a = oracle.approximate_cholesky(ei, wts, n, t, o_v, o_n, perm=perm if o_v == "random" else None, shuffle_seed=9, mode="frontier")
                b = ops.approximate_cholesky(torch.from_numpy(np.ascontiguousarray(ei)).cuda(), None if wts is None else torch.from_numpy(wts).cuda(), n, t, o_v, o_n,
                                             perm=torch.from_numpy(perm) if o_v == "random" else None, seed=9, mode="frontier").numpy()
                assert a.shape == b.shape and np.array_equal(a, b), (nm, o_v, o_n, wts is not None)
    n = 3000
    ei = ba_graph(n, 10, 2)
    perm = rng.permutation(n)
    ex = oracle.approximate_cholesky(ei, None, n, n // 2, o_v, "asc", perm=perm if o_v == "random" else None, shuffle_seed=9)
    fr = oracle.approximate_cholesky(ei, None, n, n // 2, o_v, "asc", perm=perm if o_v == "random" else None, shuffle_seed=9, mode="frontier")
    assert ex.shape != fr.shape or not np.array_equal(ex, fr)
    b = ops.approximate_cholesky(torch.from_numpy(ei).cuda(), None, n, n // 2, o_v, "asc", perm=torch.from_numpy(perm) if o_v == "random" else None, seed=9).numpy()
    assert ex.shape == b.shape and np.array_equal(ex, b), "the default mode is exact again"


def test_frontier_mode_batched_and_hubs(ops, flow_env):
    from rlap_amd import graphs
    rng = np.random.RandomState(2)
    sizes = [300, 777, 64, 2048]
    eis = [graphs.barabasi_albert(s, 4, 100 + i) for i, s in enumerate(sizes)]
    big, node_ptr = graphs.batch_disjoint(eis, sizes)
    perm = np.concatenate([rng.permutation(s) for s in sizes]).astype(np.int64)
    ts = [s // 2 for s in sizes]
    for flow in ("0", "1"):   # the round kernel and the dataflow kernel draw the same numbers
        flow_env["RLAP_FLOW"] = flow
        sc, rp = ops.approximate_cholesky_batched(big.cuda(), None, node_ptr, ts, "random", "asc", perm=torch.from_numpy(perm), seed=5, mode="frontier")
        sc = sc.cpu().numpy()
        off = 0
        for g, s in enumerate(sizes):
            a = oracle.approximate_cholesky(eis[g].numpy(), None, s, ts[g], "random", "asc", perm=perm[off:off + s], shuffle_seed=5 + g, mode="frontier")
            b = sc[rp[g]:rp[g + 1]].copy()
            b[:, :2] -= off
            assert a.shape == b.shape and np.array_equal(a, b), (flow, g)
            off += s
    flow_env["RLAP_FLOW"] = "1"
    n = 3000   # hubs that take thousands of appended entries: entries one elimination pushes into a column share a tag
    a0 = np.concatenate([np.zeros(n - 2, dtype=np.int64), np.ones(n - 2, dtype=np.int64), np.arange(2, n - 1)])
    b0 = np.concatenate([np.arange(2, n), np.arange(2, n), np.arange(3, n)])
    ei = symmetrize(a0, b0, n)
    perm = np.concatenate([[0, 1], 2 + rng.permutation(n - 2)])
    for t in (n - 3, n - 1):
        a = oracle.approximate_cholesky(ei, None, n, t, "random", "asc", perm=perm, shuffle_seed=3, mode="frontier")
        b = ops.approximate_cholesky(torch.from_numpy(ei).cuda(), None, n, t, "random", "asc", perm=torch.from_numpy(perm), seed=3, mode="frontier").numpy()
        assert a.shape == b.shape and np.array_equal(a, b), t
