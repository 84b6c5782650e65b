"""GPU parity: the HIP path (through the C ABI, rlap_amd.ops) against the CPU oracle
on the same seeded inputs.  Bar: indices and row order bit-exact, weights bit-exact
(the kernels reproduce the reference's operation order; -ffp-contract=off)."""
import json
import os
import struct

import numpy as np
import pytest
import torch

import oracle
from util import ba_graph, canonical, clique, grid2d, path, star, sym_weights

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "survey_appendix_c.json")))
MAKERS = {"path": path, "clique": clique, "star": star}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from rlap_amd import ops as _ops
    return _ops


def gpu_call(ops, ei, w, n, t, o_v, o_n, perm=None, seed=0):
    ei_t = torch.from_numpy(np.ascontiguousarray(ei)).cuda()
    w_t = None if w is None else torch.from_numpy(np.asarray(w, dtype=np.float64)).cuda()
    p_t = None if perm is None else torch.from_numpy(np.asarray(perm, dtype=np.int64))
    out = ops.approximate_cholesky(ei_t, w_t, n, t, o_v, o_n, perm=p_t, seed=seed)
    assert out.dtype == torch.float64 and out.device.type == "cpu"   # reference contract
    return out.numpy()


def _where(a, b):
    """First differing row and the extent of the difference (which column, how many of its rows, same multiset or not)."""
    d = np.flatnonzero((a != b).any(axis=1))
    if d.size == 0:
        return ""
    col = a[d[0], 1]
    rows = np.flatnonzero(a[:, 1] == col)
    same_set = sorted(map(tuple, a[rows].tolist())) == sorted(map(tuple, b[rows].tolist()))
    return (f" [{d.size} rows differ, first {d[0]}: got {a[d[0]].tolist()} expected {b[d[0]].tolist()}; column {int(col)} has {rows.size} rows "
            f"({int((a[rows] != b[rows]).any(axis=1).sum())} differ, same rows in another order: {same_set}); columns touched: "
            f"{np.unique(a[d, 1]).size}]")


def assert_same(a, b, what=""):
    assert a.shape == b.shape, f"{what}: rows {a.shape} vs {b.shape}"
    assert np.array_equal(a[:, :2], b[:, :2]), f"{what}: indices differ" + _where(a, b)
    assert np.array_equal(a[:, 2], b[:, 2]), f"{what}: weights differ, max abs {np.abs(a[:, 2] - b[:, 2]).max()}"


def test_rng_table_matches_mt19937_64(ops):
    u = ops.rng_uniforms(20000).cpu().numpy()
    ref, _ = oracle.uniforms(20000)
    assert np.array_equal(u, ref)
    for i, b in enumerate(GOLD["rng"]["u_bits"]):
        assert struct.unpack("<Q", struct.pack("<d", float(u[i])))[0] == int(b, 16)


def test_wave_parallel_std_sort_matches_libstdcxx(ops):
    """The wave-parallel emulation (parallel Hoare partition + per-segment insertion) must give
    libstdc++'s permutation, ties included."""
    import ctypes
    from rlap_amd import _lib
    rng = np.random.RandomState(0)
    arrays = []
    for trial in range(1500):
        n = int(rng.choice([1, 2, 15, 16, 17, 18, 24, 31, 32, 33, 40, 48, 56, 63, 64, 65, 100, 127, 128, 129, 257, 300, 320, 321, 350, 384, 400, 512]))
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
    offs = np.zeros(len(arrays) + 1, dtype=np.int32)
    offs[1:] = np.cumsum([len(a) for a in arrays])
    keys = torch.from_numpy(np.concatenate(arrays)).cuda()
    offs_t = torch.from_numpy(offs).cuda()
    lib, h = ops._handle(torch.device("cuda", 0))
    # bit 0: descending; bit 1: arrays of <= 64 elements go through the register-resident variant (one wave per array);
    # bit 2: the half-wave variant, two arrays of <= 32 elements side by side (longer ones are skipped);
    # bit 3: the level-synchronous variant (every segment of a recursion level partitioned in the same pass); bit 4: the same over an
    # index array whose comparison looks the keys up (128-slot candidates; arrays beyond 128 keys are skipped)
    for desc in (0, 1, 2, 3, 4, 5, 8, 9, 16, 17):
        out = torch.full((int(offs[-1]),), -7, dtype=torch.int32, device="cuda")
        rc = lib.rlap_debug_wave_sort(h, keys.data_ptr(), offs_t.data_ptr(), len(arrays), desc, out.data_ptr())
        assert rc == 0
        got = out.cpu().numpy()
        for a_i, k in enumerate(arrays):
            if (desc & 4) and len(k) > 32:
                continue
            if (desc & 16) and len(k) > 128:
                continue
            exp = oracle.stdsort_perm(k, bool(desc & 1))
            assert np.array_equal(got[offs[a_i]:offs[a_i + 1]], exp), (a_i, len(k), desc)


def test_sorts_follow_the_depth_limit_branch(ops):
    """Adversarial keys that push libstdc++'s introsort into its heap-sort branch (util.introsort_killer): the LDS
    variant follows it; the register variants report it (the kernels then take the sequential restatement)."""
    from util import introsort_killer
    arrays, hits = [], []
    for n in (33, 40, 64, 100, 200, 384, 512):
        k, hit = introsort_killer(n)
        for keys in (k, -k, np.concatenate([k, k[: n // 3]])[:512]):
            arrays.append(keys); hits.append(hit)
    offs = np.zeros(len(arrays) + 1, dtype=np.int32)
    offs[1:] = np.cumsum([len(a) for a in arrays])
    keys = torch.from_numpy(np.concatenate(arrays)).cuda()
    offs_t = torch.from_numpy(offs).cuda()
    lib, h = ops._handle(torch.device("cuda", 0))
    for desc in (0, 1, 2, 3, 8, 9):   # (8: the level-synchronous variant reports the depth limit, the hook starts over with the LDS variant)
        out = torch.full((int(offs[-1]),), -7, dtype=torch.int32, device="cuda")
        assert lib.rlap_debug_wave_sort(h, keys.data_ptr(), offs_t.data_ptr(), len(arrays), desc, out.data_ptr()) == 0
        got = out.cpu().numpy()
        for a_i, k in enumerate(arrays):
            exp = oracle.stdsort_perm(k, bool(desc & 1))
            g = got[offs[a_i]:offs[a_i + 1]]
            if (desc & 2) and len(k) <= 64 and (g == -1).all():
                continue   # register variant: depth limit reported
            assert np.array_equal(g, exp), (a_i, len(k), desc)


def test_killer_weights_through_the_op(ops):
    """Weights that are an introsort killer: the o_n order of a long surviving column (output pass) and of a long
    eliminated column (single-vertex path) goes through the heap-sort branch of the restatement."""
    from util import introsort_killer
    for n in (120, 300, 700, 2500):
        ei = star(n)
        k, hit = introsort_killer(n - 1)
        assert hit
        deg = np.bincount(ei[0], minlength=n)
        hub = int(np.argmax(deg))
        w = np.empty(ei.shape[1])
        leaf = np.where(ei[0] == hub, ei[1], ei[0])          # the leaf of every directed entry
        rank_of_leaf = {int(v): i for i, v in enumerate(sorted(set(leaf.tolist())))}
        for ties in (False, True):   # with ties the single-vertex path cannot use a stable rank and runs the restatement too
          kk = np.floor(k / 2) if ties else k
          w[:] = [1.0 + kk[rank_of_leaf[int(v)]] for v in leaf]   # positive, symmetric (distinct per undirected edge without ties)
          for o_n in ("asc", "desc"):
            a = oracle.approximate_cholesky(ei, w, n, 0, "degree", o_n)            # hub survives: output pass
            b = gpu_call(ops, ei, w, n, 0, "degree", o_n)
            assert_same(b, a, f"killer star{n} output {o_n} ties={ties}")
            perm = np.array([v for v in range(n) if v != hub] + [hub])             # hub popped first (from the back)
            a = oracle.approximate_cholesky(ei, w, n, 5, "random", o_n, perm=perm)
            b = gpu_call(ops, ei, w, n, 5, "random", o_n, perm=perm)
            assert_same(b, a, f"killer star{n} eliminated {o_n} ties={ties}")


def test_identity_round_trip(ops):
    # reference tests/test_rlap.py:12-20
    for _ in range(3):
        a = torch.randn(100, 100).double()
        assert torch.allclose(a, ops.identity(a), atol=1e-8)
    a = torch.randn(37, 5, dtype=torch.float64, device="cuda")
    assert torch.equal(a, ops.identity(a))


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_appendix_c_goldens(ops, case):
    ei = MAKERS[case["graph"]](case["n"])
    out = gpu_call(ops, ei, None, case["n"], case["t"], case["o_v"], case["o_n"])
    exp = np.array(case["rows"], dtype=np.float64).reshape(-1, 3)
    assert_same(out, exp, case["name"])


SMALL = [("K6", clique(6), 6), ("P9", path(9), 9), ("star7", star(7), 7), ("grid5x6", grid2d(5, 6), 30),
         ("BA100_50", ba_graph(100, 50, 0), 100), ("BA500_3", ba_graph(500, 3, 1), 500)]


@pytest.mark.parametrize("o_v", ["degree", "random", "coarsen"])
@pytest.mark.parametrize("o_n", ["asc", "desc", "random"])
def test_small_graphs_all_modes(ops, o_v, o_n):
    for name, ei, n in SMALL:
        perm = np.random.RandomState(7).permutation(n) if o_v == "random" else None
        for t in sorted({0, 1, n // 2, n - 1, n + 5}):
            a = oracle.approximate_cholesky(ei, None, n, t, o_v, o_n, perm=perm, shuffle_seed=3)
            b = gpu_call(ops, ei, None, n, t, o_v, o_n, perm=perm, seed=3)
            assert_same(b, a, f"{name} t={t}")
        w = sym_weights(ei, n, 5)
        a = oracle.approximate_cholesky(ei, w, n, n // 2, o_v, o_n, perm=perm, shuffle_seed=4)
        b = gpu_call(ops, ei, w, n, n // 2, o_v, o_n, perm=perm, seed=4)
        assert_same(b, a, f"{name} weighted")


@pytest.mark.parametrize("n,m,o_v,o_n", [
    (2708, 2, "random", "asc"),      # Cora-sized stand-in (BASELINE config 2)
    (4096, 8, "random", "asc"),      # one graph of config 5
    (4096, 8, "degree", "asc"),
    (20000, 10, "degree", "asc"),    # config 3's shape, oracle-sized
    (20000, 10, "degree", "desc"),
    (20000, 7, "coarsen", "asc"),    # config 4's mode
    (5000, 40, "random", "random"),  # long columns: exercises the sequential fallback
])
def test_medium_ba_graphs(ops, n, m, o_v, o_n):
    ei = ba_graph(n, m, 100 + m)
    perm = np.random.RandomState(11).permutation(n) if o_v == "random" else None
    for w in (None, sym_weights(ei, n, 3)):
        a = oracle.approximate_cholesky(ei, w, n, n // 2, o_v, o_n, perm=perm, shuffle_seed=21)
        b = gpu_call(ops, ei, w, n, n // 2, o_v, o_n, perm=perm, seed=21)
        assert_same(b, a, f"BA({n},{m}) {o_v}/{o_n} weighted={w is not None}")


@pytest.mark.parametrize("o_n", ["asc", "desc", "random"])
def test_long_columns_random_order(ops, o_n):
    """o_v="random" meets hubs early: columns beyond the 64-slot batch candidates (wave path, <= 384 entries)
    and beyond the wave path (long-column path, LDS sort records + global scratch), with multi-edges
    created by earlier eliminations."""
    cases = []
    n = 1500
    ei = star(n)
    deg = np.bincount(ei[0], minlength=n)
    hub = int(np.argmax(deg))
    for pos in (0, 5, 700):          # the hub is popped first / after a few leaves / in the middle (pop = from the back)
        rest = [v for v in np.random.RandomState(pos).permutation(n) if v != hub]
        order = rest[:pos] + [hub] + rest[pos:]
        cases.append((f"star{n}@{pos}", ei, n, np.array(order[::-1])))
    n = 3000
    ei = ba_graph(n, 40, 11)
    deg = np.bincount(ei[0], minlength=n)
    hubs = list(np.argsort(-deg)[:12])
    rest = [v for v in np.random.RandomState(3).permutation(n) if v not in set(hubs)]
    order = rest[:400] + hubs + rest[400:]      # 400 eliminations first: fill-in and multi-edges reach the hubs
    cases.append(("BA3000_40 hubs", ei, n, np.array(order[::-1])))
    for name, ei, n, perm in cases:
        for w in (None, sym_weights(ei, n, 9)):
            t = n // 2
            a = oracle.approximate_cholesky(ei, w, n, t, "random", o_n, perm=perm, shuffle_seed=6)
            b = gpu_call(ops, ei, w, n, t, "random", o_n, perm=perm, seed=6)
            assert_same(b, a, f"{name} {o_n} {'unit' if w is None else 'weighted'}")


@pytest.mark.parametrize("o_v", ["degree", "coarsen"])
def test_long_serial_column_under_the_pq_orders(ops, o_v):
    """a column beyond the wave path's LDS under the PQ orders goes through the one-lane serial elimination while the
    helper wave waits -- it must be told that no help is needed before that starts (it used to spin through it and could run
    into its time limit).  K_2200: every column has 2,199 entries, the first pops are serial."""
    n = 2200
    ei = clique(n)
    for w in (None, sym_weights(ei, n, 4)):
        a = oracle.approximate_cholesky(ei, w, n, 3, o_v, "asc", shuffle_seed=2)
        b = gpu_call(ops, ei, w, n, 3, o_v, "asc", seed=2)
        assert_same(b, a, f"K{n} {o_v} {'unit' if w is None else 'weighted'}")


@pytest.mark.parametrize("wide", ["1", "0"])
@pytest.mark.parametrize("o_n", ["asc", "desc", "random"])
def test_wide_candidates_random_order(ops, monkeypatch, o_n, wide):
    """o_v="random", 128-slot candidates (chosen for graphs with 8 or more entries per vertex; RLAP_WIDE forces the choice either
    way here): columns of 65..128 entries stay inside the round -- two entries per lane, multi-edges folded in the candidate,
    both orders by the level-synchronous std::sort restatement over an index array.  Dense, sparse, weighted (tie-free) and
    unit-weight (all ties) graphs, a star (the hub takes the long-column path), a grid."""
    monkeypatch.setenv("RLAP_WIDE", wide)
    cases = [("BA3000_12", ba_graph(3000, 12, 1), 3000), ("BA700_60", ba_graph(700, 60, 3), 700), ("BA260_100", ba_graph(260, 100, 4), 260),
             ("BA20000_7", ba_graph(20000, 7, 5), 20000), ("BA2708_2", ba_graph(2708, 2, 6), 2708), ("star900", star(900), 900),
             ("grid40x30", grid2d(40, 30), 1200), ("K90", clique(90), 90)]
    for name, ei, n in cases:
        perm = np.random.RandomState(len(name)).permutation(n)
        for w in (None, sym_weights(ei, n, 9)):
            for t in sorted({n // 2, n - 1}):
                a = oracle.approximate_cholesky(ei, w, n, t, "random", o_n, perm=perm, shuffle_seed=6)
                b = gpu_call(ops, ei, w, n, t, "random", o_n, perm=perm, seed=6)
                assert_same(b, a, f"{name} t={t} {o_n} {'unit' if w is None else 'weighted'} wide={wide}")
    if o_n != "random":
        # a hub of 65..128 neighbours whose weights are an introsort killer with ties: the level-synchronous sort of the candidate
        # meets the depth limit and the sequential restatement takes over (heap-sort branch)
        from util import introsort_killer
        for n in (100, 128):
            ei = star(n)
            k, hit = introsort_killer(n - 1)
            assert hit
            hub = int(np.argmax(np.bincount(ei[0], minlength=n)))
            leaf = np.where(ei[0] == hub, ei[1], ei[0])
            rank_of_leaf = {int(v): i for i, v in enumerate(sorted(set(leaf.tolist())))}
            kk = np.floor(k / 2)
            w = np.array([1.0 + kk[rank_of_leaf[int(v)]] for v in leaf])
            for first in (0, 3):   # the hub is popped first / after three leaves
                rest = [v for v in range(n) if v != hub]
                perm = np.array((rest[:first] + [hub] + rest[first:])[::-1])
                a = oracle.approximate_cholesky(ei, w, n, 10, "random", o_n, perm=perm, shuffle_seed=6)
                b = gpu_call(ops, ei, w, n, 10, "random", o_n, perm=perm, seed=6)
                assert_same(b, a, f"killer star{n} hub@{first} {o_n} wide={wide}")


def test_wide_candidates_batched(ops):
    """128-slot candidates in a batched call of the 1024-thread shape (fewer graphs than CUs, dense graphs): every graph against
    the oracle, graph g with seed + g."""
    from rlap_amd import graphs
    G = 40
    eis, ns = [], []
    for g in range(G):
        n = 500 + 13 * g
        eis.append(torch.from_numpy(ba_graph(n, 16 + g % 5, 70 + g))); ns.append(n)
    big, node_ptr = graphs.batch_disjoint(eis, ns)
    perms = [np.random.RandomState(g).permutation(n) for g, n in enumerate(ns)]
    for o_n in ("asc", "random"):
        sc, rp = ops.approximate_cholesky_batched(big.cuda(), None, node_ptr, [n // 2 for n in ns], "random", o_n,
                                                  perm=torch.from_numpy(np.concatenate(perms)), seed=9)
        sc = sc.cpu().numpy()
        for g in range(G):
            ref = oracle.approximate_cholesky(eis[g].numpy(), None, ns[g], ns[g] // 2, "random", o_n, perm=perms[g], shuffle_seed=9 + g)
            got = sc[int(rp[g]):int(rp[g + 1])].copy()
            got[:, :2] -= int(node_ptr[g])
            assert_same(got, ref, f"batched wide graph {g} {o_n}")


@pytest.mark.parametrize("o_n", ["asc", "desc", "random"])
def test_long_surviving_columns(ops, o_n):
    """Output pass: surviving columns beyond the 512-entry tiers (LDS record array, up to 8192 entries) and
    beyond that (records in global scratch), unit weights (all ties) and tie-free weights."""
    for n in (3000, 12000):
        ei = star(n)
        for w in (None, sym_weights(ei, n, 2)):
            for t in (0, 50):
                a = oracle.approximate_cholesky(ei, w, n, t, "degree", o_n, shuffle_seed=8)
                b = gpu_call(ops, ei, w, n, t, "degree", o_n, seed=8)
                assert_same(b, a, f"star{n} t={t} {o_n} {'unit' if w is None else 'weighted'}")


def test_many_far_bucket_moves(ops):
    """PQ move order: a round whose moves spread over more than 256 buckets above the lowest, with hundreds of
    them beyond (hubs of degree >= 400 next to vertices of degree ~25) -- the counting placement's second class
    and its bitonic fall-back.  Leaves (degree 10) are eliminated 128 a round; every one touches 5 hubs and 5 mids."""
    rs = np.random.RandomState(5)
    n_leaf, n_mid, n_hub = 3000, 600, 400
    n = n_leaf + n_mid + n_hub
    hub0, mid0 = n_leaf + n_mid, n_leaf
    src, dst = [], []
    for a in range(n_hub):
        for b in range(a + 1, n_hub):
            src.append(hub0 + a); dst.append(hub0 + b)
    for v in range(n_leaf):
        for h in rs.choice(n_hub, 5, replace=False):
            src.append(v); dst.append(hub0 + int(h))
        for q in rs.choice(n_mid, 5, replace=False):
            src.append(v); dst.append(mid0 + int(q))
    src = np.array(src); dst = np.array(dst)
    ei = np.stack([np.concatenate([src, dst]), np.concatenate([dst, src])]).astype(np.int64)
    order = np.lexsort((ei[0], ei[1]))
    ei = ei[:, order]
    for o_n in ("asc", "desc"):
        for w in (None, sym_weights(ei, n, 4)):
            a = oracle.approximate_cholesky(ei, w, n, 2500, "degree", o_n, shuffle_seed=2)
            b = gpu_call(ops, ei, w, n, 2500, "degree", o_n, seed=2)
            assert_same(b, a, f"far buckets {o_n} {'unit' if w is None else 'weighted'}")


def _random_graph(rs, n, p, dense_hub):
    src, dst = np.nonzero(np.triu(rs.rand(n, n) < p, 1))
    if dense_hub and n > 3:   # one vertex adjacent to almost everything: long columns, shared targets
        h = int(rs.randint(n))
        others = np.array([v for v in range(n) if v != h and rs.rand() < 0.9], dtype=np.int64)
        src = np.concatenate([src, np.full(others.shape[0], h)]); dst = np.concatenate([dst, others])
    a = np.minimum(src, dst); b = np.maximum(src, dst)
    key = np.unique(a * n + b)
    a, b = key // n, key % n
    ei = np.stack([np.concatenate([a, b]), np.concatenate([b, a])]).astype(np.int64)
    return ei[:, np.lexsort((ei[0], ei[1]))]


@pytest.mark.parametrize("block", range(6))
def test_fuzz_small_graphs_against_oracle(ops, block):
    """Seeded random graphs (sparse to dense, with and without a hub), random mode, num_remove, unit / tie-heavy /
    tie-free weights: every result bit-exact against the oracle."""
    rs = np.random.RandomState(1000 + block)
    for trial in range(40):
        n = int(rs.choice([1, 2, 3, 5, 8, 17, 33, 40, 65, 90, 130, 200]))
        p = float(rs.choice([0.0, 0.02, 0.1, 0.3, 0.7, 1.0]))
        ei = _random_graph(rs, n, p, dense_hub=bool(rs.rand() < 0.3))
        o_v = str(rs.choice(["degree", "random", "coarsen"]))
        o_n = str(rs.choice(["asc", "desc", "random"]))
        t = int(rs.choice([0, 1, n // 3, n // 2, max(n - 1, 0), n + 3]))
        wkind = int(rs.randint(3))
        if wkind == 0 or ei.shape[1] == 0:
            w = None
        elif wkind == 1:   # few distinct values: ties everywhere, but not all equal
            w = sym_weights(ei, n, int(rs.randint(1 << 30)))
            w = np.round(w * 2) / 2 + 0.5
        else:
            w = sym_weights(ei, n, int(rs.randint(1 << 30)))
        perm = rs.permutation(n) if o_v == "random" else None
        seed = int(rs.randint(1 << 30))
        a = oracle.approximate_cholesky(ei, w, n, t, o_v, o_n, perm=perm, shuffle_seed=seed)
        b = gpu_call(ops, ei, w, n, t, o_v, o_n, perm=perm, seed=seed)
        assert_same(b, a, f"block {block} trial {trial}: n={n} p={p} {o_v}/{o_n} t={t} w={wkind}")


def test_reference_unit_test_shape(ops):
    # reference tests/test_rlap.py:39-61: BA(100, 50), ones((1,E)) weights, t=50, random/asc
    n = 100
    ei = torch.from_numpy(ba_graph(n, 50, 5))
    w = torch.ones((1, ei.shape[1]))
    sc = ops.approximate_cholesky(edge_index=ei, edge_weights=w, num_nodes=n, num_remove=50, o_v="random", o_n="asc")
    assert sc.dtype == torch.double
    idx = torch.Tensor(sc[:, :2]).long().t()
    adj = torch.zeros(n, n)
    adj[idx[0], idx[1]] = 1
    assert torch.allclose(adj, adj.t(), atol=1e-8)


def test_edge_cases(ops):
    # empty graph, isolated vertices, zero weights, duplicates, (E,) and (1,E) weights, t out of range
    out = ops.approximate_cholesky(torch.zeros((2, 0), dtype=torch.int64).cuda(), None, 5, 3, "degree", "asc")
    assert out.shape == (0, 3)
    ei = np.array([[0, 1, 0, 1, 1, 2, 4, 5], [1, 0, 1, 0, 2, 1, 5, 4]])
    w = np.array([1.0, 1.0, 0.5, 0.5, 0.0, 0.0, 2.0, 2.0])
    for t in (0, 2, 6, 100):
        a = oracle.approximate_cholesky(ei, w, 7, t, "degree", "asc")
        b = gpu_call(ops, ei, w, 7, t, "degree", "asc")
        assert_same(b, a, f"edge t={t}")
        b2 = ops.approximate_cholesky(torch.from_numpy(ei), torch.from_numpy(w).reshape(1, -1), 7, t, "degree", "asc").numpy()
        assert_same(b2, a, "(1,E) weights, CPU input")
    with pytest.raises(ValueError):
        ops.approximate_cholesky(torch.tensor([[0, 1, 2], [1, 0, 1]]).cuda(), None, 3, 1, "degree", "asc")
    with pytest.raises(ValueError):
        ops.approximate_cholesky(torch.tensor([[0, 9], [9, 0]]).cuda(), None, 3, 1, "degree", "asc")
    with pytest.raises(AssertionError):
        ops.approximate_cholesky(torch.tensor([[0, 1], [1, 0]]).cuda(), None, 2, 1, "bogus", "asc")


def test_torch_op_schema(ops):
    # torch.ops.extension_cpp.approximate_cholesky(edge_info, ...) called with kwargs as rlap/ops.py:52-58 does
    n = 60
    ei = ba_graph(n, 4, 2)
    info = torch.from_numpy(np.concatenate([ei.astype(np.float64), np.ones((1, ei.shape[1]))], 0)).t().contiguous()
    out = torch.ops.extension_cpp.approximate_cholesky.default(edge_info=info, num_nodes=n, num_remove=30, o_v="degree", o_n="asc")
    a = oracle.approximate_cholesky(ei, None, n, 30, "degree", "asc")
    assert_same(out.numpy(), a, "torch op")
    x = torch.randn(10, 4).double()
    assert torch.equal(torch.ops.extension_cpp.identity.default(a=x), x)


def test_batched_equals_separate_calls(ops):
    from rlap_amd import graphs
    rs = np.random.RandomState(0)
    eis, ns, ts = [], [], []
    for g in range(12):
        n = int(rs.choice([1, 2, 50, 200, 333]))
        m = 3 if n > 10 else 1
        ei = ba_graph(n, m, 40 + g) if n > m else np.zeros((2, 0), dtype=np.int64)
        eis.append(torch.from_numpy(ei)); ns.append(n); ts.append(n // 2)
    big, node_ptr = graphs.batch_disjoint(eis, ns)
    for o_v, o_n in [("degree", "asc"), ("random", "desc"), ("coarsen", "asc"), ("degree", "random"), ("random", "random")]:
        perms = [np.random.RandomState(g).permutation(n) for g, n in enumerate(ns)]
        perm = torch.from_numpy(np.concatenate(perms)) if o_v == "random" else None
        sc, row_ptr = ops.approximate_cholesky_batched(big.cuda(), None, node_ptr, ts, o_v, o_n, perm=perm, seed=5)
        sc = sc.cpu().numpy()
        for g in range(12):
            # the batched contract (include/rlap_hip.h): graph g == a separate call on graph g with seed + g
            off = int(node_ptr[g])
            a = oracle.approximate_cholesky(eis[g].numpy(), None, ns[g], ts[g], o_v, o_n, perm=perms[g], shuffle_seed=5 + g)
            b = sc[int(row_ptr[g]):int(row_ptr[g + 1])].copy()
            b[:, :2] -= off
            assert_same(b, a, f"graph {g} {o_v}/{o_n}")
            if g in (3, 7):   # ... and == a separate call of the op itself
                c = gpu_call(ops, eis[g].numpy(), None, ns[g], ts[g], o_v, o_n, perm=perms[g], seed=5 + g)
                assert_same(b, c, f"graph {g} {o_v}/{o_n} vs single call")
        # whole-batch invariants for every mode: symmetric edge set inside each graph's id range
        for g in range(12):
            b = sc[int(row_ptr[g]):int(row_ptr[g + 1])]
            if b.shape[0]:
                assert b[:, :2].min() >= int(node_ptr[g]) and b[:, :2].max() < int(node_ptr[g + 1])
                fw = set(map(tuple, b[:, :2].astype(int)))
                assert all((c, r) in fw for r, c in fw)


@pytest.mark.parametrize("o_v,o_n", [("degree", "asc"), ("degree", "random"), ("random", "asc"), ("random", "desc"), ("coarsen", "asc"), ("random", "random")])
def test_batched_many_graphs(ops, o_v, o_n):
    """More graphs than twice the CUs: the batch runs with the 256-thread workgroup shape (four per CU).
    Every 16th graph is compared with the oracle; sizes include dense ones whose columns leave the 32/64-slot
    candidates (single-vertex wave path) and a star whose centre needs the long-column path."""
    from rlap_amd import graphs
    rs = np.random.RandomState(1)
    eis, ns, ts = [], [], []
    G = 640
    for g in range(G):
        kind = g % 16
        if kind == 0:
            n = 700; ei = star(n)
        elif kind == 1:
            n = 260; ei = ba_graph(n, 60, 500 + g)
        else:
            n = int(rs.choice([3, 40, 150, 300])); ei = ba_graph(n, min(5, n - 1), 500 + g)
        eis.append(torch.from_numpy(ei)); ns.append(n); ts.append(n // 2)
    big, node_ptr = graphs.batch_disjoint(eis, ns)
    perms = [np.random.RandomState(g).permutation(n) for g, n in enumerate(ns)]
    perm = torch.from_numpy(np.concatenate(perms)) if o_v == "random" else None
    sc, row_ptr = ops.approximate_cholesky_batched(big.cuda(), None, node_ptr, ts, o_v, o_n, perm=perm, seed=5)
    sc = sc.cpu().numpy()
    for g in list(range(0, G, 16)) + list(range(1, G, 16)) + list(range(2, G, 37)):
        off = int(node_ptr[g])
        a = oracle.approximate_cholesky(eis[g].numpy(), None, ns[g], ts[g], o_v, o_n, perm=perms[g], shuffle_seed=5 + g)
        b = sc[int(row_ptr[g]):int(row_ptr[g + 1])].copy()
        b[:, :2] -= off
        assert_same(b, a, f"graph {g} {o_v}/{o_n}")


def test_determinism_run_twice(ops):
    n = 3000
    ei = ba_graph(n, 6, 9)
    a = gpu_call(ops, ei, None, n, n // 2, "degree", "asc")
    b = gpu_call(ops, ei, None, n, n // 2, "degree", "asc")
    assert np.array_equal(a, b)


def test_device_native_adapters(ops):
    # the PyGCL-style adapter (scripts/augmentor_benchmarks.py:75-96) without leaving the GPU
    from rlap_amd.adapters import rLap, rLapDGL
    n = 400
    ei = ba_graph(n, 5, 12)
    x = torch.randn(n, 8, device="cuda")
    aug = rLap(0.5, o_v="degree", o_n="asc")
    g = aug(x, torch.from_numpy(ei).cuda(), None)
    ref = oracle.approximate_cholesky(ei, None, n, n // 2, "degree", "asc")
    assert g.edge_index.is_cuda and g.edge_weights is None and g.x is x
    assert np.array_equal(g.edge_index.cpu().numpy(), ref[:, :2].astype(np.int64).T)
    gw = rLap(0.5, o_v="degree", o_n="asc", keep_weights=True)(x, torch.from_numpy(ei).cuda(), None)
    assert np.array_equal(gw.edge_weights.cpu().numpy(), ref[:, 2])
    ei2, nn = rLapDGL(0.5, o_v="degree", o_n="asc").augment((torch.from_numpy(ei).cuda(), n))
    assert nn == n and np.array_equal(ei2.cpu().numpy(), ref[:, :2].astype(np.int64).T)


def test_to_undirected_on_device_feeds_the_op(ops):
    # the step before the path: one-directional edges -> symmetric, coalesced, on the device
    from rlap_amd import graphs
    rs = np.random.RandomState(0)
    n = 500
    src = rs.randint(0, n, 3000); dst = rs.randint(0, n, 3000)
    keep = src != dst
    ei = torch.from_numpy(np.stack([src[keep], dst[keep]])).cuda()
    und = graphs.to_undirected(ei, n)
    assert und.is_cuda
    a = und.cpu().numpy()
    fw = set(map(tuple, a.T))
    assert all((c, r) in fw for r, c in fw) and len(fw) == a.shape[1]
    got = ops.approximate_cholesky(und, None, n, n // 2, "degree", "asc").numpy()
    ref = oracle.approximate_cholesky(a, None, n, n // 2, "degree", "asc")
    assert_same(got, ref, "to_undirected -> approximate_cholesky")


def test_ppr_diffusion_adapter(ops):
    # the only caller that uses the output WEIGHTS (augmentor_benchmarks.py:121-171)
    from rlap_amd.adapters import rLapPPRDiffusion, compute_ppr
    n = 300
    ei = ba_graph(n, 4, 21)
    x = torch.randn(n, 4, device="cuda")
    aug = rLapPPRDiffusion(0.5, o_v="degree", o_n="asc", alpha=0.2, eps=1e-4)
    g = aug(x, torch.from_numpy(ei).cuda(), None)
    # same thing from the oracle's output with numpy
    ref = oracle.approximate_cholesky(ei, None, n, n // 2, "degree", "asc")
    nodes = np.unique(ref[:, :2].astype(np.int64))
    rel = -np.ones(n, dtype=np.int64); rel[nodes] = np.arange(len(nodes))
    A = np.zeros((len(nodes), len(nodes)))
    np.add.at(A, (rel[ref[:, 0].astype(int)], rel[ref[:, 1].astype(int)]), ref[:, 2])
    d = A.sum(1); dinv = np.where(d > 0, d ** -0.5, 0)
    S = 0.2 * np.linalg.inv(np.eye(len(nodes)) - 0.8 * (dinv[:, None] * A * dinv[None, :]))
    S[S < 1e-4] = 0
    d2 = S.sum(1); d2inv = np.where(d2 > 0, d2 ** -0.5, 0)
    S = d2inv[:, None] * S * d2inv[None, :]       # the closing transition_matrix('sym') of PyGCL's compute_ppr (unpinned: PyGCL absent)
    got = np.zeros_like(S)
    gi = g.edge_index.cpu().numpy(); got[rel[gi[0]], rel[gi[1]]] = g.edge_weights.cpu().numpy()
    assert np.allclose(got, S, rtol=1e-8, atol=1e-12)
    assert aug(x, torch.from_numpy(ei).cuda(), None) is g      # cached like the reference


def test_large_graph_properties(ops):
    """Size-independent checks at a size the oracle is not run for (invariants of a16-a19)."""
    from rlap_amd import graphs
    n = 300_000
    ei = graphs.barabasi_albert(n, 6, 77).cuda()
    for o_v in ("degree", "coarsen"):
        sc = ops.approximate_cholesky(ei, None, n, n // 2, o_v, "asc", seed=3, return_device="same")
        r, c, w = sc[:, 0].long(), sc[:, 1].long(), sc[:, 2]
        assert (w > 0).all()
        # every surviving undirected edge appears exactly twice, once per direction, with matching weights
        key_f = r * n + c
        key_b = c * n + r
        sf, of = torch.sort(key_f)
        sb, ob = torch.sort(key_b)
        assert torch.equal(sf, sb)
        assert torch.allclose(w[of], w[ob], rtol=1e-12, atol=0)
        assert (sf[1:] != sf[:-1]).all()                      # multi-edges were merged
        # exactly n - min(t, n-1) vertices may appear as columns; none of them eliminated twice
        cols = torch.unique(c)
        assert cols.numel() <= n - n // 2
        # run twice: same result (deterministic for degree; coarsen with a fixed seed)
        sc2 = ops.approximate_cholesky(ei, None, n, n // 2, o_v, "asc", seed=3, return_device="same")
        assert torch.equal(sc, sc2)


# ---------------------------------------------------------------------------
# BASELINE config 5 at full size: 1024 x BA(4096, m=8), num_remove = 2048 each, one batched call per mode;
# EVERY graph is compared with the oracle for the two headline modes (about 20 s of CPU).
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config5():
    from rlap_amd import graphs
    G, n, m = 1024, 4096, 8
    eis = [graphs.barabasi_albert(n, m, 1000 + g) for g in range(G)]
    big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
    perms = [np.random.RandomState(g).permutation(n) for g in range(G)]
    return G, n, eis, big.cuda(), node_ptr, perms


@pytest.mark.parametrize("o_v,o_n,every", [("degree", "asc", 1), ("random", "asc", 1), ("coarsen", "asc", 16), ("degree", "random", 16), ("random", "random", 16)])
def test_config5_full_size(ops, config5, o_v, o_n, every):
    G, n, eis, big, node_ptr, perms = config5
    perm = torch.from_numpy(np.concatenate(perms)) if o_v == "random" else None
    sc, rp = ops.approximate_cholesky_batched(big, None, node_ptr, [n // 2] * G, o_v, o_n, perm=perm, seed=5)
    assert ops.last_stats["n_eliminated"] == G * (n // 2)
    sc = sc.cpu().numpy()
    for g in range(0, G, every):
        ref = oracle.approximate_cholesky(eis[g].numpy(), None, n, n // 2, o_v, o_n, perm=perms[g], shuffle_seed=5 + g)
        got = sc[int(rp[g]):int(rp[g + 1])].copy()
        got[:, :2] -= g * n
        assert_same(got, ref, f"config 5 graph {g} {o_v}/{o_n}")


@pytest.mark.parametrize("o_v", ["degree", "random"])
def test_config5_run_to_run(ops, config5, o_v):
    """The batched call is deterministic: slots of the append pool are handed to the 1024 workgroups in whatever order they
    ask, the rows returned must not depend on it (20 runs compared on the device)."""
    G, n, eis, big, node_ptr, perms = config5
    perm = torch.from_numpy(np.concatenate(perms)) if o_v == "random" else None
    first = None
    for it in range(20):
        sc, rp = ops.approximate_cholesky_batched(big, None, node_ptr, [n // 2] * G, o_v, "asc", perm=perm, seed=5, return_device="same")
        if first is None:
            first = sc.clone()
        else:
            assert sc.shape == first.shape and bool(torch.equal(sc, first)), f"run {it} differs from run 0" + _where(sc.cpu().numpy(), first.cpu().numpy())


def test_bad_perm_is_rejected(ops):
    # the injected node_id vector is validated on the device (range + duplicates) -> ValueError, no fault
    n = 300
    ei = torch.from_numpy(ba_graph(n, 3, 1)).cuda()
    good = np.random.RandomState(0).permutation(n)
    for bad in (np.where(good == 5, n + 7, good), np.where(good == 5, -1, good), np.where(good == 5, 6, good)):
        with pytest.raises(ValueError):
            ops.approximate_cholesky(ei, None, n, n // 2, "random", "asc", perm=torch.from_numpy(bad))
    # batched: GLOBAL ids where LOCAL ids are expected (the likeliest mistake)
    from rlap_amd import graphs
    eis = [torch.from_numpy(ba_graph(50, 3, g)) for g in range(3)]
    big, node_ptr = graphs.batch_disjoint(eis, [50] * 3)
    glob = torch.cat([torch.from_numpy(np.random.RandomState(g).permutation(50)) + 50 * g for g in range(3)])
    with pytest.raises(ValueError):
        ops.approximate_cholesky_batched(big.cuda(), None, node_ptr, [25] * 3, "random", "asc", perm=glob)
    # and the handle still works afterwards
    a = oracle.approximate_cholesky(ei.cpu().numpy(), None, n, n // 2, "random", "asc", perm=good)
    b = gpu_call(ops, ei.cpu().numpy(), None, n, n // 2, "random", "asc", perm=good)
    assert_same(b, a, "after rejected perms")


@pytest.mark.parametrize("kind", ["pool", "log", "rng", "scratch"])
def test_overflow_retry_path(ops, kind):
    """The growth limits of the workspace (append pool, PQ log, uniform table, long-column scratch) are hit on
    purpose (rlap_debug_set_limits): the call must repeat itself and still return the oracle's rows."""
    if kind == "scratch":
        n = 12000; ei = star(n); w = sym_weights(ei, n, 2); o_v, o_n, t = "degree", "asc", 0     # a surviving column beyond 8192 entries
    else:
        n = 3000; ei = ba_graph(n, 6, 9); w = None; o_v, o_n, t = "degree", "asc", n // 2
    ref = oracle.approximate_cholesky(ei, w, n, t, o_v, o_n)
    lim = {"pool": dict(pool_factor=0.0), "log": dict(log_factor=0.0), "rng": dict(rng_len=100), "scratch": dict(scratch_entries=64)}[kind]
    ops.debug_set_limits(**lim)
    try:
        got = gpu_call(ops, ei, w, n, t, o_v, o_n)
        retries = ops.last_stats["n_retries"]
    finally:
        ops.debug_set_limits()
    assert_same(got, ref, f"retry {kind}")
    assert retries > 0, f"{kind}: the limit was not hit"
    got = gpu_call(ops, ei, w, n, t, o_v, o_n)
    assert ops.last_stats["n_retries"] == 0
    assert_same(got, ref, f"after retry {kind}")


@pytest.mark.parametrize("o_v", ["degree", "random"])
def test_overflow_retry_path_batched(ops, config5, o_v):
    """The same in a batched call: some of the 1024 workgroups meet the limit in the middle of a round, others finish; the
    repeated attempt must not see anything of the first one."""
    G, n, eis, big, node_ptr, perms = config5
    perm = torch.from_numpy(np.concatenate(perms)) if o_v == "random" else None
    ref, rp0 = ops.approximate_cholesky_batched(big, None, node_ptr, [n // 2] * G, o_v, "asc", perm=perm, seed=5, return_device="same")
    assert ops.last_stats["n_retries"] == 0
    for lim in (dict(pool_factor=0.0), dict(log_factor=0.0), dict(rng_len=1000)):
        if o_v == "random" and "log_factor" in lim:
            continue   # no PQ log in this mode
        ops.debug_set_limits(**lim)
        try:
            sc, rp = ops.approximate_cholesky_batched(big, None, node_ptr, [n // 2] * G, o_v, "asc", perm=perm, seed=5, return_device="same")
            retries = ops.last_stats["n_retries"]
        finally:
            ops.debug_set_limits()
        assert retries > 0, f"{lim}: the limit was not hit"
        assert sc.shape == ref.shape and bool(torch.equal(sc, ref)), f"{o_v} {lim}: differs after {retries} retries" + _where(sc.cpu().numpy(), ref.cpu().numpy())


def test_two_threads_two_streams(ops):
    """Re-entrancy (reference: a fresh ApproximateCholesky per call, py_api_binder.cc:57): two Python threads, each on
    its own stream, call the op concurrently; every result is bit-exact."""
    import threading
    jobs = []
    for k in range(2):
        n = 6000 + 500 * k
        ei = ba_graph(n, 5 + k, 30 + k)
        perm = np.random.RandomState(k).permutation(n)
        jobs.append((n, ei, perm, oracle.approximate_cholesky(ei, None, n, n // 2, "degree", "asc"),
                     oracle.approximate_cholesky(ei, None, n, n // 2, "random", "asc", perm=perm)))
    errors = []

    def worker(k):
        try:
            n, ei, perm, ref_d, ref_r = jobs[k]
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                ei_t = torch.from_numpy(ei).cuda()
                for _ in range(6):
                    a = ops.approximate_cholesky(ei_t, None, n, n // 2, "degree", "asc").numpy()
                    b = ops.approximate_cholesky(ei_t, None, n, n // 2, "random", "asc", perm=torch.from_numpy(perm)).numpy()
                    assert a.shape == ref_d.shape and np.array_equal(a, ref_d)
                    assert b.shape == ref_r.shape and np.array_equal(b, ref_r)
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors


def test_from_edges_fused_symmetrise_and_num_nodes(ops):
    # one-directional edges in, to_undirected + coalesce + num_nodes = max + 1 inside the op
    rs = np.random.RandomState(3)
    n = 700
    src = rs.randint(0, n - 40, 4000); dst = rs.randint(0, n - 40, 4000)      # ids stop short of n: max + 1 < n
    keep = src != dst
    src, dst = src[keep], dst[keep]
    from util import symmetrize
    und = symmetrize(src.astype(np.int64), dst.astype(np.int64), n)
    n_ref = int(und.max()) + 1
    for o_v, o_n in (("degree", "asc"), ("coarsen", "asc"), ("degree", "random")):
        ref = oracle.approximate_cholesky(und, None, n_ref, int(0.5 * n_ref), o_v, o_n, shuffle_seed=11)
        got, nn = ops.approximate_cholesky_from_edges(torch.from_numpy(np.stack([src, dst])).cuda(), None, None, None, o_v, o_n,
                                                      remove_frac=0.5, symmetrize=True, seed=11)
        assert nn == n_ref and got.is_cuda
        assert_same(got.cpu().numpy(), ref, f"from_edges {o_v}/{o_n}")
    # weighted, both directions present once: PyG to_undirected(reduce="add") semantics = summed duplicates
    w1 = rs.uniform(0.5, 1.5, src.shape[0])
    both_r = np.concatenate([src, dst]); both_c = np.concatenate([dst, src]); both_w = np.concatenate([w1, w1])
    ref = oracle.approximate_cholesky(np.stack([both_r, both_c]), both_w, n_ref, 100, "degree", "asc")
    got, _ = ops.approximate_cholesky_from_edges(torch.from_numpy(np.stack([src, dst])).cuda(), torch.from_numpy(w1).cuda(), None, 100,
                                                 "degree", "asc", symmetrize=True)
    a, b = canonical(got.cpu().numpy()), canonical(ref)
    assert a.shape == b.shape and np.array_equal(a[:, :2], b[:, :2]) and np.allclose(a[:, 2], b[:, 2], rtol=1e-12)
    # o_v="random" without an injected perm: the node_id vector is drawn on the device -> valid, symmetric, reproducible
    ei = torch.from_numpy(ba_graph(500, 4, 3)).cuda()
    r1, _ = ops.approximate_cholesky_from_edges(ei, None, 500, 250, "random", "asc", symmetrize=False, seed=99)
    r2, _ = ops.approximate_cholesky_from_edges(ei, None, 500, 250, "random", "asc", symmetrize=False, seed=99)
    r3, _ = ops.approximate_cholesky_from_edges(ei, None, 500, 250, "random", "asc", symmetrize=False, seed=100)
    assert torch.equal(r1, r2) and not (r1.shape == r3.shape and torch.equal(r1, r3))
    fw = set(map(tuple, r1[:, :2].long().cpu().numpy()))
    assert all((c, r) in fw for r, c in fw)
    assert torch.unique(r1[:, 1]).numel() <= 250


def test_adapter_num_nodes_rule(ops):
    # reference rule (augmentor_benchmarks.py:77): num_nodes = edge_index.max() + 1, even when x has more rows
    from rlap_amd.adapters import rLap
    n = 400
    ei = ba_graph(n, 5, 12)
    x = torch.randn(n + 30, 8, device="cuda")            # 30 trailing isolated nodes
    g = rLap(0.5, o_v="degree", o_n="asc")(x, torch.from_numpy(ei).cuda(), None)
    ref = oracle.approximate_cholesky(ei, None, n, n // 2, "degree", "asc")
    assert np.array_equal(g.edge_index.cpu().numpy(), ref[:, :2].astype(np.int64).T)
    g2 = rLap(0.5, o_v="degree", o_n="asc", num_nodes_from_x=True)(x, torch.from_numpy(ei).cuda(), None)
    ref2 = oracle.approximate_cholesky(ei, None, n + 30, (n + 30) // 2, "degree", "asc")
    assert np.array_equal(g2.edge_index.cpu().numpy(), ref2[:, :2].astype(np.int64).T)


def _load_edge_list(path):
    """(2,E) int64 from .npy / .npz (first array) / whitespace text with two columns."""
    if path.endswith(".npy"):
        a = np.load(path)
    elif path.endswith(".npz"):
        z = np.load(path); a = z[z.files[0]]
    else:
        a = np.loadtxt(path, dtype=np.int64)
    a = np.asarray(a, dtype=np.int64)
    return a if a.shape[0] == 2 else a.T


@pytest.mark.parametrize("env,o_v", [("RLAP_CORA_EDGES", "random"), ("RLAP_ARXIV_EDGES", "coarsen")])
def test_real_datasets_if_present(ops, env, o_v):
    """BASELINE configs 2 (Cora) and 4 (ogbn-arxiv): the datasets are not in the image; point the environment variable
    at an edge list (.npy/.npz/.txt, either orientation) to run them.  The BA stand-ins above cover the same modes."""
    path = os.environ.get(env)
    if not path or not os.path.exists(path):
        pytest.skip(f"{env} not set: dataset absent from this image")
    from util import symmetrize
    raw = _load_edge_list(path)
    keep = raw[0] != raw[1]
    n = int(raw.max()) + 1
    und = symmetrize(raw[0][keep], raw[1][keep], n)      # scripts/node_shared.py:326-327 (to_undirected)
    perm = np.random.RandomState(0).permutation(n) if o_v == "random" else None
    ref = oracle.approximate_cholesky(und, None, n, n // 2, o_v, "asc", perm=perm, shuffle_seed=3)
    got = gpu_call(ops, und, None, n, n // 2, o_v, "asc", perm=perm, seed=3)
    assert_same(got, ref, f"{env} {o_v}/asc")


def test_exchange_format_pack_unpack(ops):
    """The 16-byte exchange rows of the multi-GPU all-gather (rlap_pack_rows / rlap_unpack_rows): bit-identical to the torch
    formulation the CPU (gloo) tests use, and a lossless round trip."""
    from rlap_amd import distributed as D
    rs = np.random.RandomState(4)
    m = 100_003
    sc = torch.from_numpy(np.stack([rs.randint(0, 1 << 30, m).astype(np.float64), rs.randint(0, 1 << 30, m).astype(np.float64),
                                    rs.rand(m) * 3 + 1e-9], axis=1))
    p_cpu = D._pack_rows(sc)
    p_gpu = D._pack_rows(sc.cuda())
    assert p_gpu.is_cuda and torch.equal(p_gpu.cpu(), p_cpu)
    back = D._unpack_rows(p_gpu)
    assert torch.equal(back.cpu(), sc) and torch.equal(D._unpack_rows(p_cpu), sc)
    assert D._pack_rows(sc[:0].cuda()).shape == (0, 2)


def test_sorted_input_skips_the_coo_sort(ops, monkeypatch):
    """Large inputs are looked at before they are sorted: a COO in (col,row) order needs no sort, one in (row,col) order (PyG
    coalesce) is read transposed -- the same matrix when it is exactly symmetric, which the twin check verifies; else the call
    repeats itself with the sort.  RLAP_SORT_SKIP_MIN lowers the size threshold so that small graphs take these paths."""
    monkeypatch.setenv("RLAP_SORT_SKIP_MIN", "1")
    n = 3000
    ei = ba_graph(n, 6, 9)                       # sorted by (col, row)
    rc_order = np.lexsort((ei[1], ei[0]))        # the same entries sorted by (row, col)
    ei_rc = ei[:, rc_order]
    shuffled = ei[:, np.random.RandomState(1).permutation(ei.shape[1])]
    w = sym_weights(ei, n, 3)
    for o_v, o_n in (("degree", "asc"), ("coarsen", "asc"), ("degree", "random")):
        for wts in (None, w):
            ref = oracle.approximate_cholesky(ei, wts, n, n // 2, o_v, o_n, shuffle_seed=4)
            got = gpu_call(ops, ei, wts, n, n // 2, o_v, o_n, seed=4)
            assert ops.last_stats["reserved"] == 1 and ops.last_stats["n_retries"] == 0
            assert_same(got, ref, f"(col,row)-sorted {o_v}/{o_n}")
            got = gpu_call(ops, ei_rc, None if wts is None else wts[rc_order], n, n // 2, o_v, o_n, seed=4)
            assert ops.last_stats["reserved"] == 2 and ops.last_stats["n_retries"] == 0
            assert_same(got, ref, f"(row,col)-sorted {o_v}/{o_n}")
            got = gpu_call(ops, shuffled, None, n, n // 2, o_v, o_n, seed=4) if wts is None else got
            if wts is None:
                assert ops.last_stats["reserved"] == 0
                assert_same(got, ref, f"shuffled {o_v}/{o_n}")
    # (row,col)-sorted input whose two directions differ in the last bits (inside isApprox's tolerance): the transposed reading
    # would swap them -- the call notices and repeats itself reading the input as given
    w2 = w.copy()
    upper = ei[0] < ei[1]
    w2[upper] = w2[upper] * (1 + 2e-16)
    ref = oracle.approximate_cholesky(ei_rc, w2[rc_order], n, n // 2, "degree", "asc")
    got = gpu_call(ops, ei_rc, w2[rc_order], n, n // 2, "degree", "asc")
    assert ops.last_stats["n_retries"] == 1 and ops.last_stats["reserved"] == 0
    assert_same(got, ref, "almost symmetric, (row,col)-sorted")
    # duplicates in sorted input are summed in input order as before
    eid = np.concatenate([ei, ei[:, :50]], axis=1)
    eid = eid[:, np.lexsort((eid[0], eid[1]))]
    wd = np.ones(eid.shape[1])
    # (duplicating one direction only makes the matrix asymmetric: duplicate both directions of the first 25 undirected edges)
    a, b = ei[0, :25], ei[1, :25]
    eid = np.concatenate([ei, np.stack([a, b]), np.stack([b, a])], axis=1)
    eid = eid[:, np.lexsort((eid[0], eid[1]))]
    ref = oracle.approximate_cholesky(eid, None, n, n // 2, "degree", "asc")
    got = gpu_call(ops, eid, None, n, n // 2, "degree", "asc")
    assert ops.last_stats["reserved"] == 1
    assert_same(got, ref, "sorted input with duplicates")
