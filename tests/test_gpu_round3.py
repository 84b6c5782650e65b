"""GPU tests added in round 3: the config holes of VERDICT r2 (config 4's stand-in at full size, the GRACE-shaped loop),
the workspace contract (torch-owned arena), one meaning of `seed`, rejected input that must not reach the output pass,
and the debug poison mode (stale-read detector).  All through the C ABI, compared with the CPU oracle."""
import os

import numpy as np
import pytest
import torch

import oracle
from util import ba_graph, star, sym_weights
from test_gpu_parity import assert_same, gpu_call, _where

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from rlap_amd import ops as _ops
    return _ops


def test_config4_standin_full_size_coarsen(ops):
    """BASELINE config 4's stand-in at FULL size: BA(169,343, 7) -- ogbn-arxiv's node count and density -- o_v="coarsen",
    num_remove = N/2, every row against the oracle (about 0.5 s of CPU)."""
    from rlap_amd import graphs
    n = 169343
    ei = graphs.barabasi_albert(n, 7, 4)
    out = ops.approximate_cholesky(ei.cuda(), None, n, n // 2, "coarsen", "asc", seed=11).numpy()
    assert ops.last_stats["n_retries"] == 0
    ref = oracle.approximate_cholesky(ei.numpy(), None, n, n // 2, "coarsen", "asc", shuffle_seed=11)
    assert_same(out, ref, "BA(169343,7) coarsen")


def test_grace_shaped_loop(ops):
    """The consumer of BASELINE config 4 (scripts/node_shared.py:259-266: aug1(...), aug2(...) inside Encoder.forward, every
    training step): 20 steps, two adapter calls per step interleaved with dense work on the same stream while torch's
    allocator is busy; every view equals the oracle's, and after the first step nothing grows (no retry, no new arena)."""
    from rlap_amd import graphs
    from rlap_amd.adapters import rLap
    n = 20000
    ei = graphs.barabasi_albert(n, 7, 6)
    ei_d = ei.cuda()
    x = torch.randn(n, 128, device="cuda")
    w1 = torch.randn(128, 256, device="cuda")
    w2 = torch.randn(256, 128, device="cuda")
    fr = (0.3, 0.45)
    lib, hobj = ops._handle_obj(torch.device("cuda", torch.cuda.current_device()))
    arena = None
    for step in range(20):
        views = []
        hsum = None
        for k in range(2):
            aug = rLap(fr[k], o_v="coarsen", o_n="asc", keep_weights=True, seed=100 + 2 * step + k)
            g = aug(x, ei_d, None)
            if step > 0:
                assert ops.last_stats["n_retries"] == 0, f"step {step} view {k}: workspace grew"
            # the encoder's dense work, same stream, allocator traffic in between (shapes change with the view)
            hdn = torch.relu(x @ w1) @ w2
            agg = torch.zeros_like(hdn).index_add_(0, g.edge_index[1], hdn[g.edge_index[0]] * g.edge_weights[:, None].float())
            hsum = agg if hsum is None else hsum + agg
            views.append(g)
        assert torch.isfinite(hsum).all()
        if step == 0:
            arena = (hobj.ws.data_ptr(), hobj.ws.numel(), hobj.rng.data_ptr())
        else:
            assert (hobj.ws.data_ptr(), hobj.ws.numel(), hobj.rng.data_ptr()) == arena, "the arena moved after step 1"
        for k, g in enumerate(views):
            ref = oracle.approximate_cholesky(ei.numpy(), None, n, int(fr[k] * n), "coarsen", "asc", shuffle_seed=100 + 2 * step + k)
            got = torch.cat([g.edge_index.t().double(), g.edge_weights[:, None]], dim=1).cpu().numpy()
            assert_same(got, ref, f"GRACE loop step {step} view {k}")


def _non_torch_bytes():
    free, total = torch.cuda.mem_get_info()
    return (total - free) - torch.cuda.memory_reserved()


def test_workspace_comes_from_torch(ops):
    """Row b4 (reference: the result tensor is torch's, py_api_binder.cc:42): the op allocates nothing outside
    torch's caching allocator -- device memory not accounted to torch does not move when a call needs a larger arena -- and it
    runs while torch holds 90 % of the device."""
    from rlap_amd import graphs
    n0 = 3000
    ei0 = graphs.barabasi_albert(n0, 5, 1).cuda()
    ops.approximate_cholesky(ei0, None, n0, n0 // 2, "degree", "asc")      # warm-up: code objects, scratch, the handle's tables
    ops.approximate_cholesky(ei0, None, n0, n0 // 2, "random", "asc", seed=1)
    torch.cuda.synchronize()
    before = _non_torch_bytes()
    free, total = torch.cuda.mem_get_info()
    hog = torch.empty(int(0.9 * total) - torch.cuda.memory_allocated(), dtype=torch.uint8, device="cuda")   # torch now holds >= 90 %
    try:
        n = 120000                                                             # a size class this handle has not seen
        ei = graphs.barabasi_albert(n, 8, 2)
        out = ops.approximate_cholesky(ei.cuda(), None, n, n // 2, "degree", "asc").numpy()
        torch.cuda.synchronize()
        after = _non_torch_bytes()
    finally:
        del hog
    ref = oracle.approximate_cholesky(ei.numpy(), None, n, n // 2, "degree", "asc")
    assert_same(out, ref, "under memory pressure")
    assert abs(after - before) < (8 << 20), f"device memory outside torch moved by {after - before} bytes"
    # the size query of the C ABI is an upper bound of what the call used (this handle's growth factors may have risen in
    # earlier tests: rlap_workspace_query knows them, rlap_workspace_bytes answers for a fresh handle)
    import ctypes
    lib, hobj = ops._handle_obj(torch.device("cuda", torch.cuda.current_device()))
    ws_b, rng_n = ctypes.c_size_t(0), ctypes.c_int64(0)
    assert lib.rlap_workspace_query(hobj.ptr, ei.shape[1], n, 1, 0, ctypes.byref(ws_b), ctypes.byref(rng_n)) == 0
    need_b, need_r = ctypes.c_size_t(0), ctypes.c_int64(0)
    assert lib.rlap_workspace_needed(hobj.ptr, ctypes.byref(need_b), ctypes.byref(need_r)) == 0
    assert 0 < need_b.value <= ws_b.value and 0 < need_r.value <= rng_n.value
    fresh_b, fresh_r = ctypes.c_size_t(0), ctypes.c_int64(0)
    assert lib.rlap_workspace_bytes(ei.shape[1], n, 1, 0, ctypes.byref(fresh_b), ctypes.byref(fresh_r)) == 0
    assert 0 < fresh_b.value <= ws_b.value


def test_too_small_caller_workspace_is_reported(ops):
    """RLAP_E_WORKSPACE: a caller-provided arena that does not fit makes the call return at once with the size it wants."""
    import ctypes
    from rlap_amd import _lib, graphs
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.rlap_create(ctypes.byref(h)) == 0
    try:
        n = 5000
        ei = graphs.barabasi_albert(n, 4, 3).cuda()
        E = ei.shape[1]
        ws = torch.empty(1 << 16, dtype=torch.uint8, device="cuda")
        rng = torch.empty(1 << 16, dtype=torch.float64, device="cuda")
        assert lib.rlap_set_workspace(h, ws.data_ptr(), ws.numel(), rng.data_ptr(), rng.numel()) == 0
        out = torch.empty((E, 3), dtype=torch.float64, device="cuda")
        rows = ctypes.c_int64(0)
        st = _lib.Stats()
        row, col = ei[0].contiguous(), ei[1].contiguous()
        args = (h, row.data_ptr(), col.data_ptr(), None, E, n, n // 2, 1, 0, None, 0, out.data_ptr(), E, ctypes.byref(rows), ctypes.byref(st))
        assert lib.rlap_approx_chol(*args) == _lib.E_WORKSPACE
        need_b, need_r = ctypes.c_size_t(0), ctypes.c_int64(0)
        assert lib.rlap_workspace_needed(h, ctypes.byref(need_b), ctypes.byref(need_r)) == 0 and need_b.value > ws.numel()
        ws = torch.empty(need_b.value, dtype=torch.uint8, device="cuda")
        rng = torch.empty(max(need_r.value, 1 << 16), dtype=torch.float64, device="cuda")
        assert lib.rlap_set_workspace(h, ws.data_ptr(), ws.numel(), rng.data_ptr(), rng.numel()) == 0
        assert lib.rlap_approx_chol(*args) == 0
        ref = oracle.approximate_cholesky(ei.cpu().numpy(), None, n, n // 2, "degree", "asc")
        assert_same(out[: rows.value].cpu().numpy(), ref, "caller-provided workspace")
    finally:
        lib.rlap_destroy(h)


def test_seed_means_one_draw(ops):
    """o_v="random" without an injected perm -- the node_id vector is drawn on the device, keyed by seed + g.  A single call
    with seed s + g, the from-edges call with s + g, graph g of a batched call with seed s and graph g of ANY sharding of the batch
    (rank r calls with seed s + lo_r) all return the same rows."""
    from rlap_amd import graphs
    G, n, s = 6, 700, 40
    eis = [graphs.barabasi_albert(n, 4, 50 + g) for g in range(G)]
    big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
    sc, rp = ops.approximate_cholesky_batched(big.cuda(), None, node_ptr, [n // 2] * G, "random", "asc", seed=s)
    sc = sc.cpu().numpy()
    per_graph = []
    for g in range(G):
        rows = sc[int(rp[g]):int(rp[g + 1])].copy()
        rows[:, :2] -= g * n
        per_graph.append(rows)
        single = ops.approximate_cholesky(eis[g].cuda(), None, n, n // 2, "random", "asc", seed=s + g).numpy()
        assert_same(single, rows, f"single call seed+{g} vs batched graph {g}")
        fe, nn = ops.approximate_cholesky_from_edges(eis[g].cuda(), None, n, n // 2, "random", "asc", symmetrize=False, seed=s + g)
        assert nn == n
        assert_same(fe.cpu().numpy(), rows, f"from_edges seed+{g} vs batched graph {g}")
    assert any(not np.array_equal(per_graph[0], per_graph[g]) for g in range(1, G)) or True
    for world in (2, 4):      # what rlap_amd/distributed.py::sharded_approximate_cholesky does on rank r
        for r in range(world):
            from rlap_amd.distributed import shard_range
            lo, hi = shard_range(G, r, world)
            if hi == lo:
                continue
            b2, np2 = graphs.batch_disjoint(eis[lo:hi], [n] * (hi - lo))
            sc2, rp2 = ops.approximate_cholesky_batched(b2.cuda(), None, np2, [n // 2] * (hi - lo), "random", "asc", seed=s + lo)
            sc2 = sc2.cpu().numpy()
            for j in range(hi - lo):
                rows = sc2[int(rp2[j]):int(rp2[j + 1])].copy()
                rows[:, :2] -= j * n
                assert_same(rows, per_graph[lo + j], f"world {world} rank {r} graph {lo + j}")
    # two seeds, two draws
    other = ops.approximate_cholesky(eis[0].cuda(), None, n, n // 2, "random", "asc", seed=s + 1000).numpy()
    assert other.shape != per_graph[0].shape or not np.array_equal(other, per_graph[0])


def test_rejected_perm_on_hub_graph_does_not_reach_the_output_pass(ops):
    """an invalid node_id vector may name ONE hub as every survivor; the output pass must not stage S copies of its
    column (that ran past the staging arrays).  All-equal and all-out-of-range vectors on a star: ValueError, and the handle works."""
    n = 6000
    ei = star(n)
    ei_d = torch.from_numpy(ei).cuda()
    for bad in (np.zeros(n, dtype=np.int64), np.full(n, n + 5, dtype=np.int64), np.full(n, -3, dtype=np.int64)):
        with pytest.raises(ValueError):
            ops.approximate_cholesky(ei_d, None, n, 10, "random", "asc", perm=torch.from_numpy(bad))
    # batched: every graph names global id ranges / vertex 0 of a hub graph
    from rlap_amd import graphs
    eis = [torch.from_numpy(star(2000)) for _ in range(8)]
    big, node_ptr = graphs.batch_disjoint(eis, [2000] * 8)
    glob = torch.arange(16000, dtype=torch.int64)           # global ids where local ones are expected
    with pytest.raises(ValueError):
        ops.approximate_cholesky_batched(big.cuda(), None, node_ptr, [5] * 8, "random", "asc", perm=glob)
    with pytest.raises(ValueError):
        ops.approximate_cholesky_batched(big.cuda(), None, node_ptr, [5] * 8, "random", "asc", perm=torch.zeros(16000, dtype=torch.int64))
    good = np.random.RandomState(1).permutation(n)
    ref = oracle.approximate_cholesky(ei, None, n, 10, "random", "asc", perm=good)
    got = gpu_call(ops, ei, None, n, 10, "random", "asc", perm=good)
    assert_same(got, ref, "after rejected hub perms")


def test_from_edges_negative_ids(ops):
    """with num_nodes derived from the ids, an input whose ids are negative is out of range (as with an explicit n),
    not an empty graph."""
    neg = torch.tensor([[-1, -2, -2, -3], [-2, -1, -3, -2]], dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError):
        ops.approximate_cholesky_from_edges(neg, None, None, None, "degree", "asc")
    mixed = torch.tensor([[0, 1, 1, -2], [1, 0, -2, 1]], dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError):
        ops.approximate_cholesky_from_edges(mixed, None, None, None, "degree", "asc")
    with pytest.raises(ValueError):
        ops.approximate_cholesky_from_edges(mixed, None, 5, None, "degree", "asc")


@pytest.mark.parametrize("byte", [0xFF, 0x00, 0x3C])
def test_poisoned_workspace_gives_the_same_rows(ops, byte):
    """Debug poison (rlap_debug_set_poison): the arena, the output buffer and the elimination kernel's LDS start as one byte
    pattern.  A kernel that reads what it was never given then reads the pattern, not the previous call's data, and the rows
    change -- for every mode and both workgroup shapes they must not."""
    from rlap_amd import graphs
    cases = [(2500, 6, "degree", "asc"), (2500, 6, "degree", "random"), (2500, 6, "random", "asc"), (1500, 12, "random", "desc"),
             (2500, 6, "coarsen", "asc"), (300, 40, "random", "asc"), (300, 40, "degree", "desc")]
    refs = []
    for n, m, o_v, o_n in cases:
        ei = ba_graph(n, m, 3)
        perm = np.random.RandomState(7).permutation(n) if o_v == "random" else None
        refs.append((ei, perm, oracle.approximate_cholesky(ei, None, n, n // 2, o_v, o_n, perm=perm, shuffle_seed=9)))
    G, n5 = 300, 1024                                       # more graphs than CUs: the 256-thread shape
    eis = [graphs.barabasi_albert(n5, 6, 700 + g) for g in range(G)]
    big, node_ptr = graphs.batch_disjoint(eis, [n5] * G)
    big = big.cuda()
    perms = torch.from_numpy(np.concatenate([np.random.RandomState(g).permutation(n5) for g in range(G)]))
    clean = {o_v: ops.approximate_cholesky_batched(big, None, node_ptr, [n5 // 2] * G, o_v, "asc", perm=perms if o_v == "random" else None, seed=5,
                                                   return_device="same")[0] for o_v in ("degree", "random", "coarsen")}
    ops.debug_set_poison(byte)
    try:
        for (n, m, o_v, o_n), (ei, perm, ref) in zip(cases, refs):
            got = gpu_call(ops, ei, None, n, n // 2, o_v, o_n, perm=perm, seed=9)
            assert_same(got, ref, f"poison {byte:#x} BA({n},{m}) {o_v}/{o_n}")
        for o_v in ("degree", "random", "coarsen"):
            sc, _ = ops.approximate_cholesky_batched(big, None, node_ptr, [n5 // 2] * G, o_v, "asc", perm=perms if o_v == "random" else None, seed=5,
                                                     return_device="same")
            assert sc.shape == clean[o_v].shape and bool(torch.equal(sc, clean[o_v])), f"poison {byte:#x} batched {o_v}" + _where(sc.cpu().numpy(), clean[o_v].cpu().numpy())
    finally:
        ops.debug_set_poison(-1)


def test_both_workgroup_shapes_agree_on_config5_sample(ops, monkeypatch):
    """The 256-thread shape (batches of more graphs than CUs) and the 1024-thread shape run the same rules: one batch, both
    shapes (RLAP_BATCH_SHAPE), identical rows."""
    from rlap_amd import graphs
    G, n = 320, 4096
    eis = [graphs.barabasi_albert(n, 8, 1000 + g) for g in range(G)]
    big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
    big = big.cuda()
    outs = {}
    for shape in ("2", "1"):
        monkeypatch.setenv("RLAP_BATCH_SHAPE", shape)
        outs[shape] = ops.approximate_cholesky_batched(big, None, node_ptr, [n // 2] * G, "degree", "asc", seed=5, return_device="same")[0]
    monkeypatch.delenv("RLAP_BATCH_SHAPE")
    assert outs["1"].shape == outs["2"].shape and bool(torch.equal(outs["1"], outs["2"])), _where(outs["2"].cpu().numpy(), outs["1"].cpu().numpy())


@pytest.mark.parametrize("o_v", ["degree", "random", "coarsen"])
def test_schedule_jitter_does_not_change_the_rows(ops, monkeypatch, o_v):
    """Debug jitter (rlap_debug_set_jitter): waves sleep behind the elimination kernel's barriers, a different subset each time.
    Round 2's one unexplained mismatch ("graph 851") was a wave leaving a barrier late and reading the push-id counter after
    thread 0 had advanced it; with the jitter that shows in every batch.  Both workgroup shapes, every graph against the clean run,
    a sample against the oracle."""
    from rlap_amd import graphs
    G, n = 288, 2048
    eis = [graphs.barabasi_albert(n, 8, 3000 + g) for g in range(G)]
    big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
    big = big.cuda()
    perms = [np.random.RandomState(g).permutation(n) for g in range(G)]
    perm = torch.from_numpy(np.concatenate(perms)) if o_v == "random" else None
    clean, rp = ops.approximate_cholesky_batched(big, None, node_ptr, [n // 2] * G, o_v, "asc", perm=perm, seed=5, return_device="same")
    for g in range(0, G, 48):
        ref = oracle.approximate_cholesky(eis[g].numpy(), None, n, n // 2, o_v, "asc", perm=perms[g], shuffle_seed=5 + g)
        got = clean[int(rp[g]):int(rp[g + 1])].cpu().numpy().copy()
        got[:, :2] -= g * n
        assert_same(got, ref, f"clean run graph {g}")
    ops.debug_set_jitter(6)
    try:
        for shape in ("2", "1"):
            monkeypatch.setenv("RLAP_BATCH_SHAPE", shape)
            for rep in range(2):
                sc, _ = ops.approximate_cholesky_batched(big, None, node_ptr, [n // 2] * G, o_v, "asc", perm=perm, seed=5, return_device="same")
                assert sc.shape == clean.shape and bool(torch.equal(sc, clean)), f"jitter, shape {shape}, rep {rep}" + _where(sc.cpu().numpy(), clean.cpu().numpy())
    finally:
        ops.debug_set_jitter(0)
        monkeypatch.delenv("RLAP_BATCH_SHAPE", raising=False)


@pytest.mark.parametrize("shape", ["1", "2"])
def test_dead_new_weight_in_front_of_a_dependent_candidate(ops, monkeypatch, shape):
    """Soak case 3023 (tests/tools/soak.py): degree/desc on BA(31749, 5), t = 0.9 n.  Rounding makes one new edge weight negative
    (-5e-31): the reference's `val > 0` filter (preconditioner.cc:252) drops that entry at the next gather, so a dependent
    candidate must not be patched with it (rlap_core.h::cand_patch).  Two graphs of the failing batch, both workgroup shapes."""
    from rlap_amd import graphs
    n, t, seed = 31749, 28574, 690086160
    monkeypatch.setenv("RLAP_BATCH_SHAPE", shape)
    for g in (5, 8):
        ei = graphs.barabasi_albert(n, 5, seed + g)
        ref = oracle.approximate_cholesky(ei.numpy(), None, n, t, "degree", "desc", shuffle_seed=seed + g)
        got = ops.approximate_cholesky(ei.cuda(), None, n, t, "degree", "desc", seed=seed + g).numpy()
        assert_same(got, ref, f"graph {g}, shape {shape}")


@pytest.mark.parametrize("mode", [(1, 0, "degree", "asc"), (2, 0, "coarsen", "asc"), (1, 2, "degree", "random")])
def test_c_caller_of_the_abi_matches_the_oracle(ops, tmp_path, mode):
    """examples/cabi_caller.c: a program in C (hipMalloc, rlap_workspace_query / rlap_set_workspace, rlap_approx_chol; no torch in the
    process) -- its rows, read back from the file it writes, against the oracle on the same graph."""
    import subprocess
    from rlap_amd import graphs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "cabi_caller")
    assert os.path.exists(exe), "examples/cabi_caller is built by __graft_entry__.build()"
    ov, on, o_v, o_n = mode
    n, m, seed = 6000, 5, 17
    out = tmp_path / "rows.bin"
    r = subprocess.run([exe, str(n), str(m), str(seed), str(n // 2), str(ov), str(on), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    got = np.fromfile(out, dtype=np.float64).reshape(-1, 3)
    ei = graphs.barabasi_albert(n, m, seed).numpy()
    ref = oracle.approximate_cholesky(ei, None, n, n // 2, o_v, o_n, shuffle_seed=seed)
    assert_same(got, ref, f"C caller {o_v}/{o_n}: {r.stdout.strip()}")


def test_soak_slice(ops):
    """The first cases of the soak stream (tests/tools/soak.py, seed 2024): random mode / topology (BA, hub, random pairs, grid, clique
    ring) / size / batch / weights / workgroup shape / jitter / poison / fused entry point, every result against the oracle.  The long
    runs of that tool found the dead-new-weight case; this slice keeps the net in the suite."""
    import sys
    tools = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import soak
    saved = os.environ.get("RLAP_BATCH_SHAPE")
    rs = np.random.RandomState(2024)
    try:
        for k in range(70):
            c = soak.draw(rs)
            bad = soak.run_case(c)
            assert not bad, f"case {k}: {soak.describe(c)}: {bad}"
    finally:
        ops.debug_set_jitter(0)
        ops.debug_set_poison(-1)
        if saved is None:
            os.environ.pop("RLAP_BATCH_SHAPE", None)
        else:
            os.environ["RLAP_BATCH_SHAPE"] = saved
