"""N>1 path on CPU: world_size-2 gloo run of the sharded batched-graph mode
(rlap_amd/distributed.py).  The per-rank compute is injected (the CPU oracle stands in
for the HIP op, which needs a GPU); what is tested is the sharding, the all-gather of
variable-length (m_g,3) blocks and the global row pointer."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_compute(ei, w, node_ptr, nrem, o_v, o_n, seed):
    import oracle
    outs, ptr = [], [0]
    G = node_ptr.numel() - 1
    ei_np = ei.numpy()
    for g in range(G):
        lo, hi = int(node_ptr[g]), int(node_ptr[g + 1])
        sel = (ei_np[1] >= lo) & (ei_np[1] < hi)
        sub = ei_np[:, sel] - lo
        sc = oracle.approximate_cholesky(sub, None, hi - lo, int(nrem[g]), o_v, o_n)
        sc[:, :2] += lo
        outs.append(sc)
        ptr.append(ptr[-1] + sc.shape[0])
    return torch.from_numpy(np.concatenate(outs, 0) if outs else np.zeros((0, 3))), torch.tensor(ptr, dtype=torch.int64)


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rlap_amd.distributed import shard_range, sharded_approximate_cholesky
    from util import ba_graph
    ns = [40, 1, 75, 120, 33]              # 5 graphs over 2 ranks: ragged shards (3 + 2)
    eis = [torch.from_numpy(ba_graph(n, 3, 10 + g)) if n > 3 else torch.zeros((2, 0), dtype=torch.int64) for g, n in enumerate(ns)]
    ts = [n // 2 for n in ns]
    sc, rp = sharded_approximate_cholesky(eis, None, ns, ts, "degree", "asc", compute_fn=_oracle_compute)
    lo, hi = shard_range(len(ns), rank, world)
    ret[rank] = (sc.numpy(), rp.numpy(), (lo, hi))
    dist.destroy_process_group()


def test_sharded_batched_mode_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from util import ba_graph
    ns = [40, 1, 75, 120, 33]
    exp = []
    for g, n in enumerate(ns):
        ei = ba_graph(n, 3, 10 + g) if n > 3 else np.zeros((2, 0), dtype=np.int64)
        exp.append(oracle.approximate_cholesky(ei, None, n, n // 2, "degree", "asc"))
    for rank in range(world):
        sc, rp, _ = ret[rank]
        assert rp.shape[0] == len(ns) + 1
        for g in range(len(ns)):
            got = sc[rp[g]:rp[g + 1]]
            assert got.shape == exp[g].shape and np.array_equal(got, exp[g]), (rank, g)
    assert ret[0][2] == (0, 3) and ret[1][2] == (3, 5)


def _worker_edge_rows(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rlap_amd.distributed import all_gather_edge_rows, all_gather_rows
    rs = np.random.RandomState(100 + rank)
    m = [7, 0, 12][rank]                       # ragged, one rank empty
    sc = torch.from_numpy(np.stack([rs.randint(0, 1 << 30, m).astype(np.float64), rs.randint(0, 1 << 30, m).astype(np.float64),
                                    rs.rand(m)], axis=1).reshape(m, 3))
    a, ca = all_gather_edge_rows(sc)
    b, cb = all_gather_rows(sc)
    ret[rank] = (a.numpy(), ca.numpy(), b.numpy(), cb.numpy())
    dist.destroy_process_group()


def test_packed_edge_row_exchange_world3():
    """bench.py --gpus N exchanges sc_edge_info with the two node ids packed into one word (16 B per row): the
    result equals the plain all-gather bit for bit, with ragged and empty blocks."""
    world = 3
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_edge_rows, args=(world, port, ret), nprocs=world, join=True)
    for rank in range(world):
        a, ca, b, cb = ret[rank]
        assert np.array_equal(ca, cb) and list(ca) == [7, 0, 12]
        assert a.shape == b.shape == (19, 3) and np.array_equal(a, b)
    assert np.array_equal(ret[0][0], ret[2][0])


def test_shard_range_partitions_everything():
    from rlap_amd.distributed import shard_range
    for G in (0, 1, 7, 8, 1024):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(G, r, world)
                cover.extend(range(lo, hi))
            assert cover == list(range(G))


def _worker_few_graphs(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rlap_amd.distributed import shard_range, sharded_approximate_cholesky
    from util import ba_graph
    ns = [30, 44]                              # 2 graphs over 3 ranks: the last rank's shard is empty
    eis = [torch.from_numpy(ba_graph(n, 3, 20 + g)) for g, n in enumerate(ns)]
    sc, rp = sharded_approximate_cholesky(eis, None, ns, [n // 2 for n in ns], "degree", "asc", compute_fn=_oracle_compute)
    ret[rank] = (sc.numpy(), rp.numpy(), shard_range(len(ns), rank, world), str(sc.device))
    dist.destroy_process_group()


def test_fewer_graphs_than_ranks_world3():
    """ranks without graphs still join both collectives with tensors of the right kind."""
    world = 3
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_few_graphs, args=(world, port, ret), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from util import ba_graph
    ns = [30, 44]
    exp = [oracle.approximate_cholesky(ba_graph(n, 3, 20 + g), None, n, n // 2, "degree", "asc") for g, n in enumerate(ns)]
    assert ret[2][2] == (2, 2)                 # empty shard
    for rank in range(world):
        sc, rp, _, devname = ret[rank]
        assert devname == "cpu"
        assert list(rp) == [0, exp[0].shape[0], exp[0].shape[0] + exp[1].shape[0]]
        for g in range(2):
            assert np.array_equal(sc[rp[g]:rp[g + 1]], exp[g]), (rank, g)


def test_empty_shard_uses_the_hip_device_when_the_hip_path_computes(monkeypatch):
    """The empty-shard branch must pick the device from how the NON-empty shards compute (HIP op -> current cuda device),
    not from a variable that was rebound just above it."""
    import rlap_amd.distributed as D
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 3)
    seen = {}
    real_zeros = torch.zeros

    def fake_zeros(*a, **k):
        if "device" in k and isinstance(k["device"], torch.device) and k["device"].type == "cuda":
            seen["device"] = k["device"]
            k = dict(k); k["device"] = "cpu"
        return real_zeros(*a, **k)
    monkeypatch.setattr(torch, "zeros", fake_zeros)
    sc, rp = D.sharded_approximate_cholesky([], None, [], [], "degree", "asc", gather=False)   # G = 0: this rank's shard is empty
    assert seen.get("device") == torch.device("cuda", 3) and sc.shape == (0, 3) and list(rp) == [0]


def test_bench_spawn_plan():
    """bench.py --gpus N without a torchrun environment: N ranks, one per GPU, rendezvous on 127.0.0.1; under
    torchrun (WORLD_SIZE set) or with --gpus 1 the process is a rank itself."""
    sys.path.insert(0, ROOT)
    import bench
    argv = ["--gpus", "4", "--steps", "2", "--warmup", "1", "--workload", "c5"]
    args = bench.parse_args(argv)
    plan = bench.spawn_plan(args, argv, {"PATH": "/usr/bin"})
    assert plan is not None and len(plan) == 4
    for r, (cmd, env) in enumerate(plan):
        assert cmd[1].endswith("bench.py") and cmd[2:] == argv
        assert env["RANK"] == env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert len({env["MASTER_PORT"] for _, env in plan}) == 1
    assert bench.spawn_plan(args, argv, {"WORLD_SIZE": "4", "RANK": "0"}) is None
    assert bench.spawn_plan(bench.parse_args(["--gpus", "1"]), ["--gpus", "1"], {}) is None
    from rlap_amd.distributed import shard_range
    assert [shard_range(bench.C5_GRAPHS, r, 8) for r in (0, 7)] == [(0, 128), (896, 1024)]
