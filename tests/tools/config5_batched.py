"""BASELINE config 5: batch of 1024 BA(4096, m=8) graphs, num_remove = N/2 each, one launch.
Checks a sample of graphs against the oracle and reports throughput."""
import sys, time
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
from rlap_amd import graphs, ops
import oracle
G, n, m = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024, 4096, 8
CPU_PROCS = int(sys.argv[sys.argv.index("--cpu-procs") + 1]) if "--cpu-procs" in sys.argv else 0
eis = [graphs.barabasi_albert(n, m, 1000 + g) for g in range(G)]
big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
big = big.cuda()
ts = [n // 2] * G
ops.set_timing(True)
for o_v in ("degree", "random"):
    perms = [np.random.RandomState(g).permutation(n) for g in range(G)]
    perm = torch.from_numpy(np.concatenate(perms)) if o_v == "random" else None
    sc, rp = ops.approximate_cholesky_batched(big, None, node_ptr, ts, o_v, "asc", perm=perm, seed=5)   # warm-up (allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc, rp = ops.approximate_cholesky_batched(big, None, node_ptr, ts, o_v, "asc", perm=perm, seed=5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = ops.last_stats
    print(f"{o_v}/asc: G={G} graphs x {n} nodes: {dt*1e3:.1f} ms  -> {G*(n//2)/dt:.3e} eliminated vertices/s, {int(rp[-1])/dt:.3e} output edges/s; phases {st['ms_setup']:.1f}/{st['ms_elim']:.1f}/{st['ms_output']:.1f} ms")
    sc = sc.cpu().numpy()
    bad = 0
    for g in list(range(0, G, max(1, G // 16))):
        ref = oracle.approximate_cholesky(eis[g].numpy(), None, n, n // 2, o_v, "asc", perm=perms[g])
        got = sc[int(rp[g]):int(rp[g + 1])].copy(); got[:, :2] -= g * n
        bad += not (got.shape == ref.shape and np.array_equal(got, ref))
    print("  oracle check on 16 graphs: mismatches =", bad)


# the fair multi-core CPU comparison -- one graph per process on the host cores (oracle = CPU port)
def _cpu_one(g):
    import numpy as _np
    ei = graphs.barabasi_albert(n, m, 1000 + g).numpy()
    pm = _np.random.RandomState(g).permutation(n)
    t0 = time.perf_counter()
    oracle.approximate_cholesky(ei, None, n, n // 2, "random", "asc", perm=pm)
    return time.perf_counter() - t0


if CPU_PROCS > 0:
    import multiprocessing as mp
    sample = 8 * CPU_PROCS
    with mp.get_context("fork").Pool(CPU_PROCS) as pool:
        t0 = time.perf_counter()
        per = pool.map(_cpu_one, range(sample))
        wall = time.perf_counter() - t0
    print(f"CPU port, random/asc, {CPU_PROCS} processes x 1 graph each: {sample} graphs in {wall*1e3:.0f} ms (incl. graph generation) -> "
          f"{sample*(n//2)/wall:.3e} eliminated vertices/s; per-graph oracle call {1e3*sum(per)/len(per):.1f} ms on one core")
