"""Dataflow elimination (RLAP_FLOW=1, rlap_flow.hip) against the oracle, bit for bit, then timings against the round kernel.
usage: flow_check.py [quick|full|time] ..."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
import oracle
from util import *
from rlap_amd import ops, graphs

mode = sys.argv[1] if len(sys.argv) > 1 else "quick"

def call(ei, w, n, t, o_n, perm, seed=3, flow=True):
    os.environ["RLAP_FLOW"] = "1" if flow else "0"
    out = ops.approximate_cholesky(torch.from_numpy(ei).cuda(), None if w is None else torch.from_numpy(w).cuda(), n, t, "random", o_n,
                                   perm=torch.from_numpy(perm), seed=seed)
    return out.numpy(), dict(ops.last_stats)

def hub_graph(n, rng):
    a0 = np.concatenate([np.zeros(n - 2, dtype=np.int64), np.ones(n - 2, dtype=np.int64), np.arange(2, n - 1)])
    b0 = np.concatenate([np.arange(2, n), np.arange(2, n), np.arange(3, n)])
    return symmetrize(a0, b0, n), np.concatenate([[0, 1], 2 + rng.permutation(n - 2)])

bad = 0
if mode in ("quick", "full"):
    cases = [("K4", clique(4), 4), ("K6", clique(6), 6), ("P9", path(9), 9), ("star7", star(7), 7), ("K40", clique(40), 40),
             ("BA100_50", ba_graph(100, 50, 0), 100), ("BA500_3", ba_graph(500, 3, 1), 500), ("BA3000_10", ba_graph(3000, 10, 2), 3000),
             ("BA400_40", ba_graph(400, 40, 5), 400)]
    if mode == "full":
        cases += [("star3000", star(3000), 3000), ("BA20000_5", graphs.barabasi_albert(20000, 5, 3).numpy(), 20000),
                  ("BA1200_200", ba_graph(1200, 200, 4), 1200), ("BA100000_10", graphs.barabasi_albert(100000, 10, 7).numpy(), 100000)]
    rng = np.random.RandomState(7)
    for nm, ei, n in cases:
        perm = rng.permutation(n)
        for o_n in ["asc", "desc", "random"]:
            for wts in (None, sym_weights(ei, n, 5)):
                for t in sorted({n // 2, n - 1} | ({1} if n < 1000 else set())):
                    a = oracle.approximate_cholesky(ei, wts, n, t, "random", o_n, perm=perm, shuffle_seed=3)
                    b, st = call(ei, wts, n, t, o_n, perm)
                    ok = a.shape == b.shape and np.array_equal(a, b)
                    bad += not ok
                    print(nm, o_n, "w" if wts is not None else "u", t, a.shape, b.shape, "OK" if ok else "MISMATCH", "long", st["n_singles"], "retries", st["n_retries"], flush=True)
                    if not ok and a.shape == b.shape:
                        rows = np.nonzero((a != b).any(1))[0]; print("  first bad rows", rows[:3], a[rows[:3]], b[rows[:3]])
    n = 3000
    ei, perm = hub_graph(n, rng)
    a = oracle.approximate_cholesky(ei, None, n, n - 3, "random", "asc", perm=perm, shuffle_seed=3)
    b, st = call(ei, None, n, n - 3, "asc", perm)
    ok = a.shape == b.shape and np.array_equal(a, b); bad += not ok
    print("hubs", a.shape, b.shape, "OK" if ok else "MISMATCH", st["n_singles"], flush=True)
    print("MISMATCHES", bad)
if mode == "time":
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or [(2708, 2), (4096, 8), (100, 50), (169343, 7), (1000000, 10)]
    import hashlib
    for n, m in shapes:
        eid = graphs.barabasi_albert(n, m, 1).cuda()
        pt = torch.from_numpy(np.random.RandomState(0).permutation(n))
        res = {}
        for flow in (False, True):
            os.environ["RLAP_FLOW"] = "1" if flow else "0"
            ops.set_timing(True)
            for _ in range(2):
                out = ops.approximate_cholesky(eid, None, n, n // 2, "random", "asc", perm=pt, return_device="same")
            torch.cuda.synchronize()
            R = 10 if n < 100000 else 3
            t0 = time.perf_counter()
            for _ in range(R):
                out = ops.approximate_cholesky(eid, None, n, n // 2, "random", "asc", perm=pt, return_device="same")
            torch.cuda.synchronize()
            gpu = (time.perf_counter() - t0) / R
            st = dict(ops.last_stats)
            res[flow] = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:10]
            print(f"BA({n},{m}) random/asc flow={int(flow)}: {gpu*1e3:.2f} ms per call (setup {st['ms_setup']:.2f} elim {st['ms_elim']:.2f} output {st['ms_output']:.2f}), long/singles {st['n_singles']}, rows {out.shape[0]}, sha {res[flow]}", flush=True)
        print("  same rows:", res[False] == res[True], flush=True)
sys.exit(1 if bad else 0)
