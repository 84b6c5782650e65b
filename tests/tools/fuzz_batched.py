"""Seeded sweep of the batched mode (both workgroup shapes): many random graphs per call, every graph compared with the oracle."""
import sys, time
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
from rlap_amd import graphs, ops
import oracle
from util import ba_graph, grid2d, star

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0; total = 0
t0 = time.time()
for call in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    G = int(rs.choice([3, 40, 300, 700]))
    o_v = str(rs.choice(["degree", "random", "coarsen"])); o_n = str(rs.choice(["asc", "desc"]))
    eis, ns, ts = [], [], []
    for g in range(G):
        kind = int(rs.randint(5))
        if kind == 0: n = int(rs.randint(1, 6)); ei = ba_graph(n, 1, g) if n > 1 else np.zeros((2, 0), dtype=np.int64)
        elif kind == 1: n = int(rs.randint(20, 400)); ei = ba_graph(n, int(rs.randint(1, 8)), g)
        elif kind == 2: a, b = int(rs.randint(2, 15)), int(rs.randint(2, 15)); n = a * b; ei = grid2d(a, b)
        elif kind == 3: n = int(rs.randint(100, 200)); ei = ba_graph(n, int(rs.randint(40, 80)), g)
        else: n = int(rs.randint(50, 600)); ei = star(n)
        eis.append(torch.from_numpy(ei)); ns.append(n); ts.append(int(rs.randint(0, n + 2)))
    big, node_ptr = graphs.batch_disjoint(eis, ns)
    perms = [np.random.RandomState(1000 * call + g).permutation(n) for g, n in enumerate(ns)]
    perm = torch.from_numpy(np.concatenate(perms)) if o_v == "random" else None
    sc, rp = ops.approximate_cholesky_batched(big.cuda(), None, node_ptr, ts, o_v, o_n, perm=perm, seed=5)
    sc = sc.cpu().numpy()
    for g in range(G):
        if o_v == "coarsen":   # keyed order hashes global ids: structure only
            b = sc[int(rp[g]):int(rp[g + 1])]
            fw = set(map(tuple, b[:, :2].astype(int)))
            ok = all((c, r) in fw for r, c in fw)
        else:
            ref = oracle.approximate_cholesky(eis[g].numpy(), None, ns[g], ts[g], o_v, o_n, perm=perms[g], shuffle_seed=5 + g)
            got = sc[int(rp[g]):int(rp[g + 1])].copy(); got[:, :2] -= int(node_ptr[g])
            ok = got.shape == ref.shape and np.array_equal(got, ref)
        total += 1
        if not ok:
            bad += 1; print(f"MISMATCH call {call} graph {g} n={ns[g]} t={ts[g]} {o_v}/{o_n}", flush=True)
print(f"{total} graphs, {bad} mismatches, {time.time()-t0:.0f}s")
