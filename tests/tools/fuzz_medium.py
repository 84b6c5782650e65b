"""Seeded sweep over medium graphs (BA with various m, grids, random regular-ish) x all modes x weights against the oracle.
Run on the GPU box; prints a line per failure and a summary."""
import sys, time
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
from rlap_amd import ops
import oracle
from util import ba_graph, grid2d, sym_weights

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
t0 = time.time()
for trial in range(N):
    kind = int(rs.randint(4))
    if kind == 0:
        n = int(rs.choice([500, 2000, 8000, 20000])); m = int(rs.choice([1, 2, 5, 12, 30])); ei = ba_graph(n, min(m, n - 1), int(rs.randint(1 << 20)))
    elif kind == 1:
        a, b = int(rs.randint(5, 80)), int(rs.randint(5, 80)); n = a * b; ei = grid2d(a, b)
    elif kind == 2:
        n = int(rs.choice([300, 1000])); m = int(rs.choice([60, 150])); ei = ba_graph(n, m, int(rs.randint(1 << 20)))   # dense: long columns
    else:
        n = int(rs.choice([3000, 10000])); ei = ba_graph(n, 3, int(rs.randint(1 << 20)))
    o_v = str(rs.choice(["degree", "random", "coarsen"])); o_n = str(rs.choice(["asc", "desc", "random"]))
    t = int(rs.choice([n // 4, n // 2, (3 * n) // 4, n - 1]))
    wk = int(rs.randint(3))
    w = None if wk == 0 else sym_weights(ei, n, int(rs.randint(1 << 30)))
    if wk == 1:
        w = np.round(w * 2) / 2 + 0.5
    perm = rs.permutation(n) if o_v == "random" else None
    seed = int(rs.randint(1 << 30))
    ref = oracle.approximate_cholesky(ei, w, n, t, o_v, o_n, perm=perm, shuffle_seed=seed)
    got = ops.approximate_cholesky(torch.from_numpy(ei).cuda(), None if w is None else torch.from_numpy(w).cuda(), n, t, o_v, o_n,
                                   perm=None if perm is None else torch.from_numpy(perm), seed=seed).numpy()
    ok = got.shape == ref.shape and np.array_equal(got, ref)
    if not ok:
        bad += 1
        print(f"MISMATCH trial {trial}: kind={kind} n={n} {o_v}/{o_n} t={t} w={wk} rows {got.shape} vs {ref.shape}", flush=True)
print(f"{N} cases, {bad} mismatches, {time.time()-t0:.0f}s")
