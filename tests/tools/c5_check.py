"""Config 5 (1024 x BA(4096,8)) through one batched call; every graph (or the listed ones) against the oracle.
usage: c5_check.py o_v [graph ids...]"""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np, torch
from rlap_amd import graphs, ops
import oracle
o_v = sys.argv[1]
G, n, m = 1024, 4096, 8
eis = [graphs.barabasi_albert(n, m, 1000 + g) for g in range(G)]
big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
perms = [np.random.RandomState(g).permutation(n) for g in range(G)]
perm = torch.from_numpy(np.concatenate(perms)) if o_v == "random" else None
which = [int(x) for x in sys.argv[2:]] or list(range(G))
for rep in range(2):
    sc, rp = ops.approximate_cholesky_batched(big.cuda(), None, node_ptr, [n // 2] * G, o_v, "asc", perm=perm, seed=5)
    sc = sc.cpu().numpy()
    bad = []
    for g in which:
        ref = oracle.approximate_cholesky(eis[g].numpy(), None, n, n // 2, o_v, "asc", perm=perms[g], shuffle_seed=5 + g)
        got = sc[int(rp[g]):int(rp[g + 1])].copy(); got[:, :2] -= g * n
        if got.shape != ref.shape or not np.array_equal(got, ref): bad.append(g)
    print(f"{o_v}/asc rep {rep}: {len(which)} graphs checked, mismatching: {bad}", flush=True)
