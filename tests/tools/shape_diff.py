"""Diagnostic: one batch of BA(4096,8) graphs in both workgroup shapes (RLAP_BATCH_SHAPE), every graph against the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from rlap_amd import graphs, ops

G = int(sys.argv[1]) if len(sys.argv) > 1 else 320
n = 4096
o_v = sys.argv[2] if len(sys.argv) > 2 else "degree"
eis = [graphs.barabasi_albert(n, 8, 1000 + g) for g in range(G)]
big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
big = big.cuda()
refs = [oracle.approximate_cholesky(eis[g].numpy(), None, n, n // 2, o_v, "asc") for g in range(G)]
for rep in range(3):
    for shape in ("2", "1"):
        os.environ["RLAP_BATCH_SHAPE"] = shape
        sc, rp = ops.approximate_cholesky_batched(big, None, node_ptr, [n // 2] * G, o_v, "asc", seed=5)
        sc = sc.cpu().numpy()
        bad = []
        for g in range(G):
            got = sc[int(rp[g]):int(rp[g + 1])].copy(); got[:, :2] -= g * n
            if got.shape != refs[g].shape or not np.array_equal(got, refs[g]):
                d = np.flatnonzero((got != refs[g]).any(axis=1)) if got.shape == refs[g].shape else []
                bad.append((g, got.shape[0], refs[g].shape[0], len(d)))
        print(f"rep {rep} shape {'256' if shape == '2' else '1024'}: {len(bad)} graphs differ from the oracle {bad[:10]} rounds={ops.last_stats['n_rounds']} singles={ops.last_stats['n_singles']}", flush=True)
# the first bad graph alone, 1024-thread shape
os.environ["RLAP_BATCH_SHAPE"] = "1"
for g in (32,):
    for rep in range(3):
        out = ops.approximate_cholesky(eis[g].cuda(), None, n, n // 2, o_v, "asc").numpy()
        print(f"graph {g} alone (1024): equal={out.shape == refs[g].shape and np.array_equal(out, refs[g])} rounds={ops.last_stats['n_rounds']}", flush=True)
