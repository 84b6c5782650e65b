"""Diagnostic: batched random-order run under schedule jitter, one workgroup shape (argv: shape jitter G n)."""
import os, sys
shape, jit, G, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
os.environ["RLAP_BATCH_SHAPE"] = shape
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rlap_amd import graphs, ops
eis = [graphs.barabasi_albert(n, 8, 3000 + g) for g in range(G)]
big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
big = big.cuda()
perm = torch.from_numpy(np.concatenate([np.random.RandomState(g).permutation(n) for g in range(G)]))
clean, rp = ops.approximate_cholesky_batched(big, None, node_ptr, [n // 2] * G, "random", "asc", perm=perm, seed=5, return_device="same")
print("clean ok", ops.last_stats["n_rounds"], ops.last_stats["n_singles"], flush=True)
ops.debug_set_jitter(jit)
sc, _ = ops.approximate_cholesky_batched(big, None, node_ptr, [n // 2] * G, "random", "asc", perm=perm, seed=5, return_device="same")
print("jitter run done, equal:", bool(sc.shape == clean.shape and torch.equal(sc, clean)), ops.last_stats["n_rounds"], ops.last_stats["n_singles"], flush=True)
