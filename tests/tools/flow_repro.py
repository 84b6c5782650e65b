"""One dataflow-elimination call against the oracle (argv: n m t o_n [weighted] [flow]); for debugger runs."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
import oracle
from util import ba_graph, sym_weights
from rlap_amd import ops
n, m, t, o_n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
weighted = len(sys.argv) > 5 and sys.argv[5] == "1"
os.environ["RLAP_FLOW"] = sys.argv[6] if len(sys.argv) > 6 else "1"
ei = ba_graph(n, m, 2)
w = sym_weights(ei, n, 5) if weighted else None
perm = np.random.RandomState(11).permutation(n)
b = ops.approximate_cholesky(torch.from_numpy(ei).cuda(), None if w is None else torch.from_numpy(w).cuda(), n, t, "random", o_n, perm=torch.from_numpy(perm), seed=3).numpy()
print("gpu done", b.shape, dict(ops.last_stats), flush=True)
a = oracle.approximate_cholesky(ei, w, n, t, "random", o_n, perm=perm, shuffle_seed=3)
print("equal:", a.shape == b.shape and np.array_equal(a, b), flush=True)
