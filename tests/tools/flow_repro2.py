"""The BA(3000,10) t=2999 case: first with the single-wave shape (the growth retries happen there), then with the helper shape."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
import oracle
from util import ba_graph
from rlap_amd import ops
os.environ["RLAP_FLOW"] = "1"
n, t = 3000, 2999
ei = ba_graph(n, 10, 2)
perm = np.random.RandomState(11).permutation(n)
a = oracle.approximate_cholesky(ei, None, n, t, "random", "asc", perm=perm, shuffle_seed=3)
for shape in ("4", "1", "1"):
    os.environ["RLAP_FLOW_SHAPE"] = shape
    b = ops.approximate_cholesky(torch.from_numpy(ei).cuda(), None, n, t, "random", "asc", perm=torch.from_numpy(perm), seed=3).numpy()
    print("shape", shape, "retries", ops.last_stats["n_retries"], "long", ops.last_stats["n_singles"], "equal:", a.shape == b.shape and np.array_equal(a, b), flush=True)
