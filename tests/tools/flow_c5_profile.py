"""Phase profile of the dataflow kernel on a config-5-like batch: argv G [shape] [waves]."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
os.environ["RLAP_FLOW"] = "1"; os.environ["RLAP_PHASE_PROFILE"] = "1"
import numpy as np, torch
from rlap_amd import graphs, ops
G = int(sys.argv[1]); n, m = 4096, 8
if len(sys.argv) > 2: os.environ["RLAP_FLOW_SHAPE"] = sys.argv[2]
if len(sys.argv) > 3: os.environ["RLAP_FLOW_WAVES"] = sys.argv[3]
eis = [graphs.barabasi_albert(n, m, 1000 + g) for g in range(G)]
big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
big = big.cuda()
perm = torch.from_numpy(np.concatenate([np.random.RandomState(g).permutation(n) for g in range(G)]))
ops.set_timing(True)
for _ in range(2):
    sc, rp = ops.approximate_cholesky_batched(big, None, node_ptr, [n // 2] * G, "random", "asc", perm=perm, seed=5, return_device="same")
print(dict(ops.last_stats), flush=True)
