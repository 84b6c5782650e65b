"""Per-phase clock stamps of the elimination kernel for one small graph (run with RLAP_PHASE_PROFILE=1 on the GPU box).
usage: phase_small.py N m o_v [o_n]"""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
import numpy as np, torch
from rlap_amd import graphs, ops
n, m, o_v = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
o_n = sys.argv[4] if len(sys.argv) > 4 else "asc"
ei = graphs.barabasi_albert(n, m, 1).cuda()
perm = torch.from_numpy(np.random.RandomState(0).permutation(n))
ops.set_timing(True)
for _ in range(2):
    ops.approximate_cholesky(ei, None, n, n // 2, o_v, o_n, perm=perm, return_device="same")
print(ops.last_stats)
