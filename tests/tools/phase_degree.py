"""Phase profile (RLAP_PHASE_PROFILE=1) of the round kernel on BA(1M,10), degree/asc: where a round's 58 us go."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RLAP_PHASE_PROFILE"] = "1"
import numpy as np, torch
from rlap_amd import graphs, ops
n, m = 1000000, 10
ei = graphs.barabasi_albert(n, m, 2).cuda()
ops.set_timing(True)
for _ in range(2):
    ops.approximate_cholesky(ei, None, n, n // 2, "degree", "asc", return_device="same")
print(dict(ops.last_stats))
