"""Phase profile of the dataflow elimination (RLAP_PHASE_PROFILE=1) on one BA graph: argv n m [waves]."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
os.environ["RLAP_FLOW"] = "1"; os.environ["RLAP_PHASE_PROFILE"] = "1"
import numpy as np, torch
from rlap_amd import graphs, ops
n, m = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3: os.environ["RLAP_FLOW_WAVES"] = sys.argv[3]
eid = graphs.barabasi_albert(n, m, 1).cuda()
pt = torch.from_numpy(np.random.RandomState(0).permutation(n))
ops.set_timing(True)
for _ in range(1):
    ops.approximate_cholesky(eid, None, n, n // 2, "random", "asc", perm=pt, return_device="same")
os.environ["RLAP_PHASE_PROFILE"] = "1"
ops.approximate_cholesky(eid, None, n, n // 2, "random", "asc", perm=pt, return_device="same")
os.environ["RLAP_PHASE_PROFILE"] = "0"
print(dict(ops.last_stats), flush=True)
