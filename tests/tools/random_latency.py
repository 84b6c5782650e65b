"""o_v = random: time per call, rounds and single-vertex fallbacks for a few graph shapes (A/B of kernel variants with RLAP_AMD_LIB).
usage: random_latency.py [N,m ...]"""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
import numpy as np, torch
from rlap_amd import graphs, ops
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(2708, 2), (19717, 2), (4096, 8), (100, 50), (169343, 7), (1000000, 10)]
for n, m in shapes:
    eid = graphs.barabasi_albert(n, m, 1).cuda()
    pt = torch.from_numpy(np.random.RandomState(0).permutation(n))
    ops.set_timing(False)
    for _ in range(2):
        ops.approximate_cholesky(eid, None, n, n // 2, "random", "asc", perm=pt, return_device="same")
    torch.cuda.synchronize()
    R = 10 if n < 100000 else 2
    t0 = time.perf_counter()
    for _ in range(R):
        out = ops.approximate_cholesky(eid, None, n, n // 2, "random", "asc", perm=pt, return_device="same")
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / R
    st = dict(ops.last_stats)
    import hashlib
    print(f"BA({n},{m}) random/asc: {gpu*1e3:.2f} ms per call, rounds {st['n_rounds']}, singles {st['n_singles']}, rows {out.shape[0]}, sha {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:10]}", flush=True)
