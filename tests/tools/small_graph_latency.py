"""Latency of one small call (BASELINE config 2 stand-in: BA(2708, m=2), t=N/2, o_v=random) against the CPU port."""
import sys, time
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
from rlap_amd import graphs, ops
import oracle
for n, m in ((2708, 2), (100, 50), (4096, 8), (19717, 2)):
    ei = graphs.barabasi_albert(n, m, 1)
    eid = ei.cuda()
    perm = np.random.RandomState(0).permutation(n)
    pt = torch.from_numpy(perm)
    for o_v in ("random", "degree"):
        ops.set_timing(False)
        for _ in range(3):
            ops.approximate_cholesky(eid, None, n, n // 2, o_v, "asc", perm=pt, return_device="same")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 20
        for _ in range(R):
            out = ops.approximate_cholesky(eid, None, n, n // 2, o_v, "asc", perm=pt, return_device="same")
        torch.cuda.synchronize()
        gpu = (time.perf_counter() - t0) / R
        ops.set_timing(True)
        ops.approximate_cholesky(eid, None, n, n // 2, o_v, "asc", perm=pt, return_device="same")
        st = dict(ops.last_stats)
        t0 = time.perf_counter()
        for _ in range(5):
            ref = oracle.approximate_cholesky(ei.numpy(), None, n, n // 2, o_v, "asc", perm=perm)
        cpu = (time.perf_counter() - t0) / 5
        ok = np.array_equal(out.cpu().numpy(), ref)
        print(f"BA({n},{m}) {o_v}/asc: GPU {gpu*1e3:.2f} ms per call (setup {st['ms_setup']:.2f} / elim {st['ms_elim']:.2f} / output {st['ms_output']:.2f}), CPU port {cpu*1e3:.2f} ms, bit-exact={ok}; rounds {st.get('n_rounds')} singles {st.get('n_singles')}")
