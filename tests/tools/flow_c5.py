"""Config-5-like batches through the round kernel and the dataflow kernel (both shapes): time, and the same rows.
usage: flow_c5.py [G ...]"""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
import numpy as np, torch
from rlap_amd import graphs, ops
n, m = 4096, 8
for G in [int(a) for a in sys.argv[1:]] or [128, 1024]:
    eis = [graphs.barabasi_albert(n, m, 1000 + g) for g in range(G)]
    big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
    big = big.cuda()
    ts = [n // 2] * G
    perm = torch.from_numpy(np.concatenate([np.random.RandomState(g).permutation(n) for g in range(G)]))
    ops.set_timing(True)
    ref = None
    for flow, shape, waves in ((0, "", ""), (1, "2", ""), (1, "1", ""), (1, "3", "")):
        os.environ["RLAP_FLOW"] = str(flow)
        for k, v in (("RLAP_FLOW_SHAPE", shape), ("RLAP_FLOW_WAVES", waves)):
            if v: os.environ[k] = v
            else: os.environ.pop(k, None)
        for _ in range(2):
            sc, rp = ops.approximate_cholesky_batched(big, None, node_ptr, ts, "random", "asc", perm=perm, seed=5, return_device="same")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 5
        for _ in range(R):
            sc, rp = ops.approximate_cholesky_batched(big, None, node_ptr, ts, "random", "asc", perm=perm, seed=5, return_device="same")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / R
        st = ops.last_stats
        same = True if ref is None else bool(sc.shape == ref.shape and torch.equal(sc, ref))
        if ref is None: ref = sc.clone()
        print(f"G={G} flow={flow} shape={shape or '-'} waves={waves or '-'}: {dt*1e3:.2f} ms (setup {st['ms_setup']:.2f} elim {st['ms_elim']:.2f} output {st['ms_output']:.2f}) same rows: {same}", flush=True)
