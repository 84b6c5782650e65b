"""Oracle parity at BASELINE-config sizes for the non-headline modes (run on the GPU box)."""
import sys, time
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
from rlap_amd import graphs, ops
import oracle
ops.set_timing(True)
cases = [(169_343, 7, "coarsen", "asc", 4), (1_000_000, 10, "random", "asc", 2), (1_000_000, 10, "degree", "desc", 2), (300_000, 10, "degree", "random", 5)]
for n, m, o_v, o_n, seed in cases:
    ei = graphs.barabasi_albert(n, m, seed)
    perm = np.random.RandomState(1).permutation(n) if o_v == "random" else None
    for weighted in (False, True):
        w = None
        if weighted:
            r, c = ei.numpy()
            und = np.minimum(r, c) * n + np.maximum(r, c)
            uq, inv = np.unique(und, return_inverse=True)
            w = np.random.RandomState(3).uniform(0.5, 1.5, uq.shape[0])[inv]
        ops.approximate_cholesky(ei.cuda(), None if w is None else torch.from_numpy(w).cuda(), n, n // 2, o_v, o_n, perm=None if perm is None else torch.from_numpy(perm), seed=9)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got = ops.approximate_cholesky(ei.cuda(), None if w is None else torch.from_numpy(w).cuda(), n, n // 2, o_v, o_n, perm=None if perm is None else torch.from_numpy(perm), seed=9, return_device="same")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = dict(ops.last_stats)
        t1 = time.perf_counter()
        ref = oracle.approximate_cholesky(ei.numpy(), w, n, n // 2, o_v, o_n, perm=perm, shuffle_seed=9)
        cpu = time.perf_counter() - t1
        got = got.cpu().numpy()
        ok = got.shape == ref.shape and np.array_equal(got, ref)
        print(f"BA({n},{m}) {o_v}/{o_n} weighted={weighted}: rows={ref.shape[0]} bit-exact={ok}  gpu {dt*1e3:.0f} ms (elim {st['ms_elim']:.0f}, out {st['ms_output']:.0f})  cpu-oracle {cpu:.2f} s", flush=True)
