"""Soak run at BASELINE-config sizes (1e5 .. 1e6 vertices, one graph per call): random mode / size / density / weights / hub,
the HIP result against the CPU oracle bit for bit.  The oracle (3-7 s a case) runs in worker threads beside the GPU calls
(ctypes releases the GIL), so the GPU is not idle while the checker works.
usage: soak_large.py SECONDS [SEED [THREADS]]"""
import hashlib
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import torch

import oracle
from rlap_amd import graphs, ops
from util import sym_weights, symmetrize

O_V = ["degree", "random", "coarsen"]
O_N = ["asc", "desc", "random"]


def draw(rs):
    c = {"o_v": O_V[rs.randint(3)], "o_n": O_N[rs.randint(3)]}
    c["n"] = int(10 ** rs.uniform(5.0, 6.0))
    c["m"] = int(rs.randint(2, 13))
    c["t"] = int(float(rs.choice([0.1, 0.5, 0.5, 0.9, 1.0])) * c["n"])
    c["weighted"] = bool(rs.rand() < 0.3)
    c["hub"] = bool(rs.rand() < 0.15)
    c["jitter"] = int(rs.choice([0, 0, 3]))
    c["poison"] = int(rs.choice([-1, -1, 90]))
    c["seed"] = int(rs.randint(1 << 30))
    return c


def describe(c):
    return " ".join(f"{k}={v}" for k, v in c.items())


def digest(a):
    return a.shape, hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def check(c, ei, w, perm):
    ref = oracle.approximate_cholesky(ei, w, c["n"], c["t"], c["o_v"], c["o_n"], perm=perm, shuffle_seed=c["seed"])
    return digest(ref)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    t_end = time.time() + budget
    pool = ThreadPoolExecutor(max_workers=threads)
    pending = []
    n_ok = 0

    def settle(block):
        nonlocal n_ok
        while pending and (block or pending[0][2].done() or len(pending) > 2 * threads):
            c, got, fut = pending.pop(0)
            ref = fut.result()
            if got != ref:
                print("MISMATCH", describe(c), got, ref, flush=True)
                sys.exit(1)
            n_ok += 1
            if n_ok % 10 == 0:
                print(f"[{n_ok} large cases bit-exact] last: {describe(c)}", flush=True)

    while time.time() < t_end:
        c = draw(rs)
        n = c["n"]
        ei = graphs.barabasi_albert(n, c["m"], c["seed"]).numpy()
        if c["hub"]:
            r2 = np.random.RandomState(c["seed"] % (1 << 31))
            h = int(r2.randint(n))
            others = np.flatnonzero(r2.rand(n) < 0.02)
            others = others[others != h].astype(np.int64)
            ei = symmetrize(np.concatenate([ei[0], np.full(others.size, h, dtype=np.int64)]), np.concatenate([ei[1], others]), n)
        w = sym_weights(ei, n, c["seed"] + 7) if c["weighted"] else None
        perm = np.random.RandomState(c["seed"] % (1 << 31)).permutation(n) if c["o_v"] == "random" else None
        fut = pool.submit(check, c, ei, w, perm)
        ops.debug_set_jitter(c["jitter"])
        ops.debug_set_poison(c["poison"])
        got = ops.approximate_cholesky(torch.from_numpy(ei).cuda(), None if w is None else torch.from_numpy(w).cuda(), n, c["t"], c["o_v"], c["o_n"],
                                       perm=None if perm is None else torch.from_numpy(perm), seed=c["seed"]).numpy()
        pending.append((c, digest(got), fut))
        settle(False)
    settle(True)
    ops.debug_set_jitter(0)
    ops.debug_set_poison(-1)
    print(f"large soak ok: {n_ok} cases bit-exact against the oracle in {budget:.0f} s")


if __name__ == "__main__":
    main()
