"""CPU soak run: the case stream of soak.py, run through the host mirror of the frontier kernel's round rules
(tests/csrc/host_mirror.cc, the code of rlap_core.h compiled with g++) instead of the GPU, against the oracle: finds errors in
the batching RULES (what may share a round, patches, pre-emption, push order) without a GPU.  Runs as many worker processes as
asked, worker k takes the cases with number = k mod workers.
usage: soak_mirror.py SECONDS [SEED [WORKERS [MAX_NODES]]]"""
import ctypes
import multiprocessing as mp
import os
import sys
import time

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
sys.path.insert(0, os.path.join(_ROOT, "tests", "tools"))
import numpy as np


def mirror_batch(lib, ei, w, n, t, o_v, o_n, B, perm, seed, bc):
    """tests/test_core_mirror.py::_mirror_batch with the retry the C ABI makes: four times the pool when it overflows."""
    import oracle
    E = ei.shape[1]
    row, col = np.ascontiguousarray(ei[0]), np.ascontiguousarray(ei[1])
    w = np.ones(E) if w is None else np.ascontiguousarray(w, dtype=np.float64)
    p = np.ascontiguousarray(perm, dtype=np.int64) if perm is not None else None
    pool = 4 * E + 64
    while True:
        out = ctypes.POINTER(ctypes.c_double)()
        rows = ctypes.c_int64()
        order = np.full(max(n, 1), -1, dtype=np.int64)
        stats = np.zeros(24, dtype=np.int64)
        rc = lib.mirror_approx_chol_batch_bc(
            ctypes.c_void_p(row.ctypes.data), ctypes.c_void_p(col.ctypes.data), ctypes.c_void_p(w.ctypes.data),
            ctypes.c_int64(E), ctypes.c_int64(n), ctypes.c_int64(t), oracle.O_V[o_v], oracle.O_N[o_n],
            ctypes.c_void_p(p.ctypes.data) if p is not None else None, ctypes.c_uint64(seed), ctypes.c_int32(pool),
            ctypes.c_int32(B), ctypes.c_int32(bc), ctypes.byref(out), ctypes.byref(rows), ctypes.c_void_p(order.ctypes.data),
            ctypes.c_void_p(stats.ctypes.data))
        if rc in (4, 6) and pool < (1 << 29):    # ST_POOL_OVERFLOW / ST_RNG_OVERFLOW
            pool *= 4
            continue
        if rc:
            raise RuntimeError(f"mirror status {rc}")
        m = rows.value
        res = np.ctypeslib.as_array(out, shape=(max(m, 1) * 3,))[: 3 * m].copy().reshape(m, 3)
        lib.mirror_free(out)
        return res, order[:n], stats


def worker(k, workers, budget, seed, max_nodes, q):
    try:
        worker_body(k, workers, budget, seed, max_nodes, q)
    except BaseException as e:      # (a worker that dies silently would leave the parent waiting)
        q.put(("ERROR", k, repr(e)))


def worker_body(k, workers, budget, seed, max_nodes, q):
    import oracle
    import soak
    from util import sym_weights
    lib = ctypes.CDLL(os.path.join(_ROOT, "tests", "csrc", "libhost_mirror.so"))
    lib.mirror_approx_chol_batch_bc.restype = ctypes.c_int
    rs = np.random.RandomState(seed)
    t_end = time.time() + budget
    case = -1
    n_run = 0
    while time.time() < t_end:
        c = soak.draw(rs)
        case += 1
        q.put(("at", k, case)) if case % workers == k and os.environ.get("SOAK_VERBOSE") else None
        if case % workers != k or c["n"] > max_nodes:
            continue
        n, t, o_v, o_n = c["n"], c["t"], c["o_v"], c["o_n"]
        for g in c["check"][:2]:
            ei = soak.make_graph(c, g)
            w = sym_weights(ei, n, c["seed"] + 7 * g) if c["weighted"] else None
            perm = np.random.RandomState(c["seed"] + g).permutation(n) if o_v == "random" else None
            ref, oref = oracle.approximate_cholesky(ei, w, n, t, o_v, o_n, perm=perm, shuffle_seed=c["seed"] + g, return_order=True)
            shapes = ((4, 32), (32, 32), (128, 32)) if o_v != "random" else ((16, 64), (64, 64), (32, 128))
            for B, bc in shapes:
                got, ogot, st = mirror_batch(lib, ei, w, n, t, o_v, o_n, B, perm, c["seed"] + g, bc)
                if got.shape != ref.shape or not np.array_equal(got, ref) or not np.array_equal(oref, ogot):
                    q.put(("MISMATCH", case, soak.describe(c), g, B, bc, got.shape, ref.shape))
                    return
        n_run += 1
    q.put(("ok", k, n_run, case))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    max_nodes = int(sys.argv[4]) if len(sys.argv) > 4 else 40000
    q = mp.Queue()
    ps = [mp.Process(target=worker, args=(k, workers, budget, seed, max_nodes, q)) for k in range(workers)]
    for p in ps:
        p.start()
    bad = 0
    total = 0
    finished = 0
    while finished < len(ps):
        r = q.get()
        print(r, flush=True)
        if r[0] == "at":
            continue
        finished += 1
        if r[0] != "ok":
            bad += 1
        else:
            total += r[2]
    for p in ps:
        p.join()
    print(f"mirror soak: {total} cases, {bad} workers stopped at a mismatch")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
