"""Schedule model of the dataflow design for o_v = "random": how long would the elimination take if
every position of the (known) order were taken by the next free wave, gathered once its earlier neighbours have committed, and
sampled in order of the uniform stream?  Runs the CPU mirror's sequential elimination with an event simulation beside it
(tests/csrc/host_mirror.cc::mirror_flow_model).  Costs per vertex (us) are taken from what one wave of today's kernel needs:
prepare = P0 + P1*len, commit = C0 + C1*len, S per link of the offset chain.
usage: flow_model.py [c3|c5|cora|<n> <m>]"""
import ctypes
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np

from rlap_amd import graphs

lib = ctypes.CDLL(os.path.join(_ROOT, "tests", "csrc", "libhost_mirror.so"))
lib.mirror_flow_model.restype = ctypes.c_int


def model(n, m, seed=0, cost=(8.0, 0.25, 5.0, 0.15, 0.05), workers=(16, 64, 256, 1024, 4096)):
    ei = graphs.barabasi_albert(n, m, seed).numpy()
    E = ei.shape[1]
    row, col = np.ascontiguousarray(ei[0]), np.ascontiguousarray(ei[1])
    w = np.ones(E)
    perm = np.ascontiguousarray(np.random.RandomState(seed).permutation(n).astype(np.int64))
    wk = np.asarray(workers, dtype=np.int32)
    c = np.asarray(cost, dtype=np.float64)
    out = np.zeros(len(workers) + 4)
    pool = 4 * E + 64
    while True:
        rc = lib.mirror_flow_model(ctypes.c_void_p(row.ctypes.data), ctypes.c_void_p(col.ctypes.data), ctypes.c_void_p(w.ctypes.data),
                                   ctypes.c_int64(E), ctypes.c_int64(n), ctypes.c_int64(n // 2), 0, ctypes.c_void_p(perm.ctypes.data),
                                   ctypes.c_uint64(seed), ctypes.c_int32(pool), ctypes.c_int32(len(workers)), ctypes.c_void_p(wk.ctypes.data),
                                   ctypes.c_void_p(c.ctypes.data), ctypes.c_void_p(out.ctypes.data))
        if rc in (4, 6) and pool < (1 << 29):
            pool *= 4
            continue
        assert rc == 0, rc
        break
    k = len(workers)
    print(f"BA({n},{m}) random/asc t=n/2: dependency levels {int(out[k])}, critical path {out[k + 1] / 1e3:.2f} ms, "
          f"gathered length mean {out[k + 2]:.1f} max {int(out[k + 3])}")
    for i, wv in enumerate(workers):
        print(f"   {wv:5d} waves ({max(1, wv // 16):4d} CUs at 16 waves): {out[i] / 1e3:9.2f} ms")


if __name__ == "__main__":
    a = sys.argv[1:] or ["c5"]
    if a[0] == "c3":
        model(1000000, 10)
    elif a[0] == "c5":
        model(4096, 8)
    elif a[0] == "cora":
        model(2708, 2)
    else:
        model(int(a[0]), int(a[1]))
