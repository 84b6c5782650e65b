"""Round statistics of the batch elimination on the CPU mirror (tests/csrc/host_mirror.cc): how many rounds,
how many vertices per round, and what ended the rounds.  usage: tests/tools/round_stats.py N m o_v o_n B bc"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest, test_core_mirror as tcm
from util import ba_graph

N, m, o_v, o_n, B, bc = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
lib = conftest.host_mirror.__wrapped__() if hasattr(conftest.host_mirror, "__wrapped__") else None
if lib is None:
    import ctypes, subprocess
    src = os.path.join(ROOT, "tests", "csrc", "host_mirror.cc"); so = os.path.join(ROOT, "tests", "csrc", "libhost_mirror.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-msse4.2", "-mavx", "-fPIC", "-shared", "-o", so, src])
    lib = ctypes.CDLL(so)
ei = ba_graph(N, m, 1)
perm = np.random.RandomState(10).permutation(N) if o_v == "random" else None
t0 = time.time()
_, _, st = tcm._mirror_batch(lib, ei, None, N, N // 2, o_v, o_n, B, perm=perm, seed=3, bc=bc)
print(f"{N=} {m=} {o_v}/{o_n} B={B} bc={bc}: rounds={st[0]} singles={st[1]} avgP={(N//2)/st[0]:.2f} contended={st[2]} "
      f"ended by: adjacent={st[3]} long={st[4]} multi-edge={st[5]} complex={st[6]} pre-empted={st[7]} nothing={st[8]}; singles: avg len {st[9]/max(st[1],1):.0f}, {st[10]} over 384 entries (avg {st[11]/max(st[10],1):.0f}), max {st[12]}; first dependent candidate: sampled-as-target={st[13]} last-neighbour={st[14]} patchable={st[15]} patch-makes-multi-edge={st[16]}  ({time.time()-t0:.1f}s)")
