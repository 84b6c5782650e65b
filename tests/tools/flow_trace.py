"""Who holds the look-back front?  One dataflow elimination with RLAP_FLOW_TRACE (per-position time stamps), then the front's
progress is replayed on the host: a position 'holds' the front for as long as its count is published later than every earlier one's.
argv: n m [perm seed]"""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
os.environ["RLAP_FLOW"] = "1"
import numpy as np, torch
from rlap_amd import graphs, ops
n, m = int(sys.argv[1]), int(sys.argv[2])
eid = graphs.barabasi_albert(n, m, 1).cuda()
pt = torch.from_numpy(np.random.RandomState(0).permutation(n))
ops.set_timing(True)
ops.approximate_cholesky(eid, None, n, n // 2, "random", "asc", perm=pt, return_device="same")
path = os.path.join(_ROOT, "gpurun_out", "flow_trace.bin")
os.environ["RLAP_FLOW_TRACE"] = path
ops.approximate_cholesky(eid, None, n, n // 2, "random", "asc", perm=pt, return_device="same")
os.environ["RLAP_FLOW_TRACE"] = ""
print(dict(ops.last_stats))
tr = np.fromfile(path, dtype=np.int64).reshape(-1, 6)
os.remove(path)
tr = tr[tr[:, 4] > 0]                      # positions that ran (sentinels have no stamps)
t0 = tr[:, 0].min()
T = (tr[:, :5] - t0) / 100.0               # us
L = tr[:, 5]
print("positions", len(T), "span %.1f ms" % (T[:, 4].max() / 1e3))
pub = T[:, 2]
front = np.maximum.accumulate(pub)         # time at which every count up to here is known
prev = np.concatenate([[0.0], front[:-1]])
hold = np.maximum(pub - prev, 0.0)         # how much later than all earlier ones this count came: the front stood here that long
print("sum of holds %.1f ms over %d holders" % (hold.sum() / 1e3, (hold > 0).sum()))
wait_pend = T[:, 1] - T[:, 0]
own = T[:, 2] - T[:, 1]
for name, sel in (("long columns (live > 896)", L > 896), ("short, waited for a neighbour > 2 us", (L <= 896) & (wait_pend > 2.0)), ("short, no wait", (L <= 896) & (wait_pend <= 2.0))):
    print("  %-40s holders %7d  hold %8.1f ms   (positions %d; mean claim->pend0 %.1f us, pend0->count %.1f us)" % (name, (hold[sel] > 0).sum(), hold[sel].sum() / 1e3, sel.sum(), wait_pend[sel].mean() if sel.any() else 0, own[sel].mean() if sel.any() else 0))
big = np.argsort(-hold)[:15]
print("largest holds: (position, hold us, live, claim->pend0 us, pend0->count us)")
for i in big: print("   ", int(i), "%.1f" % hold[i], int(L[i]), "%.1f" % wait_pend[i], "%.1f" % own[i])
hs = np.sort(hold[hold > 0])[::-1]
print("hold histogram: >1ms %d (%.1f ms), 100us-1ms %d (%.1f ms), 10-100us %d (%.1f ms), <10us %d (%.1f ms)" % (
    (hs > 1000).sum(), hs[hs > 1000].sum() / 1e3, ((hs <= 1000) & (hs > 100)).sum(), hs[(hs <= 1000) & (hs > 100)].sum() / 1e3,
    ((hs <= 100) & (hs > 10)).sum(), hs[(hs <= 100) & (hs > 10)].sum() / 1e3, (hs <= 10).sum(), hs[hs <= 10].sum() / 1e3))
lb = T[:, 3] - T[:, 2]
print("count -> look-back done: mean %.1f us; look-back done -> end: mean %.1f us" % (lb.mean(), (T[:, 4] - T[:, 3]).mean()))
# how far behind the front's arrival does a position finish its look-back?
lag = T[:, 3] - np.maximum(front, pub)
print("look-back done minus (front reached me): mean %.1f us, p50 %.1f, p90 %.1f, p99 %.1f" % (lag.mean(), np.percentile(lag, 50), np.percentile(lag, 90), np.percentile(lag, 99)))
