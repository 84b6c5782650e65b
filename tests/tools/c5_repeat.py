"""Run-to-run determinism of a batched call (config 5: 1024 x BA(4096,8)): the same call many times, every output compared
with the first one on the device.  usage: c5_repeat.py o_v iters [o_n]"""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
import numpy as np, torch
from rlap_amd import graphs, ops
o_v, iters = sys.argv[1], int(sys.argv[2])
o_n = sys.argv[3] if len(sys.argv) > 3 else "asc"
G, n, m = 1024, 4096, 8
eis = [graphs.barabasi_albert(n, m, 1000 + g) for g in range(G)]
big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
bigc = big.cuda()
perm = torch.from_numpy(np.concatenate([np.random.RandomState(g).permutation(n) for g in range(G)])) if o_v == "random" else None
first = None
nbad = 0
for it in range(iters):
    sc, rp = ops.approximate_cholesky_batched(bigc, None, node_ptr, [n // 2] * G, o_v, o_n, perm=perm, seed=5, return_device="same")
    if first is None:
        first, rp0 = sc.clone(), rp.clone()
        continue
    same = sc.shape == first.shape and bool(torch.equal(sc, first))
    if not same:
        nbad += 1
        bad = []
        rpl = [int(x) for x in rp]
        rp0l = [int(x) for x in rp0]
        for g in range(G):
            a, b = sc[rpl[g]:rpl[g + 1]], first[rp0l[g]:rp0l[g + 1]]
            if a.shape != b.shape or not bool(torch.equal(a, b)): bad.append(g)
        print(f"iteration {it}: differs from the first run in graphs {bad[:20]} (retries {dict(ops.last_stats)['n_retries']})", flush=True)
print(f"{o_v}/{o_n}: {iters} runs, {nbad} differ from the first", flush=True)
