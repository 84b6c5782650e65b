"""Config 5 degree/asc after calls that leave other data in the workspace (reused, grown buffers): every graph against the oracle."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np, torch
from rlap_amd import graphs, ops
import oracle
G, n, m = 1024, 4096, 8
eis = [graphs.barabasi_albert(n, m, 1000 + g) for g in range(G)]
big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
bigc = big.cuda()
refs = {}
def check(tag):
    sc, rp = ops.approximate_cholesky_batched(bigc, None, node_ptr, [n // 2] * G, "degree", "asc", seed=5)
    sc = sc.cpu().numpy()
    bad = []
    for g in range(G):
        if g not in refs: refs[g] = oracle.approximate_cholesky(eis[g].numpy(), None, n, n // 2, "degree", "asc", shuffle_seed=5 + g)
        got = sc[int(rp[g]):int(rp[g + 1])].copy(); got[:, :2] -= g * n
        if got.shape != refs[g].shape or not np.array_equal(got, refs[g]):
            d = -1
            if got.shape == refs[g].shape: d = int(np.argmax((got != refs[g]).any(axis=1)))
            bad.append((g, got.shape[0], refs[g].shape[0], d))
    print(f"{tag}: mismatching (graph, rows, ref rows, first differing row): {bad}  stats {dict(ops.last_stats)['n_retries']}", flush=True)
# dirty the workspace the way the test suite does: other shapes, modes and sizes first
for k, (nn, mm, ov) in enumerate([(200000, 10, "random"), (20000, 7, "coarsen"), (300000, 5, "degree"), (700, 60, "random")]):
    ei = graphs.barabasi_albert(nn, mm, 7 + k).cuda()
    ops.approximate_cholesky(ei, None, nn, nn // 2, ov, "desc", seed=3, return_device="same")
    check(f"after BA({nn},{mm}) {ov}")
