"""Soak run: randomized parity checks of the HIP path against the CPU oracle under the debug perturbations
(schedule jitter, poisoned workspace), single and batched calls, both workgroup shapes, all (o_v, o_n) modes, unit and tie-free
weights.  Prints one line per case and a summary; exits non-zero on the first mismatch.
usage: soak.py SECONDS [SEED]"""
import os
import sys
import time

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import torch

import oracle
from rlap_amd import graphs, ops
from util import sym_weights

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_cases = n_graphs = 0
O_V = ["degree", "random", "coarsen"]
O_N = ["asc", "desc", "random"]
while time.time() < t_end:
    o_v, o_n = O_V[rs.randint(3)], O_N[rs.randint(3)]
    jitter = int(rs.choice([0, 0, 2, 5, 9]))
    poison = int(rs.choice([-1, 0, 255, 90]))
    shape = str(rs.choice(["", "1", "2"]))
    batched = rs.rand() < 0.5
    weighted = rs.rand() < 0.35
    kind = rs.randint(4)
    if kind == 0:
        n, m = int(rs.randint(40, 400)), int(rs.randint(1, 30))
    elif kind == 1:
        n, m = int(rs.randint(400, 6000)), int(rs.randint(1, 12))
    elif kind == 2:
        n, m = int(rs.randint(100, 900)), int(rs.randint(20, 70))       # dense: long columns, multi-edges
    else:
        n, m = int(rs.randint(6000, 40000)), int(rs.randint(2, 9))
    m = max(1, min(m, n - 1))
    G = int(rs.choice([2, 5, 40, 300])) if batched else 1
    if batched and n * G > 600000:
        G = max(2, 600000 // n)
    frac = float(rs.choice([0.1, 0.5, 0.5, 0.9, 1.0]))
    seed = int(rs.randint(1 << 30))
    if shape:
        os.environ["RLAP_BATCH_SHAPE"] = shape
    else:
        os.environ.pop("RLAP_BATCH_SHAPE", None)
    ops.debug_set_jitter(jitter)
    ops.debug_set_poison(poison)
    eis = [graphs.barabasi_albert(n, m, seed + g) for g in range(G)]
    ws = [sym_weights(e.numpy(), n, seed + 7 * g) if weighted else None for g, e in enumerate(eis)]
    perms = [np.random.RandomState(seed + g).permutation(n) for g in range(G)]
    t = int(frac * n)
    desc = f"{o_v}/{o_n} n={n} m={m} G={G} t={t} weighted={weighted} jitter={jitter} poison={poison} shape={shape or 'auto'}"
    if batched:
        big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
        w = None if not weighted else torch.from_numpy(np.concatenate(ws))
        perm = torch.from_numpy(np.concatenate(perms)) if o_v == "random" else None
        sc, rp = ops.approximate_cholesky_batched(big.cuda(), None if w is None else w.cuda(), node_ptr, [t] * G, o_v, o_n, perm=perm, seed=seed)
        sc = sc.cpu().numpy()
        check = range(G) if G <= 8 else sorted(set(rs.randint(0, G, size=6).tolist()))
        for g in check:
            ref = oracle.approximate_cholesky(eis[g].numpy(), ws[g], n, t, o_v, o_n, perm=perms[g], shuffle_seed=seed + g)
            got = sc[int(rp[g]):int(rp[g + 1])].copy()
            got[:, :2] -= g * n
            if got.shape != ref.shape or not np.array_equal(got, ref):
                print("MISMATCH", desc, "graph", g, got.shape, ref.shape, flush=True)
                sys.exit(1)
            n_graphs += 1
    else:
        got = ops.approximate_cholesky(eis[0].cuda(), None if not weighted else torch.from_numpy(ws[0]).cuda(), n, t, o_v, o_n,
                                       perm=torch.from_numpy(perms[0]) if o_v == "random" else None, seed=seed).numpy()
        ref = oracle.approximate_cholesky(eis[0].numpy(), ws[0], n, t, o_v, o_n, perm=perms[0], shuffle_seed=seed)
        if got.shape != ref.shape or not np.array_equal(got, ref):
            print("MISMATCH", desc, got.shape, ref.shape, flush=True)
            sys.exit(1)
        n_graphs += 1
    n_cases += 1
    if n_cases % 10 == 0:
        print(f"[{n_cases} cases, {n_graphs} graphs checked] last: {desc}", flush=True)
ops.debug_set_jitter(0)
ops.debug_set_poison(-1)
print(f"soak ok: {n_cases} cases, {n_graphs} graphs bit-exact against the oracle in {budget:.0f} s")
