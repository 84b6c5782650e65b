"""Soak run: randomized parity checks of the HIP path against the CPU oracle under the debug perturbations
(schedule jitter, poisoned workspace), single and batched calls, both workgroup shapes, all (o_v, o_n) modes, unit and tie-free
weights.  Prints a line every ten cases and a summary; exits non-zero on the first mismatch (naming the case number).
usage: soak.py SECONDS [SEED [FIRST_CASE [REPEAT]]]   -- FIRST_CASE skips the cases before it (same random stream);
REPEAT > 0 runs only FIRST_CASE, that many times, and counts the mismatches"""
import os
import sys
import time

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np

O_V = ["degree", "random", "coarsen"]
O_N = ["asc", "desc", "random"]


def draw(rs):
    """One case of the stream (every random draw of a case happens here, so a case can be replayed by number)."""
    c = {}
    c["o_v"], c["o_n"] = O_V[rs.randint(3)], O_N[rs.randint(3)]
    c["jitter"] = int(rs.choice([0, 0, 2, 5, 9]))
    c["poison"] = int(rs.choice([-1, 0, 255, 90]))
    c["shape"] = str(rs.choice(["", "1", "2"]))
    batched = rs.rand() < 0.5
    c["weighted"] = bool(rs.rand() < 0.35)
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
    c["seed"] = int(rs.randint(1 << 30))
    c["n"], c["m"], c["G"], c["t"], c["batched"] = n, m, G, int(frac * n), batched
    c["topo"] = str(rs.choice(["ba", "ba", "ba", "hub", "er", "grid", "cliques"]))
    # single unweighted PQ-order calls: a quarter goes through the fused entry point (one direction of every edge, shuffled;
    # num_nodes and num_remove found on the device)
    c["from_edges"] = bool(rs.rand() < 0.25) and not batched and not c["weighted"] and c["o_v"] != "random"
    c["check"] = list(range(G)) if G <= 8 else sorted(set(rs.randint(0, G, size=6).tolist()))
    return c


def make_graph(c, g):
    """Graph g of case c: (2,E) int64 numpy, symmetric and coalesced.  Topologies besides BA(n, m): `hub` = BA plus one vertex
    joined to a third of the others (long columns: the big single-vertex paths), `er` = uniform random pairs (isolated and
    degree-1 vertices), `grid` = 2-D grid of about n vertices, `cliques` = ring of (m+2)-cliques (multi-edges from the first
    elimination on)."""
    from rlap_amd import graphs
    from util import symmetrize
    n, m, seed = c["n"], c["m"], c["seed"] + g
    rs = np.random.RandomState(seed % (1 << 31))
    topo = c.get("topo", "ba")
    if topo == "ba":
        return graphs.barabasi_albert(n, m, seed).numpy()
    if topo == "hub":
        ei = graphs.barabasi_albert(n, m, seed).numpy()
        h = int(rs.randint(n))
        others = np.flatnonzero(rs.rand(n) < 0.33)
        others = others[others != h].astype(np.int64)
        return symmetrize(np.concatenate([ei[0], np.full(others.size, h, dtype=np.int64)]), np.concatenate([ei[1], others]), n)
    if topo == "er":
        k = max(1, n * m // 2)
        a, b = rs.randint(0, n, size=k).astype(np.int64), rs.randint(0, n, size=k).astype(np.int64)
        keep = a != b
        if not keep.any():
            a, b, keep = np.array([0], dtype=np.int64), np.array([1], dtype=np.int64), np.array([True])
        return symmetrize(a[keep], b[keep], n)
    if topo == "grid":
        h = max(2, int(np.sqrt(n)))
        w = max(2, n // h)
        idx = np.arange(h * w).reshape(h, w)
        a = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()]).astype(np.int64)
        b = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()]).astype(np.int64)
        return symmetrize(a, b, n)          # (vertices h*w .. n-1 stay isolated)
    q = min(m + 2, n)                       # cliques
    nq = n // q
    a, b = [], []
    iu, ju = np.triu_indices(q, 1)
    for z in range(nq):
        a.append(z * q + iu)
        b.append(z * q + ju)
        a.append(np.array([z * q + q - 1]))
        b.append(np.array([((z + 1) % nq) * q]))
    a, b = np.concatenate(a).astype(np.int64), np.concatenate(b).astype(np.int64)
    keep = a != b
    return symmetrize(a[keep], b[keep], n)


def describe(c):
    return (f"{c.get('topo', 'ba')} {c['o_v']}/{c['o_n']} n={c['n']} m={c['m']} G={c['G']} t={c['t']} weighted={c['weighted']} jitter={c['jitter']} "
            f"poison={c['poison']} shape={c['shape'] or 'auto'} seed={c['seed']}" + (" from_edges" if c.get("from_edges") else ""))


def run_case(c, check_all=False):
    """Returns the list of graphs of the case whose rows differ from the oracle's."""
    import torch
    import oracle
    from rlap_amd import graphs, ops
    from util import sym_weights
    n, G, t, seed, o_v, o_n = c["n"], c["G"], c["t"], c["seed"], c["o_v"], c["o_n"]
    mode = os.environ.get("SOAK_MODE", "exact")   # "frontier": counter-based uniforms in the library and in the oracle
    if c["shape"]:
        os.environ["RLAP_BATCH_SHAPE"] = c["shape"]
    else:
        os.environ.pop("RLAP_BATCH_SHAPE", None)
    ops.debug_set_jitter(c["jitter"])
    ops.debug_set_poison(c["poison"])
    eis = [torch.from_numpy(make_graph(c, g)) for g in range(G)]
    ws = [sym_weights(e.numpy(), n, seed + 7 * g) if c["weighted"] else None for g, e in enumerate(eis)]
    perms = [np.random.RandomState(seed + g).permutation(n) for g in range(G)]
    bad = []
    if c["batched"]:
        big, node_ptr = graphs.batch_disjoint(eis, [n] * G)
        w = None if not c["weighted"] else torch.from_numpy(np.concatenate(ws))
        perm = torch.from_numpy(np.concatenate(perms)) if o_v == "random" else None
        sc, rp = ops.approximate_cholesky_batched(big.cuda(), None if w is None else w.cuda(), node_ptr, [t] * G, o_v, o_n, perm=perm, seed=seed, mode=mode)
        sc = sc.cpu().numpy()
        for g in (range(G) if check_all else c["check"]):
            ref = oracle.approximate_cholesky(eis[g].numpy(), ws[g], n, t, o_v, o_n, perm=perms[g], shuffle_seed=seed + g, mode=mode)
            got = sc[int(rp[g]):int(rp[g + 1])].copy()
            got[:, :2] -= g * n
            if got.shape != ref.shape or not np.array_equal(got, ref):
                bad.append((g, got.shape, ref.shape))
    elif c.get("from_edges"):
        e = eis[0].numpy()
        up = e[:, e[0] < e[1]]
        up = up[:, np.random.RandomState(seed).permutation(up.shape[1])]
        n_ref = int(e.max()) + 1                       # (isolated top ids drop out: the reference's max + 1 rule)
        frac = t / n
        got, nn = ops.approximate_cholesky_from_edges(torch.from_numpy(np.ascontiguousarray(up)).cuda(), None, None, None, o_v, o_n,
                                                      remove_frac=frac, symmetrize=True, seed=seed, mode=mode)
        ref = oracle.approximate_cholesky(e, None, n_ref, int(frac * n_ref), o_v, o_n, shuffle_seed=seed, mode=mode)
        got = got.cpu().numpy()
        if nn != n_ref or got.shape != ref.shape or not np.array_equal(got, ref):
            bad.append((0, got.shape, ref.shape))
    else:
        got = ops.approximate_cholesky(eis[0].cuda(), None if not c["weighted"] else torch.from_numpy(ws[0]).cuda(), n, t, o_v, o_n,
                                       perm=torch.from_numpy(perms[0]) if o_v == "random" else None, seed=seed, mode=mode).numpy()
        ref = oracle.approximate_cholesky(eis[0].numpy(), ws[0], n, t, o_v, o_n, perm=perms[0], shuffle_seed=seed, mode=mode)
        if got.shape != ref.shape or not np.array_equal(got, ref):
            bad.append((0, got.shape, ref.shape))
    return bad


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    repeat = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    for _ in range(first):
        draw(rs)
    if repeat > 0:
        c = draw(rs)
        print("case", first, describe(c), flush=True)
        n_bad = 0
        for r in range(repeat):
            bad = run_case(c, check_all=True)
            n_bad += bool(bad)
            print(f"  run {r}: {'ok' if not bad else bad}", flush=True)
        print(f"{n_bad} of {repeat} runs differ from the oracle")
        sys.exit(1 if n_bad else 0)
    t_end = time.time() + budget
    n_cases, n_graphs = first, 0
    while time.time() < t_end:
        c = draw(rs)
        bad = run_case(c)
        if bad:
            print("MISMATCH case", n_cases, describe(c), bad, flush=True)
            sys.exit(1)
        n_graphs += len(c["check"])
        n_cases += 1
        if n_cases % 10 == 0:
            print(f"[{n_cases} cases, {n_graphs} graphs checked] last: {describe(c)}", flush=True)
    from rlap_amd import ops
    ops.debug_set_jitter(0)
    ops.debug_set_poison(-1)
    print(f"soak ok: cases {first}..{n_cases - 1}, {n_graphs} graphs bit-exact against the oracle in {budget:.0f} s")


if __name__ == "__main__":
    main()
