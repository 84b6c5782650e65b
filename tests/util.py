"""Shared test helpers: seeded synthetic graphs (numpy) and canonical forms."""
import numpy as np


def ba_graph(n, m, seed):
    """Barabasi-Albert graph (networkx-style repeated-endpoint list), returned as a
    symmetric, coalesced (2,E) int64 edge_index sorted by (col,row)."""
    rng = np.random.RandomState(seed)
    targets = list(range(m))
    rep = []
    src, dst = [], []
    for i in range(m, n):
        src.extend([i] * len(targets))
        dst.extend(targets)
        rep.extend(targets)
        rep.extend([i] * len(targets))
        chosen = set()
        while len(chosen) < m:
            chosen.add(rep[rng.randint(len(rep))])
        targets = sorted(chosen)
    a = np.array(src, dtype=np.int64)
    b = np.array(dst, dtype=np.int64)
    return symmetrize(a, b, n)


def symmetrize(a, b, n):
    r = np.concatenate([a, b])
    c = np.concatenate([b, a])
    key = np.unique(c * n + r)
    return np.stack([key % n, key // n]).astype(np.int64)


def clique(n):
    a, b = np.triu_indices(n, 1)
    return symmetrize(a.astype(np.int64), b.astype(np.int64), n)


def path(n):
    a = np.arange(n - 1, dtype=np.int64)
    return symmetrize(a, a + 1, n)


def star(n):
    a = np.zeros(n - 1, dtype=np.int64)
    return symmetrize(a, np.arange(1, n, dtype=np.int64), n)


def grid2d(h, w):
    idx = np.arange(h * w).reshape(h, w)
    a = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()])
    b = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()])
    return symmetrize(a.astype(np.int64), b.astype(np.int64), h * w)


def sym_weights(edge_index, n, seed, lo=0.5, hi=1.5):
    """Tie-free symmetric weights: w(a,b) = w(b,a) drawn U(lo,hi) per undirected edge."""
    r, c = edge_index
    lo_id = np.minimum(r, c)
    hi_id = np.maximum(r, c)
    und = lo_id * n + hi_id
    uniq, inv = np.unique(und, return_inverse=True)
    rng = np.random.RandomState(seed)
    w = rng.uniform(lo, hi, size=uniq.shape[0])
    return w[inv]


def canonical(sc):
    """Rows sorted by (col,row): the edge-set view every caller consumes (L1 parity)."""
    sc = np.asarray(sc)
    if sc.shape[0] == 0:
        return sc
    order = np.lexsort((sc[:, 0], sc[:, 1]))
    return sc[order]
