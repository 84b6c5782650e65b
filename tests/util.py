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


def introsort_killer(n):
    """Keys (distinct ints as float64) that drive libstdc++'s std::sort (threshold 16, median-of-3 to first, unguarded
    Hoare partition, depth limit 2*floor(log2 n)) into its heap-sort branch: McIlroy's adversary ("A Killer Adversary for
    Quicksort", 1999) run against a restatement of the introsort loop.  Returns (keys, hit_depth_limit)."""
    GAS = 1 << 60
    val = [GAS] * n
    state = {"nsolid": 0, "cand": 0}

    def less(x, y):   # x, y: item ids
        if val[x] == GAS and val[y] == GAS:
            if x == state["cand"]:
                val[x] = state["nsolid"]
            else:
                val[y] = state["nsolid"]
            state["nsolid"] += 1
        if val[x] == GAS:
            state["cand"] = x
        elif val[y] == GAS:
            state["cand"] = y
        return val[x] < val[y]

    a = list(range(n))
    depth0 = 2 * (n.bit_length() - 1)
    hit = False
    stack = [(0, n, depth0)]
    while stack:
        first, last, depth = stack.pop()
        while last - first > 16:
            if depth == 0:
                hit = True
                break
            depth -= 1
            ia, ib, ic = first + 1, first + (last - first) // 2, last - 1
            if less(a[ia], a[ib]):
                pick = ib if less(a[ib], a[ic]) else (ic if less(a[ia], a[ic]) else ia)
            elif less(a[ia], a[ic]):
                pick = ia
            elif less(a[ib], a[ic]):
                pick = ic
            else:
                pick = ib
            a[first], a[pick] = a[pick], a[first]
            pv = a[first]
            f, l = first + 1, last
            while True:
                while less(a[f], pv):
                    f += 1
                l -= 1
                while less(pv, a[l]):
                    l -= 1
                if not f < l:
                    break
                a[f], a[l] = a[l], a[f]
                f += 1
            stack.append((f, last, depth))
            last = f
    for i in range(n):
        if val[i] == GAS:
            val[i] = state["nsolid"]
            state["nsolid"] += 1
    return np.array(val, dtype=np.float64), hit
