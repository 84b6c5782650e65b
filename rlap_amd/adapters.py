"""Device-native augmentor adapters.

Mirrors of the reference's L2 adapters that keep everything on the GPU:
  * `rLap`     -- PyGCL-style augmentor, scripts/augmentor_benchmarks.py:68-96
  * `rLapDGL`  -- DGL-style augmentor, CCA-SSG/aug.py:33-63
PyGCL / DGL are optional: with them installed the classes return their graph types,
without them a small named tuple with the same fields.
"""
from collections import namedtuple
from typing import Optional

import torch

from . import ops

Graph = namedtuple("Graph", ["x", "edge_index", "edge_weights"])


def _schur(edge_index, edge_weights, x, frac, o_v, o_n, seed, num_nodes_from_x, symmetrize):
    """One call of the op the way the reference adapters make it: num_nodes = edge_index.max() + 1 and
    num_remove = int(frac * num_nodes) (scripts/augmentor_benchmarks.py:77-78), both found inside the C ABI
    (no torch reduction, no `.item()`).  `num_nodes_from_x=True` is the explicit opt-in to x.shape[0] instead
    (differs from the reference when the trailing nodes are isolated: more vertices enter the queue)."""
    n = int(x.shape[0]) if (num_nodes_from_x and x is not None) else None
    return ops.approximate_cholesky_from_edges(edge_index, edge_weights, n, None, o_v, o_n, remove_frac=frac,
                                               symmetrize=symmetrize, seed=seed, return_device="same")


class rLap:
    """PyGCL-style augmentor: `aug(x, edge_index, edge_weight)` or `aug.augment(g)` with g.unfold().

    Reference (scripts/augmentor_benchmarks.py:68-96): num_remove = int(frac * num_nodes);
    the returned graph drops the Schur-complement weights (`edge_weights=None`, :96) unless
    keep_weights=True.
    """

    def __init__(self, frac: float, o_v: str = "random", o_n: str = "asc", keep_weights: bool = False, seed: Optional[int] = None,
                 num_nodes_from_x: bool = False, symmetrize: bool = False):
        self.frac = frac
        self.o_v = o_v
        self.o_n = o_n
        self.keep_weights = keep_weights
        self.seed = seed
        self.num_nodes_from_x = num_nodes_from_x
        self.symmetrize = symmetrize      # True: one-directional input is made undirected inside the op (fused to_undirected)

    def augment(self, g):
        x, edge_index, edge_weights = g.unfold() if hasattr(g, "unfold") else g
        sc, num_nodes = _schur(edge_index, edge_weights, x, self.frac, self.o_v, self.o_n, self.seed, self.num_nodes_from_x, self.symmetrize)
        self.num_remove = int(self.frac * num_nodes)
        sampled_edge_index = sc[:, :2].long().t().contiguous()          # stays on the device
        w = sc[:, 2].contiguous() if self.keep_weights else None
        try:  # PyGCL present: return its Graph type
            import GCL.augmentors as A  # type: ignore
            return A.Graph(x=x, edge_index=sampled_edge_index, edge_weights=w)
        except Exception:
            return Graph(x, sampled_edge_index, w)

    def __call__(self, x, edge_index, edge_weight=None):
        return self.augment(Graph(x, edge_index, edge_weight))


def compute_ppr(edge_index: torch.Tensor, edge_weight: Optional[torch.Tensor], num_nodes: int, alpha: float = 0.2,
                eps: float = 1e-4, add_self_loop: bool = False, normalize_out: bool = True):
    """Dense personalised-PageRank diffusion, S = alpha (I - (1-alpha) D^-1/2 A D^-1/2)^-1, entries
    below `eps` dropped, then (normalize_out) the kept entries normalised symmetrically once more,
    D_S^-1/2 S D_S^-1/2 with D_S the row sums of the thresholded matrix.

    This restates PyGCL's `GCL.augmentors.functional.compute_ppr` (called by
    scripts/augmentor_benchmarks.py:152-159 with ignore_edge_attr=False, add_self_loop=False), which chains
    PyG's GDC steps: transition_matrix('sym') -> diffusion_matrix_exact('ppr') -> sparsify_dense('threshold')
    -> transition_matrix('sym').  PyGCL (pinned by the reference: requirements.txt:4, PyGCL==0.1.2) and PyG are third-party
    packages absent from this container and not installable (no network), so this row is UNPINNED: it follows the published
    semantics of those functions, not a run of them.  `normalize_out=False` gives
    the diffusion matrix before the last step.  torch ops on the input's device; meant for the sizes the
    reference uses it on (the Schur-complement subgraph)."""
    dev = edge_index.device
    w = torch.ones(edge_index.shape[1], dtype=torch.float64, device=dev) if edge_weight is None else edge_weight.to(torch.float64)
    adj = torch.zeros((num_nodes, num_nodes), dtype=torch.float64, device=dev)
    adj.index_put_((edge_index[0], edge_index[1]), w, accumulate=True)
    if add_self_loop:
        adj = adj + torch.eye(num_nodes, dtype=torch.float64, device=dev)
    deg = adj.sum(1)
    dinv = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
    a_hat = dinv[:, None] * adj * dinv[None, :]
    s = alpha * torch.linalg.inv(torch.eye(num_nodes, dtype=torch.float64, device=dev) - (1 - alpha) * a_hat)
    s = torch.where(s >= eps, s, torch.zeros_like(s))
    if normalize_out:
        d2 = s.sum(1)
        d2inv = torch.where(d2 > 0, d2.pow(-0.5), torch.zeros_like(d2))
        s = d2inv[:, None] * s * d2inv[None, :]
    idx = s.nonzero(as_tuple=False).t().contiguous()
    return idx, s[idx[0], idx[1]]


class rLapPPRDiffusion:
    """scripts/augmentor_benchmarks.py:99-171: Schur complement (weights kept) -> induced subgraph on
    the surviving nodes, relabelled -> PPR diffusion -> original ids; result cached for
    `refresh_cache_freq` calls like the reference."""

    def __init__(self, frac, o_v="random", o_n="asc", alpha=0.2, eps=1e-4, use_cache=True, refresh_cache_freq=50, seed=None,
                 num_nodes_from_x=False, normalize_out=True, weights_dtype=torch.float64):
        self.frac, self.o_v, self.o_n, self.alpha, self.eps = frac, o_v, o_n, alpha, eps
        self.use_cache, self.refresh_cache_freq = use_cache, refresh_cache_freq
        self._cache, self.refresh_cache_counter, self.seed = None, 0, seed
        self.num_nodes_from_x, self.normalize_out = num_nodes_from_x, normalize_out
        # The reference hands compute_ppr `torch.Tensor(sparse_edge_info[:, -1])` (augmentor_benchmarks.py:144-146).  On the torch of
        # this image (2.10) that expression keeps float64 (checked: torch.Tensor(f64 tensor).dtype is float64); on the legacy
        # constructor of older torch builds it is the default tensor type, float32.  float64 is the default here;
        # weights_dtype=torch.float32 rounds the Schur-complement weights once before the diffusion, as such a build would.
        self.weights_dtype = weights_dtype

    def augment(self, g):
        if self._cache is not None and self.use_cache and self.refresh_cache_counter < self.refresh_cache_freq:
            self.refresh_cache_counter += 1
            return self._cache
        x, edge_index, edge_weights = g.unfold() if hasattr(g, "unfold") else g
        sc, num_nodes = _schur(edge_index, edge_weights, x, self.frac, self.o_v, self.o_n, self.seed, self.num_nodes_from_x, False)
        self.num_remove = int(self.frac * num_nodes)
        ei = sc[:, :2].long().t()
        nodes = torch.unique(ei, sorted=True)                       # surviving nodes that still have edges
        relabel = torch.full((num_nodes,), -1, dtype=torch.int64, device=ei.device)
        relabel[nodes] = torch.arange(nodes.numel(), device=ei.device)
        sub_ei = relabel[ei]
        d_ei, d_w = compute_ppr(sub_ei, sc[:, 2].to(self.weights_dtype), nodes.numel(), alpha=self.alpha, eps=self.eps, normalize_out=self.normalize_out)
        res = Graph(x, nodes[d_ei], d_w)
        self._cache, self.refresh_cache_counter = res, 0
        return res

    def __call__(self, x, edge_index, edge_weight=None):
        return self.augment(Graph(x, edge_index, edge_weight))


class rLapDGL:
    """DGL-style augmentor (CCA-SSG/aug.py:33-63): edges -> (2,E) -> op (edge_weights=None) -> new graph."""

    def __init__(self, frac: float, o_v: str = "random", o_n: str = "asc", seed: Optional[int] = None):
        self.frac = frac
        self.o_v = o_v
        self.o_n = o_n
        self.seed = seed

    def augment(self, graph):
        try:
            import dgl  # type: ignore
        except ImportError:
            dgl = None
        if dgl is not None and hasattr(graph, "edges"):
            src, dst = graph.edges()
            num_nodes = graph.num_nodes()
        else:  # (edge_index, num_nodes) stand-in
            edge_index, num_nodes = graph
            src, dst = edge_index[0], edge_index[1]
        edge_index = torch.stack([src, dst])
        sc = ops.approximate_cholesky(edge_index, None, num_nodes, int(self.frac * num_nodes), self.o_v, self.o_n,
                                      seed=self.seed, return_device="same")
        ei = sc[:, :2].long().t()
        if dgl is not None and hasattr(graph, "edges"):
            return dgl.graph((ei[0], ei[1]), num_nodes=num_nodes)
        return ei, num_nodes
