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


def _num_nodes(edge_index: torch.Tensor, x: Optional[torch.Tensor]) -> int:
    if x is not None:
        return int(x.shape[0])
    # reference: edge_index.max().item() + 1 (augmentor_benchmarks.py:77) -- one host sync
    return int(edge_index.max().item()) + 1 if edge_index.numel() else 0


class rLap:
    """PyGCL-style augmentor: `aug(x, edge_index, edge_weight)` or `aug.augment(g)` with g.unfold().

    Reference (scripts/augmentor_benchmarks.py:68-96): num_remove = int(frac * num_nodes);
    the returned graph drops the Schur-complement weights (`edge_weights=None`, :96) unless
    keep_weights=True.
    """

    def __init__(self, frac: float, o_v: str = "random", o_n: str = "asc", keep_weights: bool = False, seed: Optional[int] = None):
        self.frac = frac
        self.o_v = o_v
        self.o_n = o_n
        self.keep_weights = keep_weights
        self.seed = seed

    def augment(self, g):
        x, edge_index, edge_weights = g.unfold() if hasattr(g, "unfold") else g
        num_nodes = _num_nodes(edge_index, x)
        num_remove = int(self.frac * num_nodes)
        sc = ops.approximate_cholesky(edge_index, edge_weights, num_nodes, num_remove, self.o_v, self.o_n,
                                      seed=self.seed, return_device="same")
        sampled_edge_index = sc[:, :2].long().t().contiguous()          # stays on the device
        w = sc[:, 2].contiguous() if self.keep_weights else None
        try:  # PyGCL present: return its Graph type
            import GCL.augmentors as A  # type: ignore
            return A.Graph(x=x, edge_index=sampled_edge_index, edge_weights=w)
        except Exception:
            return Graph(x, sampled_edge_index, w)

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
