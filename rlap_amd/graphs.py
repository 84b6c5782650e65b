"""Synthetic inputs for benches and tests (host side; not part of the hot path).

`barabasi_albert` follows the networkx / PyG construction (repeated-endpoint list),
seeded, and returns the symmetric coalesced edge_index the reference's tests feed
the op (tests/test_rlap.py:25-31: barabasi_albert_graph + to_undirected).
"""
import numpy as np
import torch

from . import _lib


def barabasi_albert(num_nodes: int, num_edges: int, seed: int = 0) -> torch.Tensor:
    lib = _lib.load()
    cap = lib.rlap_util_ba_graph(num_nodes, num_edges, seed, None, None)
    row = np.empty(max(cap, 1), dtype=np.int64)
    col = np.empty(max(cap, 1), dtype=np.int64)
    E = lib.rlap_util_ba_graph(num_nodes, num_edges, seed, row.ctypes.data, col.ctypes.data)
    return torch.from_numpy(np.stack([row[:E], col[:E]]))


def batch_disjoint(edge_indices, num_nodes):
    """Concatenate graphs into one disjoint union; returns (edge_index, node_ptr)."""
    ptr = [0]
    parts = []
    for ei, n in zip(edge_indices, num_nodes):
        parts.append(ei + ptr[-1])
        ptr.append(ptr[-1] + int(n))
    return torch.cat(parts, dim=1), torch.tensor(ptr, dtype=torch.int64)


def to_undirected(edge_index: torch.Tensor, num_nodes: int = None) -> torch.Tensor:
    """Device-side stand-in for the step the reference's scripts run before the op (PyG `to_undirected`,
    scripts/node_shared.py:326-327, tests/test_rlap.py:31): add the reverse of every edge and drop duplicates.
    Runs where `edge_index` lives (one sort of 64-bit keys); returns edges sorted by (row, col).  Pass `num_nodes`
    to avoid the host sync of `edge_index.max()`.  When the result only feeds the op, skip this helper:
    `ops.approximate_cholesky_from_edges(..., symmetrize=True)` does the same inside its COO->CSR kernels."""
    if edge_index.numel() == 0:
        return edge_index
    n = int(num_nodes) if num_nodes is not None else int(edge_index.max().item()) + 1
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1).to(torch.int64)
    key = torch.unique(both[0] * n + both[1])          # sorted, duplicates removed
    return torch.stack([torch.div(key, n, rounding_mode="floor"), key % n])
