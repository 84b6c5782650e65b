"""Batched-graph mode across the GPUs of one node.

The augmentor is embarrassingly parallel over graphs: rank r eliminates a contiguous
block of graphs on its own GPU (no data-path collective), then the variable-length
(m_g,3) f64 outputs are exchanged with one all-gather of the row counts and one
all-gather of the padded payload (RCCL over xGMI when the backend is "nccl";
gloo on CPU tensors in the unit tests).
"""
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(num_graphs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: the first (num_graphs % world) ranks get one extra."""
    base, rem = divmod(num_graphs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(local: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather (m_r, C) blocks of different lengths. Returns (cat, counts[world])."""
    world = dist.get_world_size(group)
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = torch.cat(counts)
    mx = int(counts.max().item())
    C = local.shape[1]
    if local.shape[0] == mx:
        pad = local.contiguous()
    else:
        pad = torch.empty((mx, C), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        pad[local.shape[0]:] = 0
    big = torch.empty((world * mx, C), dtype=local.dtype, device=local.device)
    try:
        dist.all_gather_into_tensor(big, pad, group=group)     # one flat receive buffer (RCCL: a single ring all-gather)
    except (RuntimeError, NotImplementedError, AttributeError):
        bufs = list(big.view(world, mx, C).unbind(0))
        dist.all_gather(bufs, pad, group=group)
    counts_h = counts.cpu()
    if bool((counts_h == mx).all()):
        return big, counts_h
    out = torch.cat([big[r * mx: r * mx + int(counts_h[r])] for r in range(world)], dim=0)
    return out, counts_h


def _pack_rows(sc: torch.Tensor) -> torch.Tensor:
    """(m,3) f64 [row, col, w] -> (m,2) int64 [(row << 32 | col), bits of w]; one HIP kernel on the GPU, torch ops on the CPU."""
    if sc.is_cuda:
        from . import ops
        sc = sc.contiguous()
        packed = torch.empty((sc.shape[0], 2), dtype=torch.int64, device=sc.device)
        lib, h = ops._handle(sc.device)
        rc = lib.rlap_pack_rows(h, sc.data_ptr(), sc.shape[0], packed.data_ptr())
        if rc != 0:
            ops._raise(rc)
        return packed
    ids = (sc[:, 0].to(torch.int64) << 32) | sc[:, 1].to(torch.int64)
    return torch.stack([ids, sc[:, 2].contiguous().view(torch.int64)], dim=1)


def _unpack_rows(allp: torch.Tensor) -> torch.Tensor:
    if allp.is_cuda:
        from . import ops
        allp = allp.contiguous()
        out = torch.empty((allp.shape[0], 3), dtype=torch.float64, device=allp.device)
        lib, h = ops._handle(allp.device)
        rc = lib.rlap_unpack_rows(h, allp.data_ptr(), allp.shape[0], out.data_ptr())
        if rc != 0:
            ops._raise(rc)
        return out
    out = torch.empty((allp.shape[0], 3), dtype=torch.float64, device=allp.device)
    out[:, 0] = (allp[:, 0] >> 32).to(torch.float64)
    out[:, 1] = (allp[:, 0] & 0xFFFFFFFF).to(torch.float64)
    out[:, 2] = allp[:, 1].contiguous().view(torch.float64)
    return out


def all_gather_edge_rows(sc: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather of (m_r, 3) f64 [row, col, w] blocks with the two node ids packed into one 64-bit word for the
    exchange: 16 bytes per row over xGMI instead of 24.  Returns the same (cat, counts) as all_gather_rows."""
    allp, counts = all_gather_rows(_pack_rows(sc), group=group)
    return _unpack_rows(allp), counts


def sharded_approximate_cholesky(
    edge_indices: Sequence[torch.Tensor],
    edge_weights: Optional[Sequence[Optional[torch.Tensor]]],
    num_nodes: Sequence[int],
    num_remove: Sequence[int],
    o_v: str,
    o_n: str,
    *,
    seed: int = 0,
    group=None,
    gather: bool = True,
    compute_fn: Optional[Callable] = None,
):
    """Every rank passes the same list of G graphs; rank r processes graphs
    shard_range(G, r, world) and (optionally) all ranks receive all outputs.

    Returns (sc_edge_info, row_ptr[G+1]) with node ids LOCAL to each graph when
    gather=True, else this rank's block (sc_edge_info, row_ptr[local G + 1]).
    `compute_fn(edge_index, weights, node_ptr, num_remove, o_v, o_n, seed)` defaults
    to the HIP batched op; tests inject a CPU function.
    """
    from . import graphs as _graphs
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    G = len(edge_indices)
    lo, hi = shard_range(G, rank, world)
    use_hip = compute_fn is None      # (recorded before the default is bound below: an empty shard must still hand
    if compute_fn is None:            #  the collectives a tensor on the same kind of device as the other ranks)
        from . import ops

        def compute_fn(ei, w, node_ptr, nrem, o_v_, o_n_, seed_):
            return ops.approximate_cholesky_batched(ei, w, node_ptr, nrem, o_v_, o_n_, seed=seed_)
    if hi > lo:
        ei, node_ptr = _graphs.batch_disjoint(edge_indices[lo:hi], num_nodes[lo:hi])
        w = None
        if edge_weights is not None and any(x is not None for x in edge_weights[lo:hi]):
            w = torch.cat([
                (x.reshape(-1).to(torch.float64) if x is not None else torch.ones(e.shape[1], dtype=torch.float64))
                for x, e in zip(edge_weights[lo:hi], edge_indices[lo:hi])])
        # graph g of the whole list runs with seed + g whatever the sharding is (the batched call adds the local index)
        sc, row_ptr = compute_fn(ei, w, node_ptr, list(num_remove[lo:hi]), o_v, o_n, seed + lo)
        # back to per-graph local ids
        if sc.shape[0]:
            # (on the device the rows live on: no host-side pass over the rows)
            gid = torch.bucketize(torch.arange(sc.shape[0], device=sc.device), row_ptr[1:].to(sc.device), right=True)
            off = node_ptr.to(sc.device)[gid].to(sc.dtype)
            sc = sc.clone()
            sc[:, 0] -= off
            sc[:, 1] -= off
        local_counts = (row_ptr[1:] - row_ptr[:-1]).cpu()
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if (use_hip and torch.cuda.is_available()) else torch.device("cpu")
        sc = torch.zeros((0, 3), dtype=torch.float64, device=dev)
        local_counts = torch.zeros(0, dtype=torch.int64)
    if not gather or world == 1:
        rp = torch.zeros(local_counts.numel() + 1, dtype=torch.int64)
        rp[1:] = torch.cumsum(local_counts, 0)
        return sc, rp
    # exchange: per-graph counts (padded to the largest shard), then rows -- with the two node ids packed into one 64-bit word,
    # 16 bytes per row over xGMI instead of 24: the exchange bench.py times (all_gather_edge_rows) is the one the API makes
    all_sc, _ = all_gather_edge_rows(sc, group=group)
    mxg = (G + world - 1) // world
    cpad = torch.zeros(mxg, dtype=torch.int64, device=sc.device)
    cpad[: local_counts.numel()] = local_counts.to(sc.device)
    cl = [torch.empty_like(cpad) for _ in range(world)]
    dist.all_gather(cl, cpad, group=group)
    counts = []
    for r in range(world):
        a, b = shard_range(G, r, world)
        counts.append(cl[r][: b - a].cpu())
    counts = torch.cat(counts)
    rp = torch.zeros(G + 1, dtype=torch.int64)
    rp[1:] = torch.cumsum(counts, 0)
    return all_sc, rp
