"""Python host side of the augmentor: the reference's operator interface
(rlap/ops.py:7-63) on top of the C ABI of librlap_hip.so.

Same positional signature, same asserts, same return contract ((m,3) float64
[row, col, w], CPU tensor by default).  Differences, all deliberate:
  * inputs stay on the GPU (no `.cpu()` round trip, ops.py:47); CPU inputs are
    copied to the current HIP device -- there is no CPU compute path;
  * node ids are not squeezed through float32 (ops.py:47 promotes to f32 first,
    exact only below 2^24);
  * asymmetric input raises ValueError instead of exit(0) (factorizers.cc:19-22);
  * the randomness the reference takes from std::random_device is drawn from
    torch's RNG (or the keyword-only `perm` / `seed`), so runs are reproducible.
"""
from typing import Optional, Sequence, Tuple, Union
import ctypes
import threading

import torch
from torch import Tensor

from . import _lib

O_V = {"random": 0, "degree": 1, "coarsen": 2}
O_N = {"asc": 0, "desc": 1, "random": 2}

# One handle (workspace + stream binding) per (device, host thread): the reference builds a fresh
# ApproximateCholesky per call (py_api_binder.cc:57), so calls from several Python threads must not share
# state.  ctypes releases the GIL during a call, so two threads really run side by side on their streams.
_tls = threading.local()
_timing = False
last_stats = None  # rlap_stats of the most recent call (dict), for benches/tests


class _Handle:
    """C-ABI handle plus the two torch tensors it works in: the per-call arena and the cached uniform table.  The library
    allocates nothing itself (rlap_set_workspace): like the reference's result tensor (py_api_binder.cc:42) the memory comes
    from torch's caching allocator, so the op runs inside a training loop that holds most of the device memory."""

    def __init__(self, lib, idx):
        self.lib = lib
        self.ptr = ctypes.c_void_p()
        self.ws = None
        self.rng = None
        with torch.cuda.device(idx):
            rc = lib.rlap_create(ctypes.byref(self.ptr))
        if rc != 0:
            raise RuntimeError(f"rlap_create failed: {_lib.status_string(rc)}")

    def fit(self, dev, ws_bytes: int, rng_entries: int):
        """Make the arena / table at least this large (grow-only, 12.5 % slack: no reallocation inside a size class)."""
        changed = False
        if self.ws is None or self.ws.numel() < ws_bytes:
            self.ws = None   # (released first: the allocator may hand the same block back, grown)
            self.ws = torch.empty(int(ws_bytes) + int(ws_bytes) // 8 + 4096, dtype=torch.uint8, device=dev)
            changed = True
        if self.rng is None or self.rng.numel() < rng_entries:
            self.rng = None
            self.rng = torch.empty(int(rng_entries) + int(rng_entries) // 8 + 1024, dtype=torch.float64, device=dev)
            changed = True
        if changed:
            rc = self.lib.rlap_set_workspace(self.ptr, self.ws.data_ptr(), self.ws.numel(), self.rng.data_ptr(), self.rng.numel())
            if rc != 0:
                _raise(rc)

    def __del__(self):
        try:
            if self.ptr:
                self.lib.rlap_destroy(self.ptr)
        except Exception:
            pass


def _device_for(t: Optional[Tensor]) -> torch.device:
    if t is not None and t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("rlap_amd needs a HIP device (MI355X); no GPU is visible and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _handle_obj(device: torch.device):
    lib = _lib.load()
    idx = device.index if device.index is not None else torch.cuda.current_device()
    handles = getattr(_tls, "handles", None)
    if handles is None:
        handles = _tls.handles = {}
    hobj = handles.get(idx)
    if hobj is None:
        hobj = handles[idx] = _Handle(lib, idx)
    h = hobj.ptr
    lib.rlap_set_stream(h, ctypes.c_void_p(torch.cuda.current_stream(idx).cuda_stream))
    lib.rlap_set_timing(h, 1 if _timing else 0)
    return lib, hobj


def _handle(device: torch.device):
    lib, hobj = _handle_obj(device)
    return lib, hobj.ptr


MODES = {"exact": 0, "frontier": 1}


def _set_mode(lib, h, mode: str):
    """`mode` of SURVEY 8(b): "exact" (default) draws the reference's one MT19937-64 stream in elimination order -- results equal
    the reference's; "frontier" draws counter-based uniforms keyed by (seed, vertex, position): the same distribution, bit-exact
    against the oracle run in that mode, not against the reference (include/rlap_hip.h::rlap_set_rng_mode)."""
    assert mode in MODES, f"mode must be one of {sorted(MODES)}"
    rc = lib.rlap_set_rng_mode(h, MODES[mode])
    if rc != 0:
        _raise(rc)


def _trim(out: torch.Tensor, rows: int) -> torch.Tensor:
    """The first `rows` rows of the (E,3) result buffer.  A view keeps the whole buffer alive, a copy costs a pass over the
    result (0.6 ms for the 1.5 GB of the 1024-graph batch): copy only when the view would pin more than a third on top."""
    res = out[:rows]
    if 4 * rows < 3 * out.shape[0]:
        res = res.clone()
    return res


def _run(hobj, dev, E: int, n_total: int, G: int, symmetrize: bool, call, st):
    """One op call inside torch-owned memory: size the arena for (E, n_total, G), call, and when the library reports
    that it wants more (a growth limit was met and the repeated attempt is of a larger size class) grow and call again."""
    lib = hobj.lib
    ws_b, rng_n = ctypes.c_size_t(0), ctypes.c_int64(0)
    if n_total is None:
        # num_nodes is found on the device inside the call: no bound is asked for up front (the only host-known one, 2 * E vertices,
        # is off by twice the mean degree -- gigabytes at ogbn-arxiv size, and RLAP_E_TOO_LARGE long before the true n is).  The call
        # itself says what it wants at the true n (RLAP_E_WORKSPACE, below) when the arena at hand is too small.
        hobj.fit(dev, 4096, 1 << 16)
    else:
        rc = lib.rlap_workspace_query(hobj.ptr, E, n_total, G, 1 if symmetrize else 0, ctypes.byref(ws_b), ctypes.byref(rng_n))
        if rc != 0:
            _raise(rc)
        hobj.fit(dev, ws_b.value, rng_n.value)
    retries = 0
    for _ in range(8):
        rc = call()
        retries += int(st.n_retries)
        if rc != _lib.E_WORKSPACE:
            break
        rc2 = lib.rlap_workspace_needed(hobj.ptr, ctypes.byref(ws_b), ctypes.byref(rng_n))
        if rc2 != 0:
            _raise(rc2)
        hobj.fit(dev, ws_b.value, rng_n.value)
    st.n_retries = retries
    return rc


def set_timing(enable: bool):
    """Fill the ms_* fields of `last_stats` with HIP-event timings (process-wide: every handle, every device)."""
    global _timing
    _timing = bool(enable)


def debug_set_limits(pool_factor: float = -1.0, log_factor: float = -1.0, rng_len: int = -1, scratch_entries: int = -1, device=None):
    """Test hook: tiny first-attempt workspace limits for the calling thread's handle, so that the
    overflow -> retry path of the C ABI runs (last_stats["n_retries"])."""
    dev = _device_for(None) if device is None else torch.device(device)
    lib, h = _handle(dev)
    rc = lib.rlap_debug_set_limits(h, float(pool_factor), float(log_factor), int(rng_len), int(scratch_entries))
    if rc != 0:
        _raise(rc)


def debug_set_poison(byte: int = -1, device=None):
    """Debug aid: fill the workspace, the output buffer and the elimination kernel's LDS with `byte` before every attempt of
    the calling thread's next calls (negative = off); see include/rlap_hip.h."""
    dev = _device_for(None) if device is None else torch.device(device)
    lib, h = _handle(dev)
    rc = lib.rlap_debug_set_poison(h, int(byte))
    if rc != 0:
        _raise(rc)


def debug_set_jitter(quarter_us: int = 0, device=None):
    """Debug aid: waves of the elimination kernel sleep `quarter_us` x 0.25 us behind its barriers, a different subset each time
    (0 = off); see include/rlap_hip.h.  Results must not depend on it."""
    dev = _device_for(None) if device is None else torch.device(device)
    lib, h = _handle(dev)
    rc = lib.rlap_debug_set_jitter(h, int(quarter_us))
    if rc != 0:
        _raise(rc)


def _raise(rc: int):
    msg = _lib.status_string(rc)
    if rc in (1, 2, 3):
        raise ValueError(f"rlap: {msg}")
    raise RuntimeError(f"rlap: {msg} (status {rc})")


def _prep_edges(edge_index: Tensor, edge_weights: Optional[Tensor], dev: torch.device):
    assert edge_index.shape[0] == 2
    E = edge_index.shape[1]
    ei = edge_index.to(device=dev, dtype=torch.int64)
    row = ei[0].contiguous()
    col = ei[1].contiguous()
    w = None
    if edge_weights is not None:
        w = edge_weights.to(device=dev, dtype=torch.float64).reshape(-1).contiguous()  # (1,E) or (E,)
        assert w.numel() == E, "edge_weights must have one entry per edge"
    return row, col, w, E


def _seed_from(seed: Optional[int]) -> int:
    if seed is None:
        return int(torch.randint(0, 2**62, (1,)).item())
    return int(seed) & (2**64 - 1)


def approximate_cholesky(
    edge_index: Tensor,
    edge_weights: Optional[Tensor],
    num_nodes: int,
    num_remove: int,
    o_v: str,
    o_n: str,
    *,
    perm: Optional[Tensor] = None,
    seed: Optional[int] = None,
    return_device: Optional[Union[str, torch.device]] = "cpu",
    mode: str = "exact",
) -> Tensor:
    """Randomized Schur complement of the graph Laplacian (reference: rlap/ops.py:7-58).

    Keyword-only extras: `perm` (o_v="random": the elimination order vector, popped
    from the back, preconditioner.cc:588-613), `seed` (draws `perm` / the neighbour
    shuffles reproducibly), `return_device` ("cpu" as the reference, None/"same" to
    keep the result on the GPU).
    """
    assert edge_index.shape[0] == 2
    assert o_v in ["random", "degree", "coarsen"]
    assert o_n in ["asc", "desc", "random"]
    global last_stats
    dev = _device_for(edge_index)
    lib, hobj = _handle_obj(dev)
    h = hobj.ptr
    _set_mode(lib, h, mode)
    with torch.cuda.device(dev):
        row, col, w, E = _prep_edges(edge_index, edge_weights, dev)
        n = int(num_nodes)
        d_perm = None
        if o_v == "random" and perm is not None:
            d_perm = perm.to(device=dev, dtype=torch.int64).contiguous()
            assert d_perm.numel() == n
        # perm None: the reference shuffles 0..n-1 with std::random_device (preconditioner.cc:594-596); here the node_id vector is
        # drawn on the device from `seed` by the C ABI's keyed shuffle -- the same draw as approximate_cholesky_from_edges and as
        # graph 0 of approximate_cholesky_batched with that seed (ONE meaning of `seed` in this module)
        shuffle_seed = _seed_from(seed) if (o_n == "random" or o_v != "degree" or mode == "frontier") else 0
        out = torch.empty((max(E, 1), 3), dtype=torch.float64, device=dev)
        rows = ctypes.c_int64(0)
        st = _lib.Stats()
        rc = _run(hobj, dev, E, n, 1, False, lambda: lib.rlap_approx_chol(
            h, row.data_ptr(), col.data_ptr(), w.data_ptr() if w is not None else None, E, n, int(num_remove),
            O_V[o_v], O_N[o_n], d_perm.data_ptr() if d_perm is not None else None, shuffle_seed,
            out.data_ptr(), out.shape[0], ctypes.byref(rows), ctypes.byref(st)), st)
        if rc != 0:
            _raise(rc)
        last_stats = st.as_dict()
        res = _trim(out, rows.value)
    if return_device is None or return_device == "same":
        return res
    return res.to(return_device)


def approximate_cholesky_from_edges(
    edge_index: Tensor,
    edge_weights: Optional[Tensor] = None,
    num_nodes: Optional[int] = None,
    num_remove: Optional[int] = None,
    o_v: str = "random",
    o_n: str = "asc",
    *,
    remove_frac: float = 0.5,
    symmetrize: bool = True,
    perm: Optional[Tensor] = None,
    seed: Optional[int] = None,
    return_device: Optional[Union[str, torch.device]] = None,
    mode: str = "exact",
) -> Tuple[Tensor, int]:
    """The op with the step before it fused into its COO->CSR kernels:
    `to_undirected` + coalesce (scripts/node_shared.py:326-327) when `symmetrize`, and
    `num_nodes = edge_index.max() + 1`, `num_remove = int(remove_frac * num_nodes)`
    (scripts/augmentor_benchmarks.py:77-78) when they are None -- found on the device, without a
    torch reduction + `.item()`.  Returns (sc_edge_info on the device, num_nodes)."""
    assert edge_index.shape[0] == 2
    assert o_v in ["random", "degree", "coarsen"]
    assert o_n in ["asc", "desc", "random"]
    global last_stats
    dev = _device_for(edge_index)
    lib, hobj = _handle_obj(dev)
    h = hobj.ptr
    _set_mode(lib, h, mode)
    with torch.cuda.device(dev):
        row, col, w, E = _prep_edges(edge_index, edge_weights, dev)
        n = -1 if num_nodes is None else int(num_nodes)
        t = -1 if num_remove is None else int(num_remove)
        d_perm = None
        if o_v == "random" and perm is not None:
            assert n >= 0, "an injected perm needs num_nodes"
            d_perm = perm.to(device=dev, dtype=torch.int64).contiguous()
            assert d_perm.numel() == n
        # (perm None: the node_id vector is drawn on the device from the seed)
        shuffle_seed = _seed_from(seed) if (o_n == "random" or o_v != "degree" or mode == "frontier") else 0
        cap = max((2 * E) if symmetrize else E, 1)
        out = torch.empty((cap, 3), dtype=torch.float64, device=dev)
        rows = ctypes.c_int64(0)
        nn = ctypes.c_int64(0)
        st = _lib.Stats()
        rc = _run(hobj, dev, E, n if n >= 0 else None, 1, bool(symmetrize), lambda: lib.rlap_approx_chol_from_edges(
            h, row.data_ptr(), col.data_ptr(), w.data_ptr() if w is not None else None, E, n, t, float(remove_frac),
            1 if symmetrize else 0, O_V[o_v], O_N[o_n], d_perm.data_ptr() if d_perm is not None else None, shuffle_seed,
            out.data_ptr(), out.shape[0], ctypes.byref(rows), ctypes.byref(nn), ctypes.byref(st)), st)
        if rc != 0:
            _raise(rc)
        last_stats = st.as_dict()
        res = _trim(out, rows.value)
    if return_device is not None and return_device != "same":
        res = res.to(return_device)
    return res, int(nn.value)


def approximate_cholesky_batched(
    edge_index: Tensor,
    edge_weights: Optional[Tensor],
    node_ptr: Union[Tensor, Sequence[int]],
    num_remove: Union[Tensor, Sequence[int]],
    o_v: str,
    o_n: str,
    *,
    perm: Optional[Tensor] = None,
    seed: Optional[int] = None,
    return_device: Optional[Union[str, torch.device]] = None,
    mode: str = "exact",
) -> Tuple[Tensor, Tensor]:
    """Batched-graph mode: graph g owns node ids [node_ptr[g], node_ptr[g+1]).

    Every graph is eliminated independently -- what G separate reference calls would
    return, concatenated, with global node ids: graph g's rows equal
    `approximate_cholesky(graph g, ..., perm=perm[g], seed=seed + g)` shifted by node_ptr[g].
    Returns (sc_edge_info, row_ptr[G+1]).  `perm` concatenates per-graph permutations of LOCAL ids.
    """
    assert edge_index.shape[0] == 2
    assert o_v in ["random", "degree", "coarsen"]
    assert o_n in ["asc", "desc", "random"]
    global last_stats
    dev = _device_for(edge_index)
    lib, hobj = _handle_obj(dev)
    _set_mode(lib, hobj.ptr, mode)
    h = hobj.ptr
    np_ = torch.as_tensor(node_ptr, dtype=torch.int64).cpu().contiguous()
    nr_ = torch.as_tensor(num_remove, dtype=torch.int64).cpu().contiguous()
    G = np_.numel() - 1
    assert nr_.numel() == G
    with torch.cuda.device(dev):
        row, col, w, E = _prep_edges(edge_index, edge_weights, dev)
        N = int(np_[-1])
        d_perm = None
        if o_v == "random":
            # perm None: every graph's node_id vector is drawn on the device from the seed (keyed shuffle, C ABI)
            if perm is not None:
                d_perm = perm.to(device=dev, dtype=torch.int64).contiguous()
                assert d_perm.numel() == N
        shuffle_seed = _seed_from(seed) if (o_n == "random" or o_v != "degree" or mode == "frontier") else 0
        out = torch.empty((max(E, 1), 3), dtype=torch.float64, device=dev)
        row_ptr = torch.zeros(G + 1, dtype=torch.int64)
        st = _lib.Stats()
        rc = _run(hobj, dev, E, N, G, False, lambda: lib.rlap_approx_chol_batched(
            h, row.data_ptr(), col.data_ptr(), w.data_ptr() if w is not None else None, E, G,
            np_.data_ptr(), nr_.data_ptr(), O_V[o_v], O_N[o_n],
            d_perm.data_ptr() if d_perm is not None else None, shuffle_seed,
            out.data_ptr(), out.shape[0], row_ptr.data_ptr(), ctypes.byref(st)), st)
        if rc != 0:
            _raise(rc)
        last_stats = st.as_dict()
        res = _trim(out, int(row_ptr[-1]))
    if return_device is not None and return_device != "same":
        res = res.to(return_device)
    return res, row_ptr


def identity(a: Tensor) -> Tensor:
    """Boundary self-test (reference: rlap/ops.py:61-63): tensor -> column-major
    staging -> tensor, on the GPU; returns a tensor on `a`'s device."""
    assert a.dim() == 2
    dev = _device_for(a)
    lib, h = _handle(dev)
    with torch.cuda.device(dev):
        x = a.to(device=dev, dtype=torch.float64).contiguous()
        tmp = torch.empty_like(x)
        out = torch.empty_like(x)
        rc = lib.rlap_identity(h, x.data_ptr(), tmp.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1])
        if rc != 0:
            _raise(rc)
    return out.to(a.device)


def rng_uniforms(count: int, device=None) -> Tensor:
    """First `count` uniforms of the sampling stream as generated on the device."""
    dev = _device_for(None) if device is None else torch.device(device)
    lib, hobj = _handle_obj(dev)
    with torch.cuda.device(dev):
        out = torch.empty(count, dtype=torch.float64, device=dev)
        hobj.fit(dev, 0, max(int(count), 1 << 16))
        rc = lib.rlap_rng_uniforms(hobj.ptr, count, out.data_ptr())
        if rc != 0:
            _raise(rc)
    return out


# ---------------------------------------------------------------------------
# torch.ops.extension_cpp.* -- the reference's dispatcher-level interface
# (py_api_binder.cc:80-88), so third-party code calling the torch op still works.
# ---------------------------------------------------------------------------
def _op_approximate_cholesky(edge_info: Tensor, num_nodes: int, num_remove: int, o_v: str, o_n: str) -> Tensor:
    ei = edge_info
    edge_index = ei[:, :2].t().to(torch.int64)
    res = approximate_cholesky(edge_index, ei[:, 2], num_nodes, num_remove, o_v, o_n,
                               return_device="cpu" if not ei.is_cuda else "same")
    return res


def _op_identity(a: Tensor) -> Tensor:
    return identity(a)


def _register_torch_ops():
    try:
        lib = torch.library.Library("extension_cpp", "DEF")
        lib.define("approximate_cholesky(Tensor edge_info, int num_nodes, int num_remove, str o_v,  str o_n) -> Tensor")
        lib.define("identity(Tensor a) -> Tensor")
    except RuntimeError:
        return None  # namespace already defined (e.g. the reference extension is loaded too)
    for key in ("CPU", "CUDA"):
        lib.impl("approximate_cholesky", _op_approximate_cholesky, key)
        lib.impl("identity", _op_identity, key)
    return lib


_torch_lib = _register_torch_ops()
