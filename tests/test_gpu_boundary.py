"""The two boundary paths no earlier test called:
  * rlap_unpack_edge_info -- the INTEGRATION.md section 3 sequence a torch C++ binding would run: packed (E,3) f64 -> COO ->
    rlap_approx_chol, straight through ctypes on a fresh handle with its own workspace (reference: rlap/ops.py:47 packs,
    reader.cc:46-56 reads the packed matrix);
  * torch.ops.extension_cpp.approximate_cholesky with a DEVICE edge_info (the CUDA dispatch key of the op the reference
    registers for CPU only, py_api_binder.cc:54-69,85-88): result on the device, same rows as the oracle."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from util import ba_graph, sym_weights

pytestmark = pytest.mark.gpu


def test_unpack_edge_info_then_approx_chol_through_the_c_abi():
    from rlap_amd import _lib
    lib = _lib.load()
    n, t = 300, 150
    ei = ba_graph(n, 5, 3)
    w = sym_weights(ei, n, 4)
    E = ei.shape[1]
    packed = torch.from_numpy(np.stack([ei[0].astype(np.float64), ei[1].astype(np.float64), w], 1)).contiguous().cuda()   # (E,3) [row, col, w]
    h = ctypes.c_void_p()
    assert lib.rlap_create(ctypes.byref(h)) == 0
    try:
        lib.rlap_set_stream(h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        row = torch.empty(E, dtype=torch.int64, device="cuda")
        col = torch.empty(E, dtype=torch.int64, device="cuda")
        wd = torch.empty(E, dtype=torch.float64, device="cuda")
        assert lib.rlap_unpack_edge_info(h, packed.data_ptr(), E, row.data_ptr(), col.data_ptr(), wd.data_ptr()) == 0
        torch.cuda.synchronize()
        assert np.array_equal(row.cpu().numpy(), ei[0]) and np.array_equal(col.cpu().numpy(), ei[1]) and np.array_equal(wd.cpu().numpy(), w)
        out = torch.empty((E, 3), dtype=torch.float64, device="cuda")
        rows = ctypes.c_int64(0)
        for o_v, o_n in (("degree", "asc"), ("random", "desc")):
            perm = np.random.RandomState(1).permutation(n)
            pt = torch.from_numpy(perm).cuda()
            rc = lib.rlap_approx_chol(h, row.data_ptr(), col.data_ptr(), wd.data_ptr(), E, n, t, oracle.O_V[o_v], oracle.O_N[o_n],
                                      pt.data_ptr() if o_v == "random" else None, 3, out.data_ptr(), E, ctypes.byref(rows), None)
            assert rc == 0, _lib.status_string(rc)
            torch.cuda.synchronize()
            a = oracle.approximate_cholesky(ei, w, n, t, o_v, o_n, perm=perm if o_v == "random" else None, shuffle_seed=3)
            b = out[: rows.value].cpu().numpy()
            assert a.shape == b.shape and np.array_equal(a, b), (o_v, o_n)
    finally:
        lib.rlap_destroy(h)


def test_torch_op_with_a_device_edge_info():
    import rlap_amd  # noqa: F401  (registers torch.ops.extension_cpp.*)
    n, t = 120, 60
    ei = ba_graph(n, 4, 2)
    info = torch.from_numpy(np.concatenate([ei.astype(np.float64), np.ones((1, ei.shape[1]))], 0)).t().contiguous().cuda()
    out = torch.ops.extension_cpp.approximate_cholesky.default(edge_info=info, num_nodes=n, num_remove=t, o_v="degree", o_n="asc")
    assert out.is_cuda and out.dtype == torch.float64 and out.shape[1] == 3
    a = oracle.approximate_cholesky(ei, None, n, t, "degree", "asc")
    assert a.shape == tuple(out.shape) and np.array_equal(a, out.cpu().numpy())
    x = torch.randn(7, 5, dtype=torch.float64, device="cuda")
    y = torch.ops.extension_cpp.identity.default(a=x)
    assert y.is_cuda and torch.equal(x, y)
