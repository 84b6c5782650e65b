"""The C-ABI library loads on a CPU-only box and exports every symbol include/rlap_hip.h declares."""
import ctypes
import os
import re

from rlap_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "rlap_hip.h")).read()
    declared = set(re.findall(r"\b(rlap_[a-z_0-9]+)\s*\(", hdr))
    declared.discard("rlap_handle_s")
    assert declared == set(_lib.EXPORTS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_status_strings_and_host_utilities():
    lib = _lib.load()
    assert _lib.status_string(0) == "ok"
    assert "symmetric" in _lib.status_string(1)
    # host-only utility: BA generator is symmetric, coalesced and sorted by (col,row)
    from rlap_amd import graphs
    ei = graphs.barabasi_albert(200, 5, 1).numpy()
    assert ei.shape[1] == 2 * 5 * (200 - 5)
    key = ei[1] * 200 + ei[0]
    assert (key[1:] > key[:-1]).all()
    fwd = set(map(tuple, ei.T))
    assert all((b, a) in fwd for a, b in fwd)


def test_ops_refuse_to_run_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rlap_amd import ops
    with pytest.raises(RuntimeError):
        ops.approximate_cholesky(torch.tensor([[0, 1], [1, 0]]), None, 2, 1, "degree", "asc")
