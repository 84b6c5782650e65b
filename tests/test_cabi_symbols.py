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


def test_elimination_kernels_keep_their_register_and_lds_budget(tmp_path):
    """The elimination kernels are built for 128 VGPRs (16 waves per CU) and, in the 256-thread shape, 40 KB of LDS (four
    workgroups per CU).  Out-of-line device functions are compiled once for all their callers with the loosest budget among
    them, so an innocent new caller (a test hook, say) can silently halve the occupancy of every kernel that shares the function:
    read the numbers back from the code object inside the built library."""
    import pytest
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not found")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call([tools[0], "--dump-section", f".hip_fatbin={fat}", _lib.LIB_PATH, str(tmp_path / "unused.so")])
    # one offload bundle per translation unit (the elimination kernel's source is compiled twice, csrc/Makefile): read them all
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert starts, "no offload bundle in the library"
    notes = ""
    for k, st in enumerate(starts):
        part = str(tmp_path / f"fat{k}.bin")
        open(part, "wb").write(blob[st: starts[k + 1] if k + 1 < len(starts) else len(blob)])
        subprocess.check_call([tools[1], "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes += subprocess.check_output([tools[2], "--notes", co], text=True)
    kernels = {}
    cur = {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(name|vgpr_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "group_segment_fixed_size":     # first of the three keys of a kernel record (alphabetical order)
            cur = {"lds": int(val)}
        elif key == "name":
            cur["name"] = val
        elif key == "private_segment_fixed_size":
            cur["scratch"] = int(val)
        elif key == "vgpr_count" and "name" in cur:
            cur["vgpr"] = int(val)
            kernels[cur["name"]] = cur
    elim = {k: v for k, v in kernels.items() if "k_eliminate_batch_t" in k}
    assert len(elim) >= 21, sorted(kernels)[:5]
    for name, k in elim.items():
        assert k["vgpr"] <= 128, (name, k)
        # stack + spill bytes per lane: 1.3-1.6 KB today (the kernels' run time follows their spill code); a jump
        # means a new private array or a stack copy of the argument block in the round loop
        assert k.get("scratch", 0) <= 1800, (name, k)
        if "ELi256EE" in name:
            assert k["lds"] <= 40960, (name, k)
        else:
            assert k["lds"] <= 163840, (name, k)


def test_header_is_plain_c99(tmp_path):
    """The boundary is a C ABI: include/rlap_hip.h compiles as pedantic C99 with nothing but <stdint.h>/<stddef.h>, and a caller
    written in C (examples/cabi_caller.c: hipMalloc + the workspace contract + one call) compiles and links against the library."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "rlap_hip.h"\nint main(void) { rlap_handle h = 0; rlap_stats s; (void)h; (void)s; return RLAP_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), "-fsyntax-only", str(src)])
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include",
                           os.path.join(root, "examples", "cabi_caller.c"), "-o", str(tmp_path / "cabi_caller"),
                           "-L", os.path.join(root, "rlap_amd"), "-lrlap_hip", "-L", "/opt/rocm/lib", "-lamdhip64"])


def test_size_query_rejects_what_int32_slot_ids_cannot_hold():
    """rlap_workspace_bytes is host arithmetic up to the point where it asks rocPRIM for its temporary sizes: the range checks in front
    of that answer without a GPU -- 2^31 directed entries (slot ids are int32: RLAP_E_TOO_LARGE), a negative count (RLAP_E_BAD_ARG)."""
    import ctypes
    from rlap_amd import _lib
    lib = _lib.load()
    b, r = ctypes.c_size_t(), ctypes.c_int64()
    q = lambda E, n, G: lib.rlap_workspace_bytes(ctypes.c_int64(E), ctypes.c_int64(n), ctypes.c_int64(G), 0, ctypes.byref(b), ctypes.byref(r))
    assert q(1 << 31, 1000, 1) == 9      # RLAP_E_TOO_LARGE
    assert q(1000, 1 << 30, 1) == 9
    assert q(-5, 10, 1) == 3             # RLAP_E_BAD_ARG
    assert lib.rlap_status_string(9).decode() == "problem exceeds int32 slot ids"
