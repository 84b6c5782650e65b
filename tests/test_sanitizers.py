"""SURVEY section 5 (race detection / sanitizers): the host-side code of the path under -fsanitize=address,undefined.
tests/csrc/mirror_sanitize_main.cc runs the device data structures and the batch-round rules (rlap_core.h via host_mirror.cc:
32-slot candidates in rounds of 7 / 32 / 128, chains of dependent candidates on cliques, the patch rule, 64- and 128-slot
candidates, the single-vertex fall-back) and the oracle in one instrumented process and compares them bit for bit."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_mirror_and_oracle_under_asan_ubsan(tmp_path):
    exe = tmp_path / "mirror_san"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", "-msse4.2", "-mavx", "-o", str(exe),
                           os.path.join(HERE, "csrc", "mirror_sanitize_main.cc")])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "0 failures" in r.stdout
