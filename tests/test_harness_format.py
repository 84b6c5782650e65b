"""the latency/memory harness prints what scripts/prepare_augmentor_stats.py:28-35 parses."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _reference_parse(text):
    # scripts/prepare_augmentor_stats.py:28-35, line for line in behaviour: split on single spaces, drop empties,
    # memory = token 3 of lines containing "aug(", latency = token 1 of lines containing "DURATION"
    mem_usage, latencies = [], []
    for line in text.splitlines(keepends=True):
        if "aug(" in line:
            tokens = [tok for tok in line.split(" ") if tok != ""]
            mem_usage.append(float(tokens[3]))
        if "DURATION" in line:
            tokens = [tok for tok in line.split(" ") if tok != ""]
            latencies.append(float(tokens[1]))
    return mem_usage, latencies


def test_lines_parse_with_the_reference_rules():
    import augmentor_latency as H
    text = H.format_memory_table(1234.5, 17.3) + H.format_duration(0.00123) + H.format_duration(2.5)
    mem, lat = _reference_parse(text)
    assert mem == [17.3] and lat == [0.00123, 2.5]
    assert H.parse_like_reference(text) == (mem, lat)
    # memory_profiler's own table layout: "Line #", usage, "MiB", increment, "MiB", occurrences, code
    row = [ln for ln in text.splitlines() if "aug(" in ln][0].split()
    assert row[2] == "MiB" and row[4] == "MiB" and row[1] == "1234.5"


@pytest.mark.gpu
@pytest.mark.parametrize("task,aug", [("node", "rLap"), ("node", "rLapPPRDiffusion"), ("graph", "rLapDegree")])
def test_harness_runs_and_parses(task, aug):
    extra = ["--nodes", "300", "--m", "3", "--graphs", "8", "--repeat", "3"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "augmentor_latency.py"), task, aug] + extra,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    mem, lat = _reference_parse(out.stdout)
    assert len(lat) == 3 and all(0 < v < 60 for v in lat)
    assert len(mem) == 1 and mem[0] >= 0
