"""Latency / memory micro-benchmark in the reference's reporting format
(scripts/augmentor_benchmarks.py:366-393, driven 10x per (augmentor, dataset, device) by
scripts/run_augmentor_benchmarks.sh:18-21), so that scripts/prepare_augmentor_stats.py:28-35 parses the
output unchanged:
  * a `DURATION: <sec> sec` line per run (the parser takes token 1 of lines containing "DURATION");
  * a memory_profiler-style table whose `aug(...)` line carries the memory increment as token 3
    (the parser takes token 3 of lines containing "aug(").  The reference profiles host memory of its CPU
    path; here the figure is DEVICE memory (MiB in use on the GPU before/after the call, from
    hipMemGetInfo: it includes the op's own workspace, which torch's allocator does not see).

  python tools/augmentor_latency.py node rLap --nodes 2708 --m 2          # Cora-sized stand-in
  python tools/augmentor_latency.py graph rLap --graphs 128 --nodes 4096  # one DataLoader-style batch
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def format_duration(seconds: float) -> str:
    return "\nDURATION: {} sec\n".format(seconds)                       # augmentor_benchmarks.py:375,392


def format_memory_table(usage_mib: float, increment_mib: float, line_no: int = 368) -> str:
    # memory_profiler's table (what `@profile()` prints for benchmark_node_memory, augmentor_benchmarks.py:366-368)
    head = "Line #    Mem usage    Increment  Occurrences   Line Contents\n" + "=" * 61 + "\n"
    row = "{:>6} {:>10.1f} MiB {:>10.1f} MiB {:>11}       aug(data.x, data.edge_index, data.edge_weight)\n".format(line_no, usage_mib, increment_mib, 1)
    return head + row


def parse_like_reference(text: str):
    """The parsing rules of scripts/prepare_augmentor_stats.py:28-35, restated for the tests."""
    mem, lat = [], []
    for line in text.splitlines(keepends=True):
        if "aug(" in line:
            tokens = [tok for tok in line.split(" ") if tok != ""]
            mem.append(float(tokens[3]))
        if "DURATION" in line:
            tokens = [tok for tok in line.split(" ") if tok != ""]
            lat.append(float(tokens[1]))
    return mem, lat


def _device_used_mib(torch):
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("task", choices=["node", "graph"])
    ap.add_argument("augmentor", choices=["rLap", "rLapDegree", "rLapCoarsen", "rLapPPRDiffusion"])
    ap.add_argument("--nodes", type=int, default=2708)
    ap.add_argument("--m", type=int, default=2)
    ap.add_argument("--graphs", type=int, default=128)
    ap.add_argument("--repeat", type=int, default=10)   # run_augmentor_benchmarks.sh:18 repeats 10x
    args = ap.parse_args(argv)
    import torch
    from rlap_amd import adapters, graphs
    fraction = 0.5                                        # augmentor_benchmarks.py:431
    o_v = {"rLap": "random", "rLapDegree": "degree", "rLapCoarsen": "coarsen", "rLapPPRDiffusion": "random"}[args.augmentor]
    cls = adapters.rLapPPRDiffusion if args.augmentor == "rLapPPRDiffusion" else adapters.rLap
    aug = cls(fraction, o_v=o_v, o_n="asc")
    dev = torch.device("cuda")
    print(args)
    if args.task == "node":
        ei = graphs.barabasi_albert(args.nodes, args.m, 1).to(dev)
        x = torch.randn(args.nodes, 16, device=dev)
        batches = [(x, ei)]
    else:
        # PyG's DataLoader(batch_size=128) hands the augmentor the disjoint union of a batch (augmentor_benchmarks.py:434,385-392)
        eis = [graphs.barabasi_albert(args.nodes, args.m, 10 + g) for g in range(args.graphs)]
        big, _ = graphs.batch_disjoint(eis, [args.nodes] * args.graphs)
        batches = [(torch.randn(args.nodes * args.graphs, 16, device=dev), big.to(dev))]
    for it in range(args.repeat):
        if hasattr(aug, "_cache"):
            aug._cache = None
        torch.cuda.synchronize()
        before = _device_used_mib(torch)
        duration = 0.0
        for xb, eb in batches:
            start = time.time()
            aug(xb, eb, None)
            torch.cuda.synchronize()
            duration += time.time() - start
        after = _device_used_mib(torch)
        if it == 0:   # the reference profiles memory once per process (a separate pass before the latency pass)
            sys.stdout.write(format_memory_table(after, after - before))
        sys.stdout.write(format_duration(duration))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
