"""Latency micro-benchmark in the reference's reporting format (scripts/augmentor_benchmarks.py:371-393):
prints `DURATION: <sec> sec` lines that scripts/prepare_augmentor_stats.py:28-35 parses.

  python tools/augmentor_latency.py node rLap --nodes 2708 --m 2          # Cora-sized stand-in
  python tools/augmentor_latency.py graph rLap --graphs 128 --nodes 4096  # one DataLoader-style batch
"""
import argparse
import sys
import time
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlap_amd import adapters, graphs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("task", choices=["node", "graph"])
    ap.add_argument("augmentor", choices=["rLap", "rLapDegree", "rLapCoarsen", "rLapPPRDiffusion"])
    ap.add_argument("--nodes", type=int, default=2708)
    ap.add_argument("--m", type=int, default=2)
    ap.add_argument("--graphs", type=int, default=128)
    ap.add_argument("--repeat", type=int, default=10)   # run_augmentor_benchmarks.sh:18 repeats 10x
    args = ap.parse_args()
    fraction = 0.5                                        # augmentor_benchmarks.py:431
    o_v = {"rLap": "random", "rLapDegree": "degree", "rLapCoarsen": "coarsen", "rLapPPRDiffusion": "random"}[args.augmentor]
    cls = adapters.rLapPPRDiffusion if args.augmentor == "rLapPPRDiffusion" else adapters.rLap
    aug = cls(fraction, o_v=o_v, o_n="asc")
    dev = torch.device("cuda")
    if args.task == "node":
        ei = graphs.barabasi_albert(args.nodes, args.m, 1).to(dev)
        x = torch.randn(args.nodes, 16, device=dev)
        for _ in range(args.repeat):
            if hasattr(aug, "_cache"):
                aug._cache = None
            torch.cuda.synchronize()
            start = time.time()
            aug(x, ei, None)
            torch.cuda.synchronize()
            print("\nDURATION: {} sec\n".format(time.time() - start))
    else:
        eis = [graphs.barabasi_albert(args.nodes, args.m, 10 + g) for g in range(args.graphs)]
        big, _ = graphs.batch_disjoint(eis, [args.nodes] * args.graphs)   # PyG DataLoader makes the same union
        big = big.to(dev)
        x = torch.randn(args.nodes * args.graphs, 16, device=dev)
        for _ in range(args.repeat):
            torch.cuda.synchronize()
            start = time.time()
            aug(x, big, None)
            torch.cuda.synchronize()
            print("\nDURATION: {} sec\n".format(time.time() - start))


if __name__ == "__main__":
    main()
