"""bench.py -- headline benchmark of the rLap augmentor on MI355X.

One "step" = one full approximate_cholesky pass (COO in HBM -> sc_edge_info in HBM)
over BASELINE.json configs[2]: synthetic Barabasi-Albert graph, 1M nodes, m=10
(~2e7 directed entries), num_remove = N/2, o_v="degree", o_n="asc".
N>1: every rank eliminates its own graph of that shape (weak scaling; the single
graph does not shard) and the (m,3) outputs are all-gathered over RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--o_v", default="degree")
    ap.add_argument("--o_n", default="asc")
    ap.add_argument("--weighted", action="store_true", help="SURVEY 8(d) variant: w ~ U(0.5,1.5) per undirected edge, seed 3 (tie-free path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from rlap_amd import graphs, ops
    n, m = args.nodes, args.m
    t = n // 2
    ei_cpu = graphs.barabasi_albert(n, m, 2 + rank)   # seed 2 = SURVEY 8(d) config C3
    ei = ei_cpu.to(dev)
    w_cpu = None
    if args.weighted:
        import numpy as np
        r_, c_ = ei_cpu.numpy()
        und = np.minimum(r_, c_) * n + np.maximum(r_, c_)
        uq, inv = np.unique(und, return_inverse=True)
        w_cpu = torch.from_numpy(np.random.RandomState(3 + rank).uniform(0.5, 1.5, uq.shape[0])[inv])
    w_dev = None if w_cpu is None else w_cpu.to(dev)
    perm = None
    if args.o_v == "random":
        g = torch.Generator(); g.manual_seed(1234 + rank)
        perm = torch.randperm(n, generator=g)
    ops.set_timing(True, dev)

    def step():
        sc = ops.approximate_cholesky(ei, w_dev, n, t, args.o_v, args.o_n, perm=perm, seed=7, return_device="same")
        if world > 1 and not args.no_gather:
            from rlap_amd.distributed import all_gather_edge_rows
            sc, _ = all_gather_edge_rows(sc)
        return sc

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    kstats = []
    rows = 0
    for _ in range(args.steps):
        sc = step()
        kstats.append(dict(ops.last_stats))
        rows = ops.last_stats["out_rows"]
    sync()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())

    if rank == 0:
        st = kstats[-1]
        n_elim = st["n_eliminated"]
        avg = lambda k: sum(s[k] for s in kstats) / len(kstats)
        ms_elim, ms_merge, ms_compact = avg("ms_elim"), avg("ms_sc_merge"), avg("ms_sc_compact")
        D, L, mrows = st["n_draws"], st["live_entries"], st["out_rows"]
        S = n - n_elim
        # algorithmic bytes: 12 B per directed entry read, 24 B per entry / row written, 8 B per uniform
        elim_bytes = 12 * (D + n_elim) + 24 * D + 8 * D
        merge_bytes = 12 * L + 4 * (S + 1) + 12 * mrows           # pass A: entries in, staged (nbr,w) out
        compact_bytes = 12 * mrows + 4 * (S + 1) + 24 * mrows     # pass B: staged rows in, (m,3) f64 out
        peak = 8000.0
        pmc = {}
        try:   # HBM bytes per launch from the committed PMC passes of this same command (profiles/)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        except OSError:
            pass
        default_cfg = (n == 1_000_000 and m == 10 and args.o_v == "degree" and args.o_n == "asc" and not args.weighted)
        def roof(name, b, ms):
            a = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            tr = pmc.get(name) if default_cfg else None
            return {"kernel": name, "bound": "hbm", "achieved": a, "peak": peak, "unit": "GB/s", "frac": a / peak,
                    "traffic": (tr["fetch_bytes"] + tr["write_bytes"]) if tr else None, "algorithmic_bytes": b, "ms": ms}
        out = {
            "metric": "eliminated-vertices/sec", "value": world * n_elim * args.steps / elapsed, "unit": "vertices/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BA(N={n}, m={m}) nnz={st['nnz']}, num_remove={t}, o_v={args.o_v}, o_n={args.o_n}, "
                                   + ("weights U(0.5,1.5)" if args.weighted else "unit weights") + "; one graph per GPU" + ("" if world == 1 or args.no_gather else " + RCCL all-gather of sc_edge_info")},
            "output_edges_per_s": world * mrows * args.steps / elapsed,
            "out_rows": mrows, "n_eliminated": n_elim, "n_draws": D,
            "phase_ms": {k: avg(k) for k in ("ms_setup", "ms_elim", "ms_output", "ms_sc_merge", "ms_sc_compact", "ms_total")},
            # dominant kernel by time: the sequential-semantics elimination wave (latency bound, not bandwidth bound)
            "roofline": roof("k_eliminate_batch", elim_bytes, ms_elim),
            "roofline_sc_merge": roof("k_sc_merge", merge_bytes, ms_merge),
            "roofline_sc_compact": roof("k_sc_compact", compact_bytes, ms_compact),
        }
        if world == 1 and not args.no_cpu_baseline:
            import numpy as np
            import oracle  # cpu_baseline leg only
            t1 = time.perf_counter()
            ref, ost = oracle.approximate_cholesky(ei_cpu.numpy(), None if w_cpu is None else w_cpu.numpy(), n, t, args.o_v, args.o_n,
                                                   perm=None if perm is None else perm.numpy(), shuffle_seed=7, return_stats=True)
            cpu_s = time.perf_counter() - t1
            got = sc.cpu().numpy()
            out["cpu_baseline"] = {
                "value": ost["n_eliminated"] / ost["t_total"], "unit": "vertices/s", "cores": 1, "kind": "port",
                "sample": f"the full workload once (oracle, 1 thread; core span {ost['t_total']:.2f}s of which elimination "
                          f"{ost['t_elim']:.2f}s; with numpy packing {cpu_s:.2f}s); cpu={_cpu_model()} nproc={os.cpu_count()}",
                "output_edges_per_s": ref.shape[0] / ost["t_total"],
            }
            out["parity_full_size"] = bool(got.shape == ref.shape and np.array_equal(got, ref))
            out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
