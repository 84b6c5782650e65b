"""bench.py -- headline benchmark of the rLap augmentor on MI355X.

One "step" = one full pass of the hot path (COO in HBM -> sc_edge_info in HBM).

  --workload c3 (default)  BASELINE.json configs[2]: synthetic Barabasi-Albert graph, 1M nodes, m=10
                           (~2e7 directed entries), num_remove = N/2, o_v="degree", o_n="asc".
                           N>1: every rank eliminates its own graph of that shape (replicas: a single
                           graph does not shard) and the (m,3) outputs are all-gathered
                           over RCCL.  "scaling": "weak".
  --workload c5            BASELINE.json configs[4]: a batch of 1024 BA(4096, m=8) graphs, num_remove = 2048
                           each, o_v="random" o_n="asc"; rank r eliminates graphs shard_range(1024, r, N)
                           (128 per GPU at N=8) with ONE batched call, then RCCL all-gather of sc_edge_info.
                           The total work is fixed: "scaling": "strong".

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (one child process per
GPU, before anything touches the GPU in this process).  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C5_GRAPHS, C5_NODES, C5_M = 1024, 4096, 8


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["c3", "c5"], default="c3")
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--graphs", type=int, default=C5_GRAPHS, help="c5: graphs in the batch (all ranks together)")
    ap.add_argument("--o_v", default=None)
    ap.add_argument("--o_n", default="asc")
    ap.add_argument("--weighted", action="store_true", help="SURVEY 8(d) variant: w ~ U(0.5,1.5) per undirected edge, seed 3 (tie-free path)")
    ap.add_argument("--mode", choices=["exact", "frontier"], default="exact",
                    help="c3: 'frontier' = counter-based uniforms (rlap_set_rng_mode; not the reference's stream: never the headline line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-runs", type=int, default=5, help="c3: full oracle runs the CPU baseline is the median of (SURVEY 8(d) says 10; five of the headline workload take about 25 s)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--master-port", type=int, default=29541)
    return ap.parse_args(argv)


def spawn_plan(args, argv, env):
    """When `--gpus N` (N>1) is given without a torchrun environment: the N child commands + environments
    (one rank per GPU, rendezvous on 127.0.0.1).  None when this process is a rank itself."""
    if args.gpus <= 1 or "WORLD_SIZE" in env:
        return None
    plan = []
    for r in range(args.gpus):
        e = dict(env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                  "MASTER_PORT": str(args.master_port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        plan.append(([sys.executable, os.path.abspath(__file__)] + list(argv), e))
    return plan


def run_spawned(plan):
    procs = [subprocess.Popen(cmd, env=e, stdout=(subprocess.PIPE if i == 0 else subprocess.DEVNULL)) for i, (cmd, e) in enumerate(plan)]
    out, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(rcs)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cores():
    """Physical cores this process may run on (affinity mask, one per (package, core id) pair)."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores = set()
    cpu = phys = core = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":")[1])
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
            elif not line.strip() and cpu is not None:
                if cpu in allowed:
                    cores.add((phys, core if core is not None else cpu))
                cpu = phys = core = None
    except OSError:
        pass
    return max(1, len(cores) if cores else len(allowed))


def _kernels_sha():
    h = hashlib.sha256()
    for f in ("rlap_kernels.hip", "rlap_flow.hip", "rlap_flow.h", "rlap_wave_sort.h", "rlap_core.h", "rlap_api.hip", "Makefile"):
        h.update(open(os.path.join(ROOT, "rlap_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def _pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/): counters cannot be read from
    inside this process.  Reported only while the kernel sources are the ones that were profiled."""
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        src = {"file": "profiles/" + name, "measured_at": d.get("_commit"), "kernels_sha": d.get("_kernels_sha"),
               "stale": d.get("_kernels_sha") != _kernels_sha()}
        return d, src
    return {}, None


def _c5_cpu_one(g):
    import numpy as np
    import oracle
    from rlap_amd import graphs
    ei = graphs.barabasi_albert(C5_NODES, C5_M, 1000 + g).numpy()
    pm = np.random.RandomState(g).permutation(C5_NODES)
    t0 = time.perf_counter()
    oracle.approximate_cholesky(ei, None, C5_NODES, C5_NODES // 2, "random", "asc", perm=pm)
    return time.perf_counter() - t0


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    plan = spawn_plan(args, argv, os.environ)
    if plan is not None:            # nothing has touched the GPU in this process
        sys.exit(run_spawned(plan))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from rlap_amd import graphs, ops
    from rlap_amd.distributed import all_gather_edge_rows, shard_range
    c5 = args.workload == "c5"
    o_v = args.o_v or ("random" if c5 else "degree")
    o_n = args.o_n
    ops.set_timing(True)

    if not c5:
        n, m = args.nodes, args.m
        t = n // 2
        ei_cpu = graphs.barabasi_albert(n, m, 2 + rank)   # seed 2 = SURVEY 8(d) config C3
        ei = ei_cpu.to(dev)
        w_cpu = None
        if args.weighted:
            r_, c_ = ei_cpu.numpy()
            und = np.minimum(r_, c_) * n + np.maximum(r_, c_)
            uq, inv = np.unique(und, return_inverse=True)
            w_cpu = torch.from_numpy(np.random.RandomState(3 + rank).uniform(0.5, 1.5, uq.shape[0])[inv])
        w_dev = None if w_cpu is None else w_cpu.to(dev)
        perm = None
        if o_v == "random":
            g = torch.Generator(); g.manual_seed(1234 + rank)
            perm = torch.randperm(n, generator=g)
        perm_dev = None if perm is None else perm.to(dev)

        def step():
            return ops.approximate_cholesky(ei, w_dev, n, t, o_v, o_n, perm=perm_dev, seed=7, return_device="same", mode=args.mode)
        units_per_rank_nominal = min(t, n - 1)
    else:
        Gtot, n, m = args.graphs, C5_NODES, C5_M
        lo, hi = shard_range(Gtot, rank, world)
        eis = [graphs.barabasi_albert(n, m, 1000 + g) for g in range(lo, hi)]
        perms = [np.random.RandomState(g).permutation(n) for g in range(lo, hi)]
        if hi > lo:
            big, node_ptr = graphs.batch_disjoint(eis, [n] * (hi - lo))
            big = big.to(dev)
            perm_dev = torch.from_numpy(np.concatenate(perms)).to(dev) if o_v == "random" else None
        ts = [n // 2] * (hi - lo)

        def step():
            if hi == lo:
                return torch.zeros((0, 3), dtype=torch.float64, device=dev)
            sc, _ = ops.approximate_cholesky_batched(big, None, node_ptr, ts, o_v, o_n, perm=perm_dev, seed=7 + lo)
            return sc
        units_per_rank_nominal = (hi - lo) * (n // 2)

    def full_step():
        sc = step()
        if world > 1 and not args.no_gather:
            sc, _ = all_gather_edge_rows(sc)
        return sc

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # the very first call also builds the MT19937-64 table and allocates the workspace: reported, not hidden
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    first = None
    if args.warmup > 0:
        first = full_step()
        torch.cuda.synchronize(dev)
    first_call_ms = 1e3 * (time.perf_counter() - t0) if args.warmup > 0 else None
    retries_first = ops.last_stats["n_retries"] if (args.warmup > 0 and ops.last_stats) else 0
    del first
    for _ in range(max(args.warmup - 1, 0)):
        full_step()
    sync()
    t0 = time.perf_counter()
    kstats = []
    for _ in range(args.steps):
        sc = full_step()
        if ops.last_stats:
            kstats.append(dict(ops.last_stats))
    sync()
    elapsed = time.perf_counter() - t0
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    units = torch.tensor([float(kstats[-1]["n_eliminated"]) if kstats else 0.0, float(kstats[-1]["out_rows"]) if kstats else 0.0],
                         dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    n_elim_all, rows_all = float(units[0].item()), float(units[1].item())

    extra = {}
    if world == 1 and not c5:
        # the same graph with its COO rows in random order (what a caller that did not coalesce hands over): the call then sorts
        # (the timed steps above are fed a (col,row)-sorted COO -- like torch_geometric's coalesced graphs -- and skip the sort)
        shuf = torch.randperm(ei.shape[1], device=dev, generator=torch.Generator(device=dev).manual_seed(11))
        ei_u = ei[:, shuf].contiguous()
        w_u = None if w_dev is None else w_dev[shuf].contiguous()
        ops.approximate_cholesky(ei_u, w_u, n, t, o_v, o_n, perm=perm_dev, seed=7, return_device="same", mode=args.mode)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(2):
            sc_u = ops.approximate_cholesky(ei_u, w_u, n, t, o_v, o_n, perm=perm_dev, seed=7, return_device="same", mode=args.mode)
        torch.cuda.synchronize(dev)
        extra["unsorted_input_ms"] = 1e3 * (time.perf_counter() - t1) / 2
        extra["unsorted_input_setup_ms"] = ops.last_stats["ms_setup"]
        extra["unsorted_input_same_rows"] = bool(sc_u.shape == sc.shape and torch.equal(sc_u, sc))
        del ei_u, w_u, sc_u, shuf
    rand_line = None
    if world == 1 and not c5 and o_v == "degree" and args.mode == "exact":
        # the same graph in the reference's DEFAULT order (o_v="random", rlap/ops.py:7-14): the multi-CU dataflow kernel; reported beside
        # the headline (never instead of it) so that the driver's record carries it -- three timed calls, parity against the oracle below
        gr = torch.Generator(); gr.manual_seed(1234)
        perm_r = torch.randperm(n, generator=gr)
        perm_r_dev = perm_r.to(dev)
        ops.approximate_cholesky(ei, w_dev, n, t, "random", o_n, perm=perm_r_dev, seed=7, return_device="same")
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        rs = []
        for _ in range(3):
            sc_r = ops.approximate_cholesky(ei, w_dev, n, t, "random", o_n, perm=perm_r_dev, seed=7, return_device="same")
            rs.append(dict(ops.last_stats))
        torch.cuda.synchronize(dev)
        rand_line = {"o_v": "random", "ms_per_step": 1e3 * (time.perf_counter() - t1) / 3, "ms_elim": sum(x["ms_elim"] for x in rs) / 3,
                     "n_eliminated": rs[-1]["n_eliminated"], "out_rows": rs[-1]["out_rows"], "n_retries": sum(x["n_retries"] for x in rs),
                     "elimination_kernel": "k_eliminate_flow" if rs[-1]["n_rounds"] == 0 else "k_eliminate_batch"}
        rand_line["value"] = rand_line["n_eliminated"] / (rand_line["ms_per_step"] * 1e-3)
    if world == 1 and c5 and args.graphs >= 8:
        # one GPU's share of the 8-GPU run (BASELINE config 5: 128 graphs per GPU), measured here: the strong-scaling floor
        # of the configuration is this call's latency, projected scaling at 8 GPUs = full batch / shard (exchange not included)
        Gs = args.graphs // 8
        big_s, node_ptr_s = graphs.batch_disjoint(eis[:Gs], [n] * Gs)
        big_s = big_s.to(dev)
        perm_s = torch.from_numpy(np.concatenate(perms[:Gs])).to(dev) if o_v == "random" else None
        for _ in range(2):
            ops.approximate_cholesky_batched(big_s, None, node_ptr_s, [n // 2] * Gs, o_v, o_n, perm=perm_s, seed=7)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        sst = []
        for _ in range(5):
            ops.approximate_cholesky_batched(big_s, None, node_ptr_s, [n // 2] * Gs, o_v, o_n, perm=perm_s, seed=7)
            sst.append(dict(ops.last_stats))
        torch.cuda.synchronize(dev)
        ms_shard = 1e3 * (time.perf_counter() - t1) / 5
        extra["shard_of_8"] = {"graphs": Gs, "ms_per_step": ms_shard,
                               "phase_ms": {k: sum(x[k] for x in sst) / len(sst) for k in ("ms_setup", "ms_elim", "ms_output")},
                               "projected_speedup_at_8_gpus": (1e3 * elapsed / args.steps) / ms_shard}
    if rank == 0:
        st = kstats[-1]
        n_elim = st["n_eliminated"]
        avg = lambda k: sum(s[k] for s in kstats) / len(kstats)   # noqa: E731
        ms_elim, ms_merge, ms_compact, ms_output = avg("ms_elim"), avg("ms_sc_merge"), avg("ms_sc_compact"), avg("ms_output")
        D, L, mrows = st["n_draws"], st["live_entries"], st["out_rows"]
        nverts = n if not c5 else (hi - lo) * n
        S = nverts - n_elim
        # algorithmic bytes: 12 B per directed entry read, 24 B per entry / row written, 8 B per uniform
        elim_bytes = 12 * (D + n_elim) + 24 * D + 8 * D if not c5 else None
        merge_bytes = 12 * L + 4 * (S + 1) + 12 * mrows           # pass A: entries in, staged (nbr,w) out
        compact_bytes = 12 * mrows + 4 * (S + 1) + 24 * mrows     # pass B: staged rows in, (m,3) f64 out
        k9_bytes = 12 * L + 4 * (S + 1) + 24 * mrows              # SURVEY K9 as a whole: gather + merge + order + compaction
        peak = 8000.0
        pmc, pmc_src = _pmc_traffic()
        default_cfg = (not c5 and n == 1_000_000 and m == 10 and o_v == "degree" and o_n == "asc" and not args.weighted and args.mode == "exact")

        random_cfg = (not c5 and n == 1_000_000 and m == 10 and o_v == "random" and o_n == "asc" and not args.weighted and args.mode == "exact")

        def roof(name, b, ms):
            a = b / (ms * 1e-3) / 1e9 if (ms > 0 and b) else 0.0
            # counters are quoted only for the command they were collected on (the default one; for the dataflow kernel `--o_v random`)
            profiled = default_cfg or (random_cfg and name == "k_eliminate_flow")
            tr = pmc.get(name) if (profiled and pmc_src and not pmc_src["stale"]) else None
            return {"kernel": name, "bound": "hbm", "achieved": a, "peak": peak, "unit": "GB/s", "frac": a / peak,
                    "traffic": (tr["fetch_bytes"] + tr["write_bytes"]) if tr else None, "traffic_source": pmc_src,
                    "algorithmic_bytes": b, "ms": ms}
        if not c5:
            workload = (f"BA(N={n}, m={m}) nnz={st['nnz']}, num_remove={n // 2}, o_v={o_v}, o_n={o_n}, "
                        + ("weights U(0.5,1.5)" if args.weighted else "unit weights") + "; one graph per GPU"
                        + ("" if args.mode == "exact" else f"; mode={args.mode} (counter-based uniforms: NOT the reference's random stream)")
                        + ("" if world == 1 or args.no_gather else " + RCCL all-gather of sc_edge_info"))
        else:
            workload = (f"{args.graphs} x BA(N={n}, m={m}), num_remove={n // 2} each, o_v={o_v}, o_n={o_n}, unit weights; "
                        f"graphs sharded {hi - lo} per GPU, one batched call per rank"
                        + ("" if world == 1 or args.no_gather else " + RCCL all-gather of sc_edge_info"))
        out = {
            "metric": "eliminated-vertices/sec", "value": n_elim_all * args.steps / elapsed, "unit": "vertices/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if c5 else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload},
            "output_edges_per_s": rows_all * args.steps / elapsed,
            "out_rows": mrows, "n_eliminated": n_elim, "n_draws": D,
            "n_rounds": st["n_rounds"], "n_singles": st["n_singles"],
            "n_retries": sum(s["n_retries"] for s in kstats), "n_retries_first_call": retries_first,
            "first_call_ms": first_call_ms,
            "phase_ms": {k: avg(k) for k in ("ms_setup", "ms_elim", "ms_output", "ms_sc_merge", "ms_sc_compact", "ms_total")},
            # dominant kernel by time: the sequential-semantics elimination (latency bound, not bandwidth bound)
            # (o_v = random on a single graph: the multi-CU dataflow kernel, rlap_flow.hip -- n_rounds is 0 there)
            "elimination_kernel": "k_eliminate_flow" if (o_v == "random" and st["n_rounds"] == 0 and n_elim > 0) else "k_eliminate_batch",
            "roofline": roof("k_eliminate_flow" if (o_v == "random" and st["n_rounds"] == 0 and n_elim > 0) else "k_eliminate_batch", elim_bytes, ms_elim) if not c5 else roof("k_sc_compact", compact_bytes, ms_compact),
            "roofline_k9": roof("k9_output_pass", k9_bytes, ms_output),
            "roofline_sc_merge": roof("k_sc_merge", merge_bytes, ms_merge),
            "roofline_sc_compact": roof("k_sc_compact", compact_bytes, ms_compact),
        }
        if world == 1 and not args.no_cpu_baseline:
            import oracle  # cpu_baseline leg only
            if not c5:
                # median of `cpu_runs` full runs of the oracle (the reference's own harness repeats its timing too,
                # run_augmentor_benchmarks.sh:18); one run of the headline workload is about 5 s on one core
                runs = []
                for _ in range(max(args.cpu_runs, 1)):
                    t1 = time.perf_counter()
                    ref, ost_k = oracle.approximate_cholesky(ei_cpu.numpy(), None if w_cpu is None else w_cpu.numpy(), n, n // 2, o_v, o_n,
                                                             perm=None if perm is None else perm.numpy(), shuffle_seed=7, return_stats=True, mode=args.mode)
                    runs.append((ost_k["t_total"], time.perf_counter() - t1, ost_k))
                runs.sort(key=lambda r: r[0])
                _, cpu_s, ost = runs[len(runs) // 2]
                got = sc.cpu().numpy()
                out["cpu_baseline"] = {
                    "value": ost["n_eliminated"] / ost["t_total"], "unit": "vertices/s", "cores": 1, "kind": "port",
                    "sample": f"the full workload, median of {len(runs)} runs (oracle, 1 thread; core span {ost['t_total']:.2f}s of which elimination "
                              f"{ost['t_elim']:.2f}s, setup {ost['t_setup']:.2f}s, output {ost['t_output']:.2f}s; with numpy packing {cpu_s:.2f}s; "
                              f"all runs: {', '.join('%.2f' % r[0] for r in runs)} s); cpu={_cpu_model()} nproc={os.cpu_count()}",
                    "runs": len(runs),
                    "output_edges_per_s": ref.shape[0] / ost["t_total"],
                }
                out["parity_full_size"] = bool(got.shape == ref.shape and np.array_equal(got, ref))
                out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
                # phase against phase: the elimination loop alone, and everything around it
                gpu_other = avg("ms_total") - ms_elim
                out["speedup_elim_only"] = (ost["t_elim"] * 1e3) / ms_elim if ms_elim > 0 else None
                out["speedup_setup_plus_output"] = ((ost["t_total"] - ost["t_elim"]) * 1e3) / gpu_other if gpu_other > 0 else None
                if rand_line is not None:
                    t1 = time.perf_counter()
                    ref_r, ost_r = oracle.approximate_cholesky(ei_cpu.numpy(), None if w_cpu is None else w_cpu.numpy(), n, n // 2, "random", o_n,
                                                               perm=perm_r.numpy(), shuffle_seed=7, return_stats=True)
                    got_r = sc_r.cpu().numpy()
                    rand_line["parity_full_size"] = bool(got_r.shape == ref_r.shape and np.array_equal(got_r, ref_r))
                    rand_line["cpu_port_s"] = ost_r["t_total"]
                    rand_line["speedup_vs_cpu_port"] = ost_r["t_total"] * 1e3 / rand_line["ms_per_step"]
                    rand_line["speedup_elim_only"] = ost_r["t_elim"] * 1e3 / rand_line["ms_elim"] if rand_line["ms_elim"] > 0 else None
                    del ref_r, got_r
            else:
                import multiprocessing as mp
                cores = min(_usable_cores(), 64)
                sample = 8 * cores     # about 8 s of single-core work in all
                with mp.get_context("fork").Pool(cores) as pool:
                    t1 = time.perf_counter()
                    per = pool.map(_c5_cpu_one, range(sample))
                    wall = time.perf_counter() - t1
                out["cpu_baseline"] = {
                    "value": sample * (n // 2) / wall, "unit": "vertices/s", "cores": cores, "kind": "port",
                    "sample": f"{sample} of the {args.graphs} graphs, one oracle call per process on {cores} processes (= usable physical cores of "
                              f"this box's CPU share; nproc={os.cpu_count()}), wall {wall:.2f}s incl. graph generation; "
                              f"one call {1e3 * sum(per) / len(per):.1f} ms on one core; cpu={_cpu_model()}",
                }
                out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
                # parity of the timed step: every 64th graph against the oracle
                got = sc.cpu().numpy()
                rp = None
                _, rp = ops.approximate_cholesky_batched(big, None, node_ptr, ts, o_v, o_n, perm=perm_dev, seed=7 + lo)
                okc = True
                for g in range(0, hi - lo, 64):
                    ref = oracle.approximate_cholesky(eis[g].numpy(), None, n, n // 2, o_v, o_n, perm=perms[g], shuffle_seed=7 + lo + g)
                    blk = got[int(rp[g]):int(rp[g + 1])].copy(); blk[:, :2] -= g * n
                    okc = okc and blk.shape == ref.shape and np.array_equal(blk, ref)
                out["parity_sampled"] = bool(okc)
        if not c5:
            out["mode"] = args.mode
        if rand_line is not None:
            out["reference_default_order"] = rand_line
        out.update(extra)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
