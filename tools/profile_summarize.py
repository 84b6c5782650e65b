"""Condense the raw rocprofv3 output of tools/profile_round.sh into the tracked files under profiles/:
  <round>_kernel_stats_bench_c3.csv   per-kernel calls / total / average duration (kernel-trace --stats)
  <round>_pmc_fetch_write_bench_c3.csv  FETCH_SIZE / WRITE_SIZE per launch (KB as reported, uncorrected)
  <round>_pmc_traffic.json            bytes per launch for the kernels bench.py prices (fetch+write)
  <round>_sq_counters_k_eliminate_batch.csv  SQ counters of the elimination kernel
usage: profile_summarize.py r01 gpurun_out/prof_r01"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

rnd, src = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
os.makedirs(PROF, exist_ok=True)


def short(name):
    m = re.search(r"rocprim::[A-Za-z0-9_]+::detail::(radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|radix_sort_block_sort|merge_sort_block_merge_impl|scan_impl|init_lookback_scan_state_kernel)", name)
    if m:
        return "rocprim::" + m.group(1)
    name = name.split("(")[0]
    return name[5:] if name.startswith("void ") else name


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return hits[0]


# 1. kernel stats
rows = list(csv.DictReader(open(one("trace/**/*kernel_stats.csv"))))
agg = {}
for r in rows:
    a = agg.setdefault(short(r["Name"]), [0, 0])
    a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
with open(os.path.join(PROF, f"{rnd}_kernel_stats_bench_c3.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (7 calls of the op: 1 warm-up + 3 timed + 3 of the unsorted-input line; rocPRIM template names shortened)\n")
    f.write("kernel,calls,total_ms,avg_us,percent\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{k},{c},{t/1e6:.3f},{t/c/1e3:.1f},{100*t/tot:.3f}\n")

# 1b. the same for the config-5 workload, when the round's script collected it (trace_c5/)
hits_c5 = glob.glob(os.path.join(src, "trace_c5/**/*kernel_stats.csv"), recursive=True)
if hits_c5:
    agg5 = {}
    for r in csv.DictReader(open(hits_c5[0])):
        a = agg5.setdefault(short(r["Name"]), [0, 0])
        a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"])
    tot5 = sum(v[1] for v in agg5.values())
    with open(os.path.join(PROF, f"{rnd}_kernel_stats_bench_c5.csv"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline (7 calls of the 1024-graph batch + 7 of its 128-graph shard: BA(4096,8), o_v=random; measured at {os.environ.get('RLAP_COMMIT', '?')}; rocPRIM template names shortened)\n")
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for k, (c, t) in sorted(agg5.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k},{c},{t/1e6:.3f},{t/c/1e3:.1f},{100*t/tot5:.3f}\n")


# 2. PMC
def load(path, counter):
    d = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[short(r["Kernel_Name"])].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    return d


fz = load(one("fetch/**/*counter_collection.csv"), "FETCH_SIZE")
wz = load(one("write/**/*counter_collection.csv"), "WRITE_SIZE")
per = {}
with open(os.path.join(PROF, f"{rnd}_pmc_fetch_write_bench_c3.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline; KB as reported\n")
    f.write("kernel,launches,fetch_KB_per_launch,write_KB_per_launch,ms_per_launch(profiled)\n")
    for k in sorted(set(fz) | set(wz), key=lambda k: -sum(v for v, _ in fz.get(k, [(0, 0)]))):
        if not k.startswith("rlap::"):
            continue
        fv = fz.get(k, [(0, 0)]); wv = wz.get(k, [(0, 0)])
        fk = sum(v for v, _ in fv) / len(fv); wk = sum(v for v, _ in wv) / len(wv)
        per[k] = (fk * 1024.0, wk * 1024.0)
        f.write(f"{k},{len(fv)},{fk:.1f},{wk:.1f},{sum(t for _, t in fv)/len(fv):.3f}\n")


def pick(prefix):
    fb = wb = 0.0
    for k, (a, b) in per.items():
        if k.startswith(prefix):
            fb += a; wb += b
    return {"fetch_bytes": fb, "write_bytes": wb}


import hashlib
_h = hashlib.sha256()
for _f in ("rlap_kernels.hip", "rlap_flow.hip", "rlap_flow.h", "rlap_wave_sort.h", "rlap_core.h", "rlap_api.hip", "Makefile"):
    _h.update(open(os.path.join(ROOT, "rlap_amd", "csrc", _f), "rb").read())
traffic = {
    "_kernels_sha": _h.hexdigest()[:16],   # bench.py reports these figures only while the kernel sources are the profiled ones
    "_commit": os.environ.get("RLAP_COMMIT"),
    "_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline`, "
               f"profiles/{rnd}_pmc_fetch_write_bench_c3.csv; bytes = KB*1024, summed over the instantiations of a kernel; FETCH_SIZE is uncorrected "
               "(gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section; narrower accesses uncalibrated)",
    "k_eliminate_batch": pick("rlap::k_eliminate_batch_t"),
    "k_eliminate_flow": pick("rlap::k_eliminate_flow"),
    "k_sc_merge": pick("rlap::k_sc_merge_t"),
    "k_sc_merge_big": pick("rlap::k_sc_merge_big"),
    "k_sc_compact": pick("rlap::k_sc_compact"),
}
# the dataflow kernel is not launched by the default command: its traffic comes from the --o_v random passes
try:
    _fr = load(one("fetch_rand/**/*counter_collection.csv"), "FETCH_SIZE"); _wr = load(one("write_rand/**/*counter_collection.csv"), "WRITE_SIZE")
    _fb = sum(sum(v for v, _ in _fr[k]) / len(_fr[k]) for k in _fr if k.startswith("rlap::k_eliminate_flow")) * 1024.0
    _wb = sum(sum(v for v, _ in _wr[k]) / len(_wr[k]) for k in _wr if k.startswith("rlap::k_eliminate_flow")) * 1024.0
    if _fb > 0:
        traffic["k_eliminate_flow"] = {"fetch_bytes": _fb, "write_bytes": _wb, "command": "bench.py --o_v random --steps 1 --warmup 0 --no-cpu-baseline (separate FETCH_SIZE / WRITE_SIZE passes), per launch"}
except SystemExit:
    pass
json.dump(traffic, open(os.path.join(PROF, f"{rnd}_pmc_traffic.json"), "w"), indent=1)

# 3. SQ counters of the elimination kernel
sq = defaultdict(float)
nlaunch = set()
for r in csv.DictReader(open(one("sq/**/*counter_collection.csv"))):
    if short(r["Kernel_Name"]).startswith("rlap::k_eliminate_batch_t"):
        sq[r["Counter_Name"]] += float(r["Counter_Value"])
        nlaunch.add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
with open(os.path.join(PROF, f"{rnd}_sq_counters_k_eliminate_batch.csv"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --pmc SQ_* -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline; SUM over the {len(nlaunch)} launches of k_eliminate_batch_t<1,0,32> "
            "in that command (1 timed + 3 of the unsorted-input line): divide by the launch count for one launch\n")
    for k in sorted(sq):
        f.write(f"\"{k}\",{sq[k]:.6f}\n")

# 4. o_v = random at the same size: the dataflow kernel (rlap_flow.hip)
hits_r = glob.glob(os.path.join(src, "trace_rand/**/*kernel_stats.csv"), recursive=True)
if hits_r:
    aggr = {}
    for r in csv.DictReader(open(hits_r[0])):
        a = aggr.setdefault(short(r["Name"]), [0, 0])
        a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"])
    totr = sum(v[1] for v in aggr.values())
    with open(os.path.join(PROF, f"{rnd}_kernel_stats_bench_c3_random.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --o_v random --steps 3 --warmup 1 --no-cpu-baseline (7 calls of the op; rocPRIM template names shortened)\n")
        f.write("kernel,calls,total_ms,avg_us,percent\n")
        for k, (c, t) in sorted(aggr.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k},{c},{t/1e6:.3f},{t/c/1e3:.1f},{100*t/totr:.3f}\n")
hits_s = glob.glob(os.path.join(src, "sq_rand/**/*counter_collection.csv"), recursive=True)
if hits_s:
    sqr = defaultdict(float); nl = set(); grid = wg = None
    for r in csv.DictReader(open(hits_s[0])):
        if short(r["Kernel_Name"]).startswith("rlap::k_eliminate_flow"):
            sqr[r["Counter_Name"]] += float(r["Counter_Value"])
            nl.add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
            grid = r.get("Grid_Size", grid); wg = r.get("Workgroup_Size", wg)
    fr = wr = None
    try:
        fr = load(one("fetch_rand/**/*counter_collection.csv"), "FETCH_SIZE"); wr = load(one("write_rand/**/*counter_collection.csv"), "WRITE_SIZE")
    except SystemExit:
        pass
    with open(os.path.join(PROF, f"{rnd}_sq_counters_k_eliminate_flow.csv"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --pmc SQ_* -- python3 bench.py --o_v random --steps 1 --warmup 0 --no-cpu-baseline; SUM over the {len(nl)} launches of k_eliminate_flow "
                f"(grid {grid} threads in workgroups of {wg}: the eliminating wave and its helper, two workgroups per CU on every CU); SQ_BUSY_CU_CYCLES / SQ_BUSY_CYCLES = compute units busy on average\n")
        for k in sorted(sqr):
            f.write(f"\"{k}\",{sqr[k]:.6f}\n")
        if fr and wr:
            for k in fr:
                if k.startswith("rlap::k_eliminate_flow"):
                    f.write(f"\"FETCH_SIZE_KB_per_launch\",{sum(v for v, _ in fr[k]) / len(fr[k]):.1f}\n")
                    f.write(f"\"WRITE_SIZE_KB_per_launch\",{sum(v for v, _ in wr.get(k, [(0, 0)])) / max(len(wr.get(k, [])), 1):.1f}\n")
    print(dict(sqr))
print(open(os.path.join(PROF, f"{rnd}_kernel_stats_bench_c3.csv")).read()[:1500])
print(json.dumps(traffic, indent=1))
print(dict(sq))
