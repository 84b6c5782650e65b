"""Shorten rocprofv3 kernel_stats.csv (rocPRIM template names) into a readable summary."""
import csv
import re
import sys


def short(name):
    m = re.search(r"rocprim::[A-Za-z0-9_]+::detail::(radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|radix_sort_block_sort|merge_sort_block_merge_impl|scan_impl|init_lookback_scan_state_kernel)", name)
    if m:
        return "rocprim::" + m.group(1)
    return name.split("(")[0]


rows = list(csv.DictReader(open(sys.argv[1])))
agg = {}
for r in rows:
    k = short(r["Name"])
    a = agg.setdefault(k, [0, 0])
    a[0] += int(r["Calls"])
    a[1] += int(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
print("kernel,calls,total_ms,avg_us,percent")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k},{c},{t/1e6:.3f},{t/c/1e3:.1f},{100*t/tot:.3f}")
