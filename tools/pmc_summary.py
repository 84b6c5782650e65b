"""Per-kernel FETCH_SIZE / WRITE_SIZE (KB) from two rocprofv3 --pmc passes (counter_collection.csv)."""
import csv
import sys
from collections import defaultdict


def load(path, counter):
    d = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0]].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    return d


f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
print("kernel,launches,fetch_KB_per_launch,write_KB_per_launch,ms_per_launch(profiled)")
for k in sorted(set(f) | set(w), key=lambda k: -sum(v for v, _ in f.get(k, [(0, 0)]))):
    if not k.startswith("rlap::"):
        continue
    fv = f.get(k, [(0, 0)])
    wv = w.get(k, [(0, 0)])
    print(f"{k},{len(fv)},{sum(v for v,_ in fv)/len(fv):.1f},{sum(v for v,_ in wv)/len(wv):.1f},{sum(t for _,t in fv)/len(fv):.3f}")
