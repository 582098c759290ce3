#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc run (counter_collection.csv + kernel_trace.csv).
usage: tools/pmc_summary.py <dir> [name filter]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    key = (r["Kernel_Name"].replace("void ecg::", "")[:52], r["Grid_Size"])
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if tr:
    for r in csv.DictReader(open(tr[0])):
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) if "Grid_Size_X" in r else r.get("Grid_Size", "")
        dur[(r["Kernel_Name"].replace("void ecg::", "")[:52], str(g))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for key, v in sorted(agg.items()):
    if flt not in key[0]:
        continue
    ds = dur.get(key)
    line = f"{key[0]:52s} grid={key[1]:>8s}"
    if ds:
        line += f" us={sum(ds) / len(ds) / 1e3:8.1f}"
    for c, vals in sorted(v.items()):
        line += f" {c}={sum(vals) / len(vals):.4g}"
    print(line)
