#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel time per step, plus gaps.
usage: tools/prof_summary.py <dir with *_kernel_stats.csv | one *_kernel_stats.csv> <steps in the run> [top]
A directory may hold several runs (gpurun merges every call's files into the same local folder): the NEWEST
*_kernel_stats.csv is taken, and the kernel trace with the same process prefix."""
import csv
import glob
import os
import sys


def main():
    d, steps = sys.argv[1], float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    f = d if os.path.isfile(d) else sorted(glob.glob(d + "/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"kernel time per step: {tot / 1e6 / steps:.3f} ms over {sum(int(r['Calls']) for r in rows) / steps:.1f} launches/step")
    for r in rows[:top]:
        name = r["Name"].replace("void ecg::", "").replace("ecg::", "")[:64]
        print(f"{name:64s} calls/step={int(r['Calls']) / steps:5.1f} avg_us={float(r['AverageNs']) / 1e3:8.1f} "
              f"us/step={float(r['TotalDurationNs']) / 1e3 / steps:8.1f} {float(r['Percentage']):5.1f}%")
    tr = [t for t in [f.replace("_kernel_stats.csv", "_kernel_trace.csv")] if os.path.exists(t)]
    if tr:
        # the conv kernels serve several layers each: split every kernel's dispatches into layers by duration
        # (clusters within +-12 %), so that a layer's average can be read against bench.py's per-entry-point time
        per = {}
        for r in csv.DictReader(open(tr[0])):
            per.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print("per-layer split of the conv kernels (dispatch durations clustered within 12 %):")
        for name, ds in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            if "conv1d" not in name:
                continue
            ds.sort(reverse=True)
            clusters = []
            for v in ds:
                if clusters and v >= 0.88 * clusters[-1][0]:
                    clusters[-1].append(v)
                else:
                    clusters.append([v])
            short = name.replace("void ecg::", "").replace("ecg::", "")[:60]
            print(f"  {short:60s} " + "  ".join(f"{sum(c) / len(c) / 1e3:7.1f} us x{len(c) / steps:4.1f}/step" for c in clusters))
        ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(tr[0])))
        busy = sum(e - s for s, e in ev)
        span = ev[-1][1] - ev[0][0]
        print(f"trace: {len(ev)} dispatches, busy {busy / 1e6:.2f} ms of span {span / 1e6:.2f} ms ({100 * busy / span:.1f}% busy)")


if __name__ == "__main__":
    main()
