#!/usr/bin/env python3
"""Turn gpurun_out/ of tools/collect_profiles.sh into the artefacts committed under profiles/.

    python tools/make_profile_artifacts.py r03

profiles/<tag>_bench*.json            the bench lines as printed
profiles/<tag>_kernel_stats.csv       rocprofv3 --stats table of the bench command
profiles/<tag>_kernel_stats_summary.txt   the same per training step (tools/prof_summary.py)
profiles/<tag>_pmc_summary.txt        per ENTRY POINT (same keys as bench.py's `layers`): launches, µs, FETCH_SIZE,
                                      WRITE_SIZE, HBM bytes, MFMA-pipe busy fraction and clock
profiles/pmc_traffic.json             {"_meta": {commit, csrc_digest, source}, "entries": {key: HBM bytes per call}}
        traffic = 2 * FETCH_SIZE + WRITE_SIZE   (rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE counts
        half the bytes of wide coalesced reads — MI355X_MICROARCH.md, HBM section — hence the factor 2);
        busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * GRBM_GUI_ACTIVE / 8), clock = GRBM_GUI_ACTIVE / 8 / duration

The counter passes run tools/pmc_step.py, which launches a marker kernel in front of every ABI call and logs the
order of the entry points: the dispatches between two markers ARE that entry point's launches.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
OUT, PROF = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
MARKER = "relu_fwd_kernel"


def newest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    if not files:
        raise SystemExit(f"nothing matches {pattern}")
    return files[-1]


def per_entry(tag_dir):
    """One --pmc pass of tools/pmc_step.py -> {entry key: [per-call {counter: sum over its launches, "_ns", "_n"}]}."""
    order = json.load(open(tag_dir + ".order.json"))["order"]
    disp = collections.OrderedDict()
    rows = list(csv.DictReader(open(newest(os.path.join(tag_dir, "**", "*counter_collection.csv")))))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        e = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "c": {}})
        e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
        e["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    segs, cur = [], None
    for e in disp.values():
        if MARKER in e["name"]:
            cur = {"_ns": 0, "_n": 0}
            segs.append(cur)
        elif cur is not None and "ecg::" in e["name"]:
            for k, v in e["c"].items():
                cur[k] = cur.get(k, 0.0) + v
            cur["_ns"] += e["ns"]
            cur["_n"] += 1
    if len(segs) != len(order):
        raise SystemExit(f"{tag_dir}: {len(segs)} marked segments but {len(order)} logged entry points")
    out = collections.defaultdict(list)
    for key, s in zip(order, segs):
        out[key].append(s)
    return out


def mean(v):
    return sum(v) / len(v)


MAX_CLOCK_GHZ = 2.4        # MI355X peak engine clock: no dispatch runs faster
STAMPED_CLOCK_GHZ = 2.1    # what in-kernel s_memtime stamps read under the conv kernels (2.08-2.13)


def mfma_busy_columns(busy_cycles, gui_active_per_xcd, d_us):
    """MFMA-pipe busy fraction of one entry point.  SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles per SIMD (1024 SIMDs).
    The denominator needs the cycles the dispatch lasted: GRBM_GUI_ACTIVE / 8 is that on long dispatches, but it keeps
    counting before the first and after the last wave, so on dispatches of a few tens of microseconds it spans MORE
    than the kernel's own timestamps and the clock derived from it (GUI_ACTIVE / 8 / duration) reads above the
    2.4 GHz the chip can run at (round 2 printed 3.18 GHz for the 42 us block-0 weight gradient).  Such rows are
    flagged and priced with duration x the stamped clock instead; `mfma_busy_floor` (duration x 2.4 GHz, the largest
    the denominator can be) is a lower bound that holds for every row."""
    clock = gui_active_per_xcd / (d_us * 1e3)
    floor = busy_cycles / (1024 * d_us * 1e3 * MAX_CLOCK_GHZ)
    if clock <= MAX_CLOCK_GHZ:
        return f" mfma_busy={busy_cycles / (1024 * gui_active_per_xcd):.3f} clock_GHz={clock:.3f} mfma_busy_floor={floor:.3f}"
    est = busy_cycles / (1024 * d_us * 1e3 * STAMPED_CLOCK_GHZ)
    return (f" mfma_busy={min(est, 1.0):.3f}(at the stamped {STAMPED_CLOCK_GHZ} GHz) clock_GHz=invalid({clock:.2f}>2.4: "
            f"GUI_ACTIVE spans beyond a short dispatch) mfma_busy_floor={floor:.3f}")


def main():
    os.makedirs(PROF, exist_ok=True)
    traffic_only = "--traffic-only" in sys.argv       # on the GPU box, between the counter passes and the bench runs
    if traffic_only:
        return traffic_tables()
    for f in glob.glob(os.path.join(OUT, f"{TAG}_bench*.json")) + glob.glob(os.path.join(OUT, f"{TAG}_input_pipeline.json")) + glob.glob(os.path.join(OUT, f"{TAG}_inference.json")):
        shutil.copy(f, os.path.join(PROF, os.path.basename(f)))
    stats = newest(os.path.join(OUT, f"prof_{TAG}", "**", "*_kernel_stats.csv"))
    shutil.copy(stats, os.path.join(PROF, f"{TAG}_kernel_stats.csv"))
    steps = 20 + 20 + 5 + 5     # timed + its per-step-event repeat + warm-up + the instrumented pass of bench.py (--priming 0)
    summ = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"),
                           stats, str(steps), "40"], capture_output=True, text=True).stdout
    open(os.path.join(PROF, f"{TAG}_kernel_stats_summary.txt"), "w").write(
        f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 --priming 0  ({steps} steps incl. warm-up, the per-step-event repeat and the instrumented pass)\n" + summ)

    c5 = glob.glob(os.path.join(OUT, f"prof_{TAG}_c5bf16", "**", "*_kernel_stats.csv"), recursive=True)
    if c5:
        c5s = newest(os.path.join(OUT, f"prof_{TAG}_c5bf16", "**", "*_kernel_stats.csv"))
        shutil.copy(c5s, os.path.join(PROF, f"{TAG}_kernel_stats_bf16_config5.csv"))
        summ5 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), c5s, str(steps), "40"],
                               capture_output=True, text=True).stdout
        open(os.path.join(PROF, f"{TAG}_kernel_stats_bf16_config5_summary.txt"), "w").write(
            "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-also --priming-seconds 0 --dtype bf16 "
            f"--length 5000 --labels 1 --steps 20 --warmup 5 --priming 0  ({steps} steps incl. warm-up, the per-step-event repeat "
            "and the instrumented pass)\n" + summ5)
    lines = traffic_tables()
    print(summ[:1500])
    print("\n".join(lines[:40]))


def traffic_tables():
    lines = ["# per entry point of one B=256 train step (tools/pmc_step.py; same keys as bench.py `layers`):",
             "# launches per call, µs per call (under the counter pass), FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them",
             "# (separate passes), hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction),",
             "# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8), clock = GRBM_GUI_ACTIVE / 8 / duration;",
             "# a derived clock above 2.4 GHz means GUI_ACTIVE spans more than the (short) dispatch: the row is flagged and priced",
             "# with duration x the stamped 2.1 GHz instead; mfma_busy_floor = busy / (1024 * duration * 2.4 GHz) always holds"]
    entries = {}
    # (suffix of the counter-pass directories, title, leg tag = bench.leg_tag(model, labels, dtype, length): the traffic table
    # is keyed by leg AND entry point — ECGCNN(5) and ECGMultimodal share every conv signature)
    variants = [("", "ECGCNN(5), 12x1000, fp32 (the headline)", "cnn5_f32_1000"), ("_mm", "ECGMultimodal, 12x1000, fp32", "mm_f32_1000"),
                ("_c5f32", "ECGCNN(1), 12x5000, fp32 (BASELINE configs[4])", "cnn1_f32_5000"),
                ("_c5bf16", "ECGCNN(1), 12x5000, bf16 mode (BASELINE configs[4])", "cnn1_bf16_5000")]
    for suffix, title, leg in variants:
      if not os.path.exists(os.path.join(OUT, f"pmc_fetch_{TAG}{suffix}.order.json")):
          continue
      lines.append(f"## {title}")
      fetch = per_entry(os.path.join(OUT, f"pmc_fetch_{TAG}{suffix}"))
      write = per_entry(os.path.join(OUT, f"pmc_write_{TAG}{suffix}"))
      sq = per_entry(os.path.join(OUT, f"pmc_sq_{TAG}{suffix}"))
      for key in fetch:
          f = mean([s.get("FETCH_SIZE", 0.0) for s in fetch[key]])
          w = mean([s.get("WRITE_SIZE", 0.0) for s in write[key]]) if key in write else float("nan")
          hb = (2 * f + w) * 1024
          entries[f"{leg}|{key}"] = int(hb)
          us = mean([s["_ns"] for s in fetch[key]]) / 1e3
          extra = ""
          if key in sq and "mfma" in "".join(k for k in [key]) or ("conv1d" in key and key in sq):
              cyc = mean([s.get("GRBM_GUI_ACTIVE", 0.0) for s in sq[key]]) / 8.0
              busy = mean([s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for s in sq[key]])
              d_us = mean([s["_ns"] for s in sq[key]]) / 1e3
              if cyc > 0 and busy > 0:
                  extra = mfma_busy_columns(busy, cyc, d_us)
          lines.append(f"{key:64s} launches={fetch[key][0]['_n']:2d} us={us:8.1f} FETCH_SIZE={f:10.1f} WRITE_SIZE={w:10.1f} "
                       f"hbm_bytes={hb:12.0f}{extra}")
    open(os.path.join(PROF, f"{TAG}_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")

    import bench
    # (the GPU box has no .git: tools/collect_profiles.sh is started with ECG_COMMIT=<short hash> in its environment)
    commit = os.environ.get("ECG_COMMIT") or subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True,
                                                             text=True).stdout.strip()
    meta = {"commit": commit, "csrc_digest": bench.csrc_digest(), "source": f"profiles/{TAG}_pmc_summary.txt",
            "formula": "hbm bytes per call = (2 * FETCH_SIZE + WRITE_SIZE) KiB summed over the entry point's launches, separate "
                       "rocprofv3 --pmc passes of tools/pmc_step.py; gfx950 FETCH_SIZE correction per MI355X_MICROARCH.md"}
    json.dump({"_meta": meta, "entries": entries}, open(os.path.join(PROF, "pmc_traffic.json"), "w"), indent=1)
    return lines


if __name__ == "__main__":
    main()
