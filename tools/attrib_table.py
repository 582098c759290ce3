#!/usr/bin/env python3
"""Binding-resource attribution per entry point from the passes of tools/attrib_counters.sh.

    python tools/attrib_table.py r05 [--out profiles/r05_attribution_bf16_config5.txt] [--filter conv1d]

For every entry point of one train step (same keys as bench.py's `layers`), from separate rocprofv3 --pmc passes:
  us            dispatch time summed over the entry point's launches (counter pass A; profiled clocks run 2-3 % low)
  mfma          SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x us x clk): share of the time the matrix pipes are busy
  occ           SQ_WAVE_CYCLES x 4 / (1024 x us x clk): waves resident per SIMD, averaged over the dispatch
  issue|wait|stall   of the resident wave time: ACTIVE_INST_ANY | WAIT_ANY (parked at s_waitcnt / s_barrier) |
                WAIT_INST_ANY (ready but not issued: pipe busy / dependency); the three are disjoint and sum to ~1
  lds           SQ_LDS_IDX_ACTIVE / (256 CUs x us x clk): share of the time the LDS arrays are busy; `cf` = the share of
                those cycles that are bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)
  valu/mfma     SQ_INSTS_VALU / SQ_INSTS_MFMA (both per wave-instruction; VALU includes the MFMAs' own count if the
                counter does — printed raw), lds/mfma = SQ_INSTS_LDS / SQ_INSTS_MFMA
  TB/s          (2 x FETCH_SIZE + WRITE_SIZE) / us  (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section)
  L2hit         TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
clk = GRBM_GUI_ACTIVE / 8 / us, capped at 2.4 GHz (short dispatches: GUI_ACTIVE spans more than the kernel; then 2.1 GHz,
the clock in-kernel stamps read under these kernels, is used and the row is marked `~`).
`bound` is the verdict of the row: the resource with the largest busy share (mfma / lds / hbm at 6.3 TB/s achievable),
or `latency` when none reaches 0.5 — i.e. the waves are parked (wait) and nothing is saturated.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--out", default=None)
    ap.add_argument("--filter", default="")
    a = ap.parse_args()
    sys.argv = [sys.argv[0], a.tag]
    import make_profile_artifacts as M
    O = os.path.join(ROOT, "gpurun_out")
    passes = {}
    for P in "ABCDE":
        d = os.path.join(O, f"attrib_{a.tag}_{P}")
        if os.path.exists(d + ".order.json"):
            passes[P] = M.per_entry(d)
    if "A" not in passes:
        raise SystemExit("no pass A")

    def m(P, key, c):
        v = passes.get(P, {}).get(key)
        if not v:
            return float("nan")
        return M.mean([s.get(c, 0.0) for s in v])

    lines = [l.rstrip() for l in __doc__.splitlines()[5:25]]
    lines = ["# " + l for l in lines]
    lines.insert(0, f"# per entry point of one train step, from the five rocprofv3 --pmc passes of tools/attrib_counters.sh {a.tag} "
                    "(tools/attrib_table.py); columns:")
    hdr = (f"{'entry point':72s} {'us':>7s} {'clk':>5s} {'mfma':>5s} {'occ':>4s} {'issue':>5s} {'wait':>5s} {'stall':>5s} "
           f"{'lds':>5s} {'cf':>4s} {'valu/mfma':>9s} {'lds/mfma':>8s} {'TB/s':>5s} {'L2hit':>5s}  bound")
    lines.append(hdr)
    for key in passes["A"]:
        if a.filter and a.filter not in key:
            continue
        us = M.mean([s["_ns"] for s in passes["A"][key]]) / 1e3
        clk = m("A", key, "GRBM_GUI_ACTIVE") / 8.0 / (us * 1e3)
        mark = " "
        if not (0.5 < clk <= 2.4):
            clk, mark = 2.1, "~"
        cyc = us * 1e3 * clk
        mfma = m("A", key, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cyc)
        wc = m("A", key, "SQ_WAVE_CYCLES")
        occ = wc * 4 / (1024 * cyc)
        issue, wait, stall = (m("A", key, c) / wc if wc else float("nan") for c in
                              ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"))
        usB = M.mean([s["_ns"] for s in passes["B"][key]]) / 1e3 if key in passes.get("B", {}) else float("nan")
        clkB = m("B", key, "GRBM_GUI_ACTIVE") / 8.0 / (usB * 1e3) if usB == usB else float("nan")
        if not (0.5 < clkB <= 2.4):
            clkB = 2.1
        idx = m("B", key, "SQ_LDS_IDX_ACTIVE")
        lds = idx / (256 * usB * 1e3 * clkB)
        cf = m("B", key, "SQ_LDS_BANK_CONFLICT") / idx if idx else 0.0
        nm = m("C", key, "SQ_INSTS_MFMA")
        vpm = m("B", key, "SQ_INSTS_VALU") / nm if nm else float("nan")
        lpm = m("B", key, "SQ_INSTS_LDS") / nm if nm else float("nan")
        hit, miss = m("C", key, "TCC_HIT_sum"), m("C", key, "TCC_MISS_sum")
        l2 = hit / (hit + miss) if hit + miss else float("nan")
        tbs = float("nan")
        if "D" in passes and "E" in passes and key in passes["D"] and key in passes["E"]:
            by = (2 * m("D", key, "FETCH_SIZE") + m("E", key, "WRITE_SIZE")) * 1024
            tbs = by / (us * 1e-6) / 1e12
        shares = {"mfma": mfma, "lds": lds, "hbm": (tbs / 6.3 if tbs == tbs else 0.0)}
        top = max(shares, key=lambda k: shares[k])
        bound = f"{top} {shares[top]:.2f}" if shares[top] >= 0.5 else f"latency (max {top} {shares[top]:.2f}, wait {wait:.2f})"

        def f(v, w, p):
            return f"{v:{w}.{p}f}" if v == v else " " * (w - 1) + "-"
        lines.append(f"{key[:72]:72s} {us:7.1f} {clk:4.2f}{mark} {f(mfma, 5, 3)} {f(occ, 4, 2)} {f(issue, 5, 2)} {f(wait, 5, 2)} "
                     f"{f(stall, 5, 2)} {f(lds, 5, 3)} {f(cf, 4, 2)} {f(vpm, 9, 2)} {f(lpm, 8, 2)} {f(tbs, 5, 2)} "
                     f"{f(l2, 5, 2)}  {bound}")
    text = "\n".join(lines) + "\n"
    if a.out:
        open(os.path.join(ROOT, a.out), "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
