#!/usr/bin/env python3
"""Where does a forward-conv workgroup spend its time?  Runs each block geometry through the diagnostic build
(make -C ptbxl-multimodal_amd/csrc STAMP=1; ECG_HIP_LIB=.../libecg_hip_stamp.so) whose kernel stamps s_memtime at
start / end of prologue / end of first chunk / end of main loop / end of epilogue, and prints per-phase medians
in microseconds (s_memtime ticks at the shader clock; s_memrealtime, 100 MHz, anchors it).  --bf16 stamps the
mixed-precision forward kernel, --long uses 12x5000 windows, --wgrad stamps the fp32 weight-gradient kernel (start / end of
prologue / end of first stage / end of the stage loop / slab written; stages per workgroup in slot 5).
The fast-FIR kernels (round 5) carry the same stamps; EXTRA='-DECG_FWD_FFA=0 -DECG_WG_FFA=0' on the make line stamps the direct form.
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("ECG_HIP_LIB", os.path.join(ROOT, "tools", "_build", "libecg_hip_stamp.so"))


def report(b, stamps, names, stages=False):
    import numpy as np
    s = stamps.cpu().numpy().reshape(-1, 8)
    order = np.nonzero(s[:, 0] != 0)[0]                 # blockIdx of every stamped workgroup
    s = s[order]
    rt0, rt1 = s[:, 7].astype(np.float64), s[:, 6].astype(np.float64)      # s_memrealtime, 100 MHz
    t = s[:, :5].astype(np.float64)
    dur_ticks = t[:, 4] - t[:, 0]
    clk = float(np.median(dur_ticks / np.maximum(rt1 - rt0, 1.0))) * 100e6     # s_memtime ticks per second
    ph = np.diff(t, axis=1)
    span_us = (rt1.max() - rt0.min()) / 100.0
    out = {"block": b, "workgroups": int(len(s)), "clock_GHz_est": round(clk / 1e9, 3), "kernel_span_us": round(span_us, 1),
           "start_spread_us": round((rt0.max() - rt0.min()) / 100.0, 2),
           "end_spread_us": round((rt1.max() - rt1.min()) / 100.0, 2),
           "median_us": {k: round(float(np.median(ph[:, i])) / clk * 1e6, 2) for i, k in
                         enumerate(names)},
           "p90_us": {k: round(float(np.percentile(ph[:, i], 90)) / clk * 1e6, 2) for i, k in
                      enumerate(names)},
           "wg_total_median_us": round(float(np.median(dur_ticks)) / clk * 1e6, 2)}
    if stages:
        nst = (s[:, 5] & 0xFFFF).astype(np.float64)
        out["stages_per_workgroup_median"] = int(np.median(nst))
        out["us_per_stage_median"] = round(float(np.median(ph[:, 2] / np.maximum(nst - 1, 1))) / clk * 1e6, 3)
        # which workgroups share a CU (HW_ID bits: cu 8-11, sh 12, se 13-15 (+ XCC)), and does launch order decide who is
        # favoured?  end time of the earlier-numbered / later-numbered workgroup of each co-resident pair
        hw = (s[:, 5] >> 16).astype(np.int64)
        cu_key = ((hw >> 8) & 0xFF) | (((s[:, 5] >> 48) & 0xF).astype(np.int64) << 8)
        end_us = (rt1 - rt0.min()) / 100.0
        first, second, gaps = [], [], []
        for key in np.unique(cu_key):
            idx = np.nonzero(cu_key == key)[0]
            if len(idx) == 2:
                a, c = (idx[0], idx[1]) if order[idx[0]] < order[idx[1]] else (idx[1], idx[0])
                first.append(end_us[a]); second.append(end_us[c]); gaps.append(int(order[c] - order[a]))
        if first:
            first, second = np.array(first), np.array(second)
            out["cu_pairs"] = {"pairs": int(len(first)), "end_us_lower_blockidx_median": round(float(np.median(first)), 1),
                               "end_us_higher_blockidx_median": round(float(np.median(second)), 1),
                               "lower_ends_first_share": round(float((first < second).mean()), 3),
                               "abs_gap_us_median": round(float(np.median(np.abs(first - second))), 1),
                               "blockidx_distance_counts": {str(k): int(v) for k, v in zip(*np.unique(gaps, return_counts=True))}}
    print(json.dumps(out))


def main():
    import numpy as np
    import torch
    bf16 = "--bf16" in sys.argv
    T = 5000 if "--long" in sys.argv else 1000
    from ecg_hip import _lib as L, functional as F
    lib = L.load()
    setter = lib.ecg_debug_set_stamp_buffer_bf16 if bf16 else lib.ecg_debug_set_stamp_buffer
    setter.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    N, K, pad = 256, 15, 7
    Lc = T
    for b, (ci, co) in enumerate([(12, 32), (32, 64), (64, 128), (128, 256)]):
        x = torch.randn(N, ci, Lc, device=dev)
        w = torch.randn(co, ci, K, device=dev) * 0.05
        bias = torch.randn(co, device=dev)
        wf, _ = (F.conv1d_pack_bf16 if bf16 else F.conv1d_pack)(w, need_bwd=False)
        if "--wgrad" in sys.argv:
            ldy = L.query("ecg_conv1d_dy_row_stride", N, ci, co, Lc, K, pad, 0)
            dy = torch.randn(N, co, ldy, device=dev)
            dy[:, :, Lc:] = 0
            dw, db = torch.empty(co, ci, K, device=dev), torch.empty(co, device=dev)
            ws = torch.empty(L.query("ecg_conv1d_bwd_weight_ws_floats", N, ci, co, Lc, K, pad), device=dev)
            stamps = torch.zeros(16384 * 8, dtype=torch.int64, device=dev)
            for rep in range(3):
                stamps.zero_()
                setter(stamps.data_ptr())
                L.call("ecg_conv1d_bwd_weight_bias_ld", L.f32(dy), ldy, L.f32(x), L.f32(dw), L.f32(db), L.f32(ws), N, ci, co,
                       Lc, K, pad, L.stream())
                torch.cuda.synchronize()
            report(b, stamps, ["prologue", "first_stage", "other_stages", "exchange+slab"], stages=True)
            Lc //= 2
            continue
        ldy = (Lc + 7) & ~7
        y = torch.empty(N, co, ldy, dtype=torch.bfloat16, device=dev) if bf16 else torch.empty(N, co, Lc, device=dev)
        P = (L.query("ecg_conv1d_fwd_bf16_yh_stat_partials", N, ci, co, Lc, K, pad, 0, 0, ldy) if bf16 else
             L.query("ecg_conv1d_fwd_stat_partials", N, ci, co, Lc, K, pad))
        part = torch.empty(co * P * 2, device=dev)
        stamps = torch.zeros(16384 * 8, dtype=torch.int64, device=dev)
        for rep in range(3):
            stamps.zero_()
            setter(stamps.data_ptr())
            if bf16:
                L.call("ecg_conv1d_fwd_bf16_yh", L.f32(x), 0, 0, L.ptr(wf), L.f32(bias), L.ptr(y), ldy, L.f32(part), N, ci, co, Lc, K, pad,
                       L.stream())
            else:
                L.call("ecg_conv1d_fwd", L.f32(x), L.f32(wf), L.f32(bias), L.f32(y), L.f32(part), N, ci, co, Lc, K, pad, L.stream())
            torch.cuda.synchronize()
        report(b, stamps, ["prologue", "first_chunk", "other_chunks", "epilogue"])
        Lc //= 2


if __name__ == "__main__":
    main()
