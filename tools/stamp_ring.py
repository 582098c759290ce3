#!/usr/bin/env python3
"""Where does a workgroup of the bf16 ring kernel (csrc/conv1d_bf16_ring.hip) spend its time?  Runs the forward and
input-gradient convs of BASELINE.json configs[4] (B=256, 12x5000; block 0 = the fp32-input forward only) through the diagnostic build
(make -C ptbxl-multimodal_amd/csrc STAMP=1) and prints per-phase medians in microseconds: prologue / the first tile's
taps / the first tile's epilogue / everything after (further tiles + statistics)."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("ECG_HIP_LIB", os.path.join(ROOT, "tools", "_build", "libecg_hip_stamp.so"))


def report(tag, stamps, flops):
    import numpy as np
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] != 0]
    rt0, rt1 = s[:, 7].astype(np.float64), s[:, 6].astype(np.float64)
    t = s[:, :5].astype(np.float64)
    clk = float(np.median((t[:, 4] - t[:, 0]) / np.maximum(rt1 - rt0, 1.0))) * 100e6
    ph = np.diff(t, axis=1)
    names = ["prologue", "tile0_taps", "tile0_epilogue", "rest"]
    span = (rt1.max() - rt0.min()) / 100.0
    print(json.dumps({"case": tag, "workgroups": int(len(s)), "tiles_per_wg": int(np.median(s[:, 5])),
                      "clock_GHz": round(clk / 1e9, 3), "kernel_span_us": round(span, 1),
                      "frac_of_bf16_peak": round(flops / (span * 1e-6) / 2.5e15, 3),
                      "median_us": {k: round(float(np.median(ph[:, i])) / clk * 1e6, 2) for i, k in enumerate(names)},
                      "p90_us": {k: round(float(np.percentile(ph[:, i], 90)) / clk * 1e6, 2) for i, k in enumerate(names)},
                      "start_spread_us": round((rt0.max() - rt0.min()) / 100.0, 2),
                      "end_spread_us": round((rt1.max() - rt1.min()) / 100.0, 2)}))


def main():
    import torch
    from ecg_hip import _lib as L, functional as F
    lib = L.load()
    setter = lib.ecg_debug_set_stamp_buffer_ring
    setter.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    N, K, pad = 256, 15, 7
    for b, (ci, co) in enumerate([(12, 32), (32, 64), (64, 128), (128, 256)]):
        Lc = 5000 >> b
        ld = (Lc + 7) & ~7
        PA = L.query("ecg_conv1d_bf16_tk_dy_stride", Lc)
        xh = torch.zeros(N, ci, ld, dtype=torch.bfloat16, device=dev)
        xh[:, :, :Lc] = torch.randn(N, ci, Lc, device=dev).to(torch.bfloat16)
        w = torch.randn(co, ci, K, device=dev) * 0.05
        bias = torch.randn(co, device=dev)
        wf, wb = F.conv1d_pack_bf16(w, need_bwd=True)
        yh = torch.empty(N, co, ld, dtype=torch.bfloat16, device=dev)
        dyh = torch.zeros(N, co, PA, dtype=torch.bfloat16, device=dev)
        dyh[:, :, :Lc] = torch.randn(N, co, Lc, device=dev).to(torch.bfloat16)
        dxh = torch.empty(N, ci, ld, dtype=torch.bfloat16, device=dev)
        x32 = torch.randn(N, ci, Lc, device=dev) if b == 0 else None      # block 0 reads the fp32 network input
        P = L.query("ecg_conv1d_fwd_bf16_yh_stat_partials", N, ci, co, Lc, K, pad, 0 if b == 0 else 1, ld, ld)
        part = torch.empty(co * P * 2, device=dev)
        stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
        flops = 2.0 * N * co * ci * K * Lc
        for name in (("fwd",) if b == 0 else ("fwd", "dgrad")):
            for rep in range(3):
                stamps.zero_()
                setter(stamps.data_ptr())
                if name == "fwd":
                    L.call("ecg_conv1d_fwd_bf16_yh", L.f32(x32) if b == 0 else L.ptr(xh), 0 if b == 0 else 1, ld, L.ptr(wf), L.f32(bias), L.ptr(yh), ld, L.f32(part), N, ci, co,
                           Lc, K, pad, L.stream())
                else:
                    L.call("ecg_conv1d_bwd_data_bf16hh", L.ptr(dyh), PA, L.ptr(wb), L.ptr(dxh), ld, N, ci, co, Lc, K, pad, L.stream())
                torch.cuda.synchronize()
            report(f"block{b}_{name}", stamps, flops)


if __name__ == "__main__":
    main()
