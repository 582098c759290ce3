#!/usr/bin/env python3
"""What-if (A/B build `make VARIANT=xbn EXTRA=-DECG_XBN=1`): the fp32 forward conv and weight gradient of blocks 1-3 with the
PREVIOUS block's BatchNorm + ReLU + MaxPool(2) applied while the x tile is staged (the pooled activation p is never written),
against the product kernels reading a materialised p.  Prints µs per call and checks the results against the product path.

    ECG_HIP_LIB=tools/_build/libecg_hip_xbn.so python tools/xbn_whatif.py [--batch 256] [--length 1000]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("ECG_HIP_LIB", os.path.join(ROOT, "tools", "_build", "libecg_hip_xbn.so"))
from ecg_hip import _lib as L, functional as F  # noqa: E402


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--length", type=int, default=1000)
    a = ap.parse_args()
    lib = L.load()
    vp, i = ctypes.c_void_p, ctypes.c_int
    lib.ecg_whatif_conv1d_fwd_xbn.argtypes = [vp, i, vp, vp, vp, vp, vp, i, i, i, i, vp]
    lib.ecg_whatif_conv1d_wgrad_xbn.argtypes = [vp, i, vp, i, vp, vp, vp, vp, i, i, i, i, vp]
    dev, N, K, pad = torch.device("cuda", 0), a.batch, 15, 7
    g = torch.Generator().manual_seed(3)
    Lc = a.length // 2
    tot = {"fwd": 0.0, "fwd_xbn": 0.0, "wgrad": 0.0, "wgrad_xbn": 0.0, "bn_pass": 0.0, "finalize": 0.0}
    for ci, co in ((32, 64), (64, 128), (128, 256)):
        yprev = torch.randn(N, ci, 2 * Lc, generator=g).to(dev)
        gamma, beta = (torch.rand(ci, generator=g) + 0.5).to(dev), (torch.randn(ci, generator=g) * 0.3).to(dev)
        mean, var = yprev.mean(dim=(0, 2)), yprev.var(dim=(0, 2), unbiased=False)
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        tab = torch.stack([mean, invstd * gamma, beta], dim=1).contiguous()           # [ci][3]
        # the materialised p (what the product path reads), through the product BatchNorm + ReLU + pool pass
        p = torch.empty(N, ci, Lc, device=dev)
        L.call("ecg_bn_relu_pool_fwd", L.f32(yprev), L.f32(gamma), L.f32(beta), L.f32(mean), L.f32(invstd), L.f32(p), N, ci, 2 * Lc, L.stream())
        w = (torch.randn(co, ci, K, generator=g) * 0.05).to(dev)
        bias = torch.randn(co, generator=g).to(dev)
        wf, _ = F.conv1d_pack(w, need_bwd=False)
        P = L.query("ecg_conv1d_fwd_stat_partials", N, ci, co, Lc, K, pad)
        part, part2 = torch.empty(co * P * 2, device=dev), torch.empty(co * P * 2, device=dev)
        y, y2 = torch.empty(N, co, Lc, device=dev), torch.empty(N, co, Lc, device=dev)
        st = L.stream()
        fwd = lambda: L.call("ecg_conv1d_fwd", L.f32(p), L.f32(wf), L.f32(bias), L.f32(y), L.f32(part), N, ci, co, Lc, K, pad, st)   # noqa: E731
        fwdx = lambda: lib.ecg_whatif_conv1d_fwd_xbn(yprev.data_ptr(), 2 * Lc, tab.data_ptr(), wf.data_ptr(), bias.data_ptr(),   # noqa: E731
                                                     y2.data_ptr(), part2.data_ptr(), N, ci, co, Lc, st)
        ldy = L.query("ecg_conv1d_dy_row_stride", N, ci, co, Lc, K, pad, 1)
        dy = torch.zeros(N, co, ldy, device=dev)
        dy[:, :, :Lc] = torch.randn(N, co, Lc, generator=g).to(dev)
        ws = torch.empty(max(1, L.query("ecg_conv1d_bwd_weight_ws_floats", N, ci, co, Lc, K, pad)), device=dev)
        dw, db, dw2, db2 = torch.empty_like(w), torch.empty_like(bias), torch.empty_like(w), torch.empty_like(bias)
        wg = lambda: L.call("ecg_conv1d_bwd_weight_bias_ld", L.f32(dy), ldy, L.f32(p), L.f32(dw), L.f32(db), L.f32(ws), N, ci, co, Lc, K, pad, st)   # noqa: E731
        wgx = lambda: lib.ecg_whatif_conv1d_wgrad_xbn(dy.data_ptr(), ldy, yprev.data_ptr(), 2 * Lc, tab.data_ptr(), dw2.data_ptr(),   # noqa: E731
                                                      db2.data_ptr(), ws.data_ptr(), N, ci, co, Lc, st)
        # the pass that would disappear, and the finalize launch that would replace the folded statistics combine
        Pp = L.query("ecg_bn_stat_partials_count", N, ci, 2 * Lc)
        sp = torch.empty(ci * Pp * 2, device=dev)
        L.call("ecg_bn_stat_partials", L.f32(yprev), L.f32(sp), N, ci, 2 * Lc, st)
        m2, i2 = torch.empty(ci, device=dev), torch.empty(ci, device=dev)
        bnp = lambda: L.call("ecg_bn_stats_relu_pool_fwd", L.f32(sp), Pp, N * 2 * Lc, None, None, None, 0.1, 1e-5, L.f32(yprev), L.f32(gamma),   # noqa: E731
                             L.f32(beta), L.f32(m2), L.f32(i2), L.f32(p), None, N, ci, 2 * Lc, 0, 0, 0, st)
        fin = lambda: L.call("ecg_bn_finalize", L.f32(sp), Pp, N * 2 * Lc, L.f32(m2), L.f32(i2), None, None, None, ci, 0.1, 1e-5, st)   # noqa: E731
        r = {"fwd": timed(fwd), "fwd_xbn": timed(fwdx), "wgrad": timed(wg), "wgrad_xbn": timed(wgx), "bn_pass": timed(bnp),
             "finalize": timed(fin)}
        torch.cuda.synchronize()
        ok = bool(torch.allclose(y, y2, atol=1e-5)) and bool(torch.allclose(dw, dw2, atol=1e-3, rtol=1e-4))
        print(f"C_in={ci} C_out={co} L={Lc} (x = pooled {2 * Lc}): {r}  results match: {ok}  max|dy|={float((y - y2).abs().max()):.2e}", flush=True)
        for k in tot:
            tot[k] += r[k]
        Lc //= 2
    print("sum over blocks 1-3:", {k: round(v, 1) for k, v in tot.items()})
    print("net per step = (fwd_xbn - fwd) + (wgrad_xbn - wgrad) - bn_pass + finalize =",
          round(tot["fwd_xbn"] - tot["fwd"] + tot["wgrad_xbn"] - tot["wgrad"] - tot["bn_pass"] + tot["finalize"], 1), "us")


if __name__ == "__main__":
    main()
