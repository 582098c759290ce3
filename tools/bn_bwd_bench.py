#!/usr/bin/env python3
"""Both forms of the fused BatchNorm + ReLU + pool backward on the four block shapes of a step (fp32, B = 256 by default):
the reduction pass + dx pass (ecg_bn_relu_pool_bwd_ld) and the one-launch register-resident form
(ecg_bn_relu_pool_bwd_one_launch) where the shape qualifies.  µs per call, HIP events on the launch stream.

    python tools/bn_bwd_bench.py [batch] [length]        (ECG_HIP_LIB selects an A/B build)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    sys.path.insert(0, p)
from ecg_hip import _lib as L  # noqa: E402

L.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000


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
    return round(ts[len(ts) // 2], 1), round(ts[0], 1)


for C, Lo, gap in ((32, T, 0), (64, T // 2, 0), (128, T // 4, 0), (256, T // 8, 1)):
    y = torch.randn(N, C, Lo, device="cuda")
    g = torch.randn((N, C) if gap else (N, C, Lo // 2), device="cuda")
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.2
    mean = y.mean(dim=(0, 2))
    invstd = 1.0 / (y.var(dim=(0, 2), unbiased=False) + 1e-5).sqrt()
    ldy = (Lo + 63) // 64 * 64
    dy, dg, db = torch.empty(N, C, ldy, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ws = torch.empty(L.query("ecg_bn_relu_pool_bwd_ws_floats", N, C, Lo), device="cuda")
    two = timed(lambda: L.call("ecg_bn_relu_pool_gap_bwd_ld" if gap else "ecg_bn_relu_pool_bwd_ld", L.f32(y), L.f32(g), L.f32(gamma),
                               L.f32(beta), L.f32(mean), L.f32(invstd), L.f32(dy), ldy, L.f32(dg), L.f32(db), L.f32(ws), N, C, Lo, 1,
                               L.stream()))
    S = L.query("ecg_bn_relu_pool_bwd_one_launch_splits", N, C, Lo, ldy)
    one = None
    if S:
        cnt = torch.zeros(max(2, L.query("ecg_bn_relu_pool_bwd_one_launch_counter_uints", N, C, Lo, ldy)), dtype=torch.int32, device="cuda")
        one = timed(lambda: L.call("ecg_bn_relu_pool_bwd_one_launch", L.f32(y), L.f32(g), L.f32(gamma), L.f32(beta), L.f32(mean),
                                   L.f32(invstd), L.f32(dy), ldy, L.f32(dg), L.f32(db), L.ptr(cnt), N, C, Lo, 1, gap, -1, L.stream()))
        assert not bool(cnt.any())
    print(f"C={C:4d} L={Lo:5d} two-pass median/min {two} us   one-launch S={S}: {one}")
