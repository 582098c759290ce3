#!/usr/bin/env python3
"""Per-layer microbenchmark of the conv entry points (the kernel-tuning harness).

    python tools/layer_bench.py [--batch 256] [--length 1000] [--reps 30] [--dtype f32|bf16] [--check]

For each of the four ConvBlock geometries (12->32->64->128->256, L halving) it times, with HIP events on the
launch stream, the forward (+ BN-statistics epilogue), the input gradient (row-padded dY, as the train step
calls it) and the weight gradient (MFMA kernel + slab reduce) — the fp32 entry points, or (--dtype bf16) the three of
the mixed-precision step on bf16 rows —, prints µs, TFLOP/s and the fraction of the fp32
(or bf16) MFMA peak, and the sum against the per-step budget.  --check compares every result with
torch.nn.functional.conv1d / autograd on the GPU (MIOpen; loose tolerance, it is only a tripwire — parity lives
in tests/).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    sys.path.insert(0, p)

GEOM = [(12, 32), (32, 64), (64, 128), (128, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--length", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16 = the entry points of the mixed-precision train step (bf16 rows on both sides: "
                         "ecg_conv1d_fwd_bf16_yh / ecg_conv1d_bwd_data_bf16hh / ecg_conv1d_bwd_weight_bias_bf16_ncl; block 0 reads "
                         "the fp32 network input)")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--blocks", default="0,1,2,3")
    ap.add_argument("--tag", default="")
    ap.add_argument("--ops", default="fwd,dgrad,wgrad", help="which entry points to time")
    a = ap.parse_args()
    import torch
    from ecg_hip import _lib as L, functional as F
    dev = torch.device("cuda", 0)
    L.load()
    L.call("ecg_check_device")
    call, q, f32, st = L.call, L.query, L.f32, L.stream
    N, K, pad = a.batch, 15, 7
    peak = 2500.0 if a.dtype == "bf16" else 157.3
    H = a.dtype == "bf16"
    rows, total = [], 0.0
    Lc = a.length
    g = torch.Generator(device="cpu").manual_seed(5)
    for b, (ci, co) in enumerate(GEOM):
        if str(b) not in a.blocks.split(","):
            Lc //= 2
            continue
        x = torch.randn(N, ci, Lc, generator=g).to(dev)
        w = (torch.randn(co, ci, K, generator=g) * 0.05).to(dev)
        bias = torch.randn(co, generator=g).to(dev)
        Lo = Lc
        flops = 2.0 * N * co * ci * K * Lo
        need_dx = b > 0
        dw, db = torch.empty_like(w), torch.empty_like(bias)
        if H:
            wf, wb = F.conv1d_pack_bf16(w, need_bwd=need_dx)
            ldh = (Lc + 7) & ~7
            ldt = q("ecg_conv1d_bf16_tk_dy_stride", Lo)
            xin_h = b > 0                      # block 0 reads the fp32 network input
            xh = torch.zeros(N, ci, ldh, dtype=torch.bfloat16, device=dev)
            xh[:, :, :Lc] = x.to(torch.bfloat16)
            yh = torch.empty(N, co, ldh, dtype=torch.bfloat16, device=dev)
            dyt = torch.zeros(N, co, ldt, dtype=torch.bfloat16, device=dev)
            dyt[:, :, :Lo] = torch.randn(N, co, Lo, generator=g).to(dev).to(torch.bfloat16)
            dxh = torch.empty(N, ci, ldh, dtype=torch.bfloat16, device=dev)
            ws = torch.empty(max(1, q("ecg_conv1d_bwd_weight_bf16_ncl_ws_floats", N, ci, co, Lc, K, pad)), device=dev)
        else:
            wf, wb = F.conv1d_pack(w, need_bwd=need_dx)
            ldy = q("ecg_conv1d_dy_row_stride", N, ci, co, Lc, K, pad, int(need_dx))
            dy = torch.zeros(N, co, ldy, device=dev)
            dy[:, :, :Lo] = torch.randn(N, co, Lo, generator=g).to(dev)
            y = torch.empty(N, co, Lo, device=dev)
            dx = torch.empty_like(x)
            ws = torch.empty(max(1, q("ecg_conv1d_bwd_weight_ws_floats", N, ci, co, Lc, K, pad)), device=dev)

        def fwd():
            if H:
                P = q("ecg_conv1d_fwd_bf16_yh_stat_partials", N, ci, co, Lc, K, pad, 1 if xin_h else 0, ldh, ldh)
                part = torch.empty(co * P * 2, device=dev)
                call("ecg_conv1d_fwd_bf16_yh", L.ptr(xh) if xin_h else f32(x), 1 if xin_h else 0, ldh, L.ptr(wf), f32(bias),
                     L.ptr(yh), ldh, f32(part), N, ci, co, Lc, K, pad, st())
            else:
                P = q("ecg_conv1d_fwd_stat_partials", N, ci, co, Lc, K, pad)
                part = torch.empty(co * P * 2, device=dev)
                call("ecg_conv1d_fwd", f32(x), f32(wf), f32(bias), f32(y), f32(part), N, ci, co, Lc, K, pad, st())

        def dgrad():
            if H:
                call("ecg_conv1d_bwd_data_bf16hh", L.ptr(dyt), ldt, L.ptr(wb), L.ptr(dxh), ldh, N, ci, co, Lc, K, pad, st())
            else:
                call("ecg_conv1d_bwd_data_ld", f32(dy), ldy, f32(wb), f32(dx), N, ci, co, Lc, K, pad, st())

        def wgrad():
            if H:
                call("ecg_conv1d_bwd_weight_bias_bf16_ncl", L.ptr(dyt), ldt, L.ptr(xh) if xin_h else f32(x), 1 if xin_h else 0,
                     ldh if xin_h else Lc, f32(dw), f32(db), f32(ws), N, ci, co, Lc, K, pad, st())
            else:
                call("ecg_conv1d_bwd_weight_bias_ld", f32(dy), ldy, f32(x), f32(dw), f32(db), f32(ws), N, ci, co, Lc, K, pad, st())

        ops = [("fwd", fwd)] + ([("dgrad", dgrad)] if need_dx else []) + [("wgrad", wgrad)]
        ops = [(n, f) for n, f in ops if n in a.ops.split(",")]
        for name, fn in ops:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            evs = []
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
            med = ts[len(ts) // 2]
            total += med
            rows.append({"block": b, "op": name, "us": round(med, 1), "min_us": round(ts[0], 1),
                         "tflops": round(flops / med / 1e6, 1), "frac": round(flops / med / 1e6 / peak, 3)})
        if a.check and H:
            xr = (xh[:, :, :Lc] if xin_h else x.to(torch.bfloat16)).float().requires_grad_(True)
            wr = w.to(torch.bfloat16).float().requires_grad_(True)
            yr = torch.nn.functional.conv1d(xr, wr, bias, padding=pad)
            yr.backward(dyt[:, :, :Lo].float())
            errs = {"y": (yh[:, :, :Lo].float() - yr).abs().max().item() / max(1.0, yr.abs().max().item()),
                    "dw": (dw - wr.grad).abs().max().item() / max(1.0, wr.grad.abs().max().item())}
            if need_dx:
                errs["dx"] = (dxh[:, :, :Lc].float() - xr.grad).abs().max().item() / max(1.0, xr.grad.abs().max().item())
            rows.append({"block": b, "check": {k: float(f"{v:.2e}") for k, v in errs.items()},
                         "ok": errs["dw"] < 1e-4 and all(v < 1e-2 for v in errs.values())})
        if a.check and not H:
            xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            br = bias.clone().requires_grad_(True)
            yr = torch.nn.functional.conv1d(xr, wr, br, padding=pad)
            yr.backward(dy[:, :, :Lo].contiguous())
            tol = 2e-3
            errs = {"y": (y - yr).abs().max().item(), "dw": (dw - wr.grad).abs().max().item() / max(1.0, wr.grad.abs().max().item()),
                    "db": (db - br.grad).abs().max().item() / max(1.0, br.grad.abs().max().item())}
            if need_dx:
                errs["dx"] = (dx - xr.grad).abs().max().item()
            rows.append({"block": b, "check": {k: float(f"{v:.2e}") for k, v in errs.items()},
                         "ok": all(v < tol for v in errs.values())})
        Lc //= 2
    for r in rows:
        print(json.dumps(r))
    print(json.dumps({"tag": a.tag, "sum_us": round(total, 1), "dtype": a.dtype, "batch": N, "length": a.length}))


if __name__ == "__main__":
    main()
