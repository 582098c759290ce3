"""Inference throughput of the path: eval-mode forward + sigmoid, one launch
per ConvBlock (conv + running-stat BatchNorm + ReLU + MaxPool folded into the conv epilogue).

    python tools/bench_eval.py [--batch 256] [--length 1000 5000] [--iters 50]

One JSON line per (model, window length): windows/s with inputs resident in HBM, per-entry-point times
from HIP events, and the conv forward's fraction of the fp32 MFMA peak.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")]
PEAK_F32_TFLOPS = 157.3


def conv_fwd_flops(T):
    f, L = 0.0, T
    for ci, co in ((12, 32), (32, 64), (64, 128), (128, 256)):
        f += 2.0 * co * ci * 15 * L
        L //= 2
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--length", type=int, nargs="*", default=[1000, 5000])
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    from ecg_hip import _lib
    from ecg_hip import functional as F
    from src.models.ecg_cnn import ECGCNN
    from src.models.ecg_multimodal import ECGMultimodal
    from src.utils.seed import set_seed
    _lib.call("ecg_check_device")
    B = a.batch
    for T in a.length:
        for name in ("ECGCNN(5)", "ECGMultimodal"):
            set_seed(42)
            model = (ECGCNN(num_labels=5) if name.startswith("ECGCNN") else ECGMultimodal()).cuda().eval()
            g = torch.Generator().manual_seed(1234)
            x = torch.randn(B, 12, T, generator=g).cuda()
            xd = torch.rand(B, 5, generator=g).cuda()
            args = (x,) if name.startswith("ECGCNN") else (x, xd)

            def step():
                with torch.no_grad():
                    return F.sigmoid(model(*args))           # reference: torch.sigmoid(logits), loop.py:63
            # prime by TIME as bench.py does (1 s: allocator, lazy code-object loading, sustained clocks) — five steps right
            # after process start left the first leg host-bound in one collection (0.91 ms per batch against 0.49)
            import time
            t_end = time.perf_counter() + 1.0
            while time.perf_counter() < t_end:
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                step()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            with _lib.kernel_timing() as kt:
                for _ in range(10):
                    step()
            per = sorted(((f"{n}{list(s)}", float(np.mean(v))) for (n, s), v in kt.result.items()), key=lambda kv: -kv[1])
            val = B / (ms * 1e-3)
            print(json.dumps({
                "metric": "inference_windows_per_s", "value": round(val, 1), "unit": "windows/s", "ms_per_batch": round(ms, 4),
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{name} eval forward + sigmoid, 12x{T} fp32, batch {B}, fused conv+BN+ReLU+pool per block"},
                "conv_frac_of_fp32_peak": round(val * conv_fwd_flops(T) / 1e12 / PEAK_F32_TFLOPS, 4),
                "entry_point_ms": {k: round(v, 4) for k, v in per[:8]}}), flush=True)


if __name__ == "__main__":
    main()
