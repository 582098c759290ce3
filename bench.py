#!/usr/bin/env python3
"""bench.py — ECG windows/s of the full train step (fwd + BCE + bwd + AdamW) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]/[3]): ECGMultimodal, 12x1000 fp32 synthetic
windows, batch 256 PER GPU (weak scaling), inputs resident in HBM, driven through the
reference's own loop API (src.training.loop_demo.train_one_epoch_demo) with the flat AdamW;
with N > 1 the gradient exchange is one RCCL all-reduce of the flat 3 MB gradient per step.
Rank 0 prints ONE JSON line.  After the timed region a second, event-instrumented pass gives
live per-kernel durations for the `roofline` object; at N=1 the CPU oracle ("port") is timed
on the host cores for `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_HBM_GBS = 8000.0
CONV_GEOM = [(12, 32), (32, 64), (64, 128), (128, 256)]


class ListLoader:
    """The loop API only needs iteration and len(loader.dataset)."""

    def __init__(self, batch, steps):
        self.batch, self.steps = batch, steps
        self.dataset = range(batch[0].shape[0] * steps)

    def __iter__(self):
        return iter([self.batch] * self.steps)

    def __len__(self):
        return self.steps


def conv_flops_per_window(T):
    """Algorithmic conv flops of one window: fwd, and the train step (fwd + wgrad + dgrad
    without block 0's dgrad) — SURVEY §8(d)."""
    L, fwd, step = T, 0.0, 0.0
    for i, (ci, co) in enumerate(CONV_GEOM):
        f = 2.0 * co * ci * 15 * L
        fwd += f
        step += f * (3 if i > 0 else 2)
        L //= 2
    return fwd, step


SINGLE_LAUNCH = ("ecg_conv1d_fwd", "ecg_conv1d_bwd_data", "ecg_conv1d_bwd_data_ld",   # entry points that are exactly one kernel
                 "ecg_conv1d_fwd_bf16", "ecg_conv1d_bwd_data_bf16")


def load_pmc_traffic():
    """HBM bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
    committed under profiles/); keyed like the kernel names below.  None when not collected."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return {}


def kernel_roofline(timings, B):
    """Price the dominant single-kernel entry point of the instrumented pass against its roof.
    (ecg_conv1d_bwd_weight_bias is two launches — MFMA kernel + slab reduce — so it is listed in
    the breakdown but not used for the per-kernel roofline.)"""
    total_ms = 0.0
    per = {}
    for (name, sig), ms in timings.items():
        per[(name, sig)] = (sum(ms) / len(ms), len(ms))
        total_ms += sum(ms)
    agg = sorted(per.items(), key=lambda kv: -kv[1][0] * kv[1][1])
    (name, sig), (avg_ms, cnt) = next(kv for kv in agg if kv[0][0] in SINGLE_LAUNCH)
    key = f"{name}{list(sig)}"
    N, ci, co, Lc, K, pad = sig[-6:]
    flops = 2.0 * N * co * ci * K * (Lc + 2 * pad - K + 1)
    bytes_ = 4.0 * N * Lc * (ci + co) + 4.0 * co * ci * K
    ach = flops / (avg_ms * 1e-3) / 1e12
    peak = 2500.0 if name.endswith("_bf16") else PEAK_F32_TFLOPS        # dense bf16 MFMA peak ~2.5 PFLOP/s
    out = {"kernel": key, "avg_ms": round(avg_ms, 4), "bound": "mfma", "achieved": round(ach, 3),
           "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
           "traffic": load_pmc_traffic().get(key),
           "algorithmic_flops": flops, "algorithmic_bytes": bytes_,
           "hbm_frac_of_algorithmic_bytes": round(bytes_ / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
           "note": "fp32 conv: arithmetic intensity 65-615 flop/B vs ridge 20 -> the fp32 MFMA peak binds, "
                   "not HBM; peak = 157.3 TFLOP/s (v_mfma_f32_32x32x2_f32 == fp32 vector rate)"}
    breakdown = [{"kernel": f"{n}{list(s)}", "avg_ms": round(a, 4), "calls": c} for (n, s), (a, c) in agg[:14]]
    return out, breakdown, total_ms


def cpu_baseline(B, T, seconds, demo, labels):
    """The oracle (stock-torch CPU restatement of the reference loop body) on the host cores."""
    from oracle import ref_models as R
    R.seed_all(42)
    model = (R.RefECGMultimodal(num_labels=labels) if demo else R.RefECGCNN(num_labels=labels)).train()
    opt = R.make_adamw(model, 1e-4, 1e-4)
    batch = R.synthetic_batch(B, T, labels, demo=demo)
    R.train_step(model, opt, batch)                     # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        R.train_step(model, opt, batch)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 200:
            break
    return {"value": round(B * n / dt, 1), "unit": "windows/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/ref_models.py train_step (stock torch CPU ops), {'ECGMultimodal' if demo else f'ECGCNN({labels})'} "
                      f"B={B} 12x{T}, {n} steps in {dt:.1f}s"}


def input_pipeline_bench(B0, lengths, iters, cpu_seconds):
    """--workload input: the step before the model, WFDB int16 samples resident in
    HBM -> z-scored fp32 windows.  One JSON line per window length: windows/s of ecg_wfdb16_zscore, its time
    against the 6 B/sample algorithmic HBM traffic, the PCIe-inclusive rate of the packed loader, and the CPU
    baseline (oracle/input_oracle.py = the reference's numpy arithmetic) on a bounded sample."""
    import tempfile
    import numpy as np
    from ecg_hip import _lib, functional as F, pack
    _lib.call("ecg_check_device")
    B = B0
    for T in lengths:
        rng = np.random.default_rng(1234)
        d = rng.integers(-3000, 3000, size=(B, T, 12)).astype(np.int16)
        gain, base = np.full((B, 12), 1000.0), np.zeros((B, 12), np.int32)
        dd, dg, db = torch.from_numpy(d).cuda(), torch.from_numpy(gain).cuda(), torch.from_numpy(base).cuda()
        for _ in range(5):
            F.wfdb16_to_windows(dd, dg, db)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            F.wfdb16_to_windows(dd, dg, db)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        with _lib.kernel_timing() as kt:
            for _ in range(10):
                F.wfdb16_to_windows(dd, dg, db)
        per = {name: float(np.mean(ms_list)) for (name, _sig), ms_list in kt.result.items()}
        samples = B * T * 12
        out = {"metric": "input_windows_per_s", "value": round(B / (ms * 1e-3), 1), "unit": "windows/s",
               "config": {"workload": f"wfdb16 -> z-scored fp32, B={B}, 12x{T}"}, "ms_per_batch": round(ms, 4),
               "dtype": "i16->f32", "data": "synthetic", "entry_point_ms": {k: round(v, 4) for k, v in per.items()}}
        # algorithmic HBM bytes: 2 B/sample in (int16) + 4 B/sample out (fp32), whatever the launch plan
        alg = 6.0 * samples
        t = per["ecg_wfdb16_zscore"]
        out["roofline"] = {"kernel": "ecg_wfdb16_zscore", "bound": "hbm", "achieved": round(alg / (t * 1e-3) / 1e9, 1),
                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(alg / (t * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                           "algorithmic_bytes": alg, "traffic": None,
                           "note": "bound by the left-to-right float32 chains (one lane per (window, lead) row) that "
                                   "make the result bit-identical to the reference's numpy arithmetic, not by HBM"}
        # packed loader, PCIe inclusive
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "b.ecgpack")
            n = B * 8
            pack.write_pack(path, np.tile(d, (8, 1, 1)), np.tile(gain, (8, 1)), np.tile(base, (8, 1)),
                            np.zeros((n, 5), np.float32), np.zeros((n, 5), np.float32))
            ld = pack.PackedBatchLoader(path, B, shuffle=True, seed=1)
            for _ in ld:
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cnt = 0
            for ep in range(3):
                ld.set_epoch(ep)
                for bt in ld:
                    cnt += bt[0].shape[0]
            torch.cuda.synchronize()
            out["loader_windows_per_s_pcie_inclusive"] = round(cnt / (time.perf_counter() - t0), 1)
        # CPU baseline: the reference's numpy arithmetic on a bounded sample (the only use of the oracle here)
        from oracle import input_oracle as io_ref
        t0, done = time.perf_counter(), 0
        while time.perf_counter() - t0 < cpu_seconds:
            io_ref.windows_from_wfdb16(d[:16], gain[:16], base[:16])
            done += 16
        out["cpu_baseline"] = {"value": round(done / (time.perf_counter() - t0), 1), "unit": "windows/s", "cores": 1,
                               "kind": "port", "sample": f"{done} windows of 12x{T} through oracle/input_oracle.py (numpy)"}
        print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["train", "input"], default="train",
                    help="train = the headline train step; input = the WFDB int16 -> z-scored window step (one GPU)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="windows per GPU")
    ap.add_argument("--length", type=int, default=1000)
    ap.add_argument("--model", choices=["multimodal", "cnn"], default="cnn",
                    help="cnn = ECGCNN (BASELINE.json configs[1], the configuration the metric is quoted on); "
                         "multimodal = ECGMultimodal with the FiLM fusion (configs[2]/[3]); the other one is timed too "
                         "and reported under 'also'")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurement of the other model")
    ap.add_argument("--labels", type=int, default=5, help="output labels (1 = the AF-binary shape of BASELINE config 5)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 (default, the parity path) or bf16 = opt-in mixed precision of BASELINE config 5: "
                         "bf16 conv operands in forward/input-grad/weight-grad, fp32 accumulate, fp32 everything else")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step as one captured hipGraph (ecg_hip.graph.GraphedTrainStep); "
                         "single GPU only; pays off when the step is host-bound (small batches)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.workload == "input":
        sys.path[:0] = [ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")]
        lengths = [args.length] if "--length" in sys.argv else [1000, 5000]
        return input_pipeline_bench(args.batch, lengths, args.steps, min(args.cpu_seconds, 10.0))

    from ecg_hip import _lib, ddp
    from ecg_hip import functional as hipF
    from ecg_hip.optim import FlatAdamW
    hipF.set_conv_precision("bf16" if args.dtype == "bf16" else "fp32")
    from src.models.ecg_cnn import ECGCNN
    from src.models.ecg_multimodal import ECGMultimodal
    from src.training.loop import train_one_epoch
    from src.training.loop_demo import train_one_epoch_demo
    from src.utils.seed import set_seed

    rank, world, local = ddp.init_distributed("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.load()
    _lib.call("ecg_check_device")

    B, T = args.batch, args.length

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def build(demo):
        set_seed(42)
        model = (ECGMultimodal(num_labels=args.labels) if demo else ECGCNN(num_labels=args.labels)).to(dev)
        if world > 1:
            ddp.broadcast_module_state(model, 0)
        opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
        g = torch.Generator().manual_seed(1234 + rank)            # each rank its own shard of the global batch
        x = torch.randn(B, 12, T, generator=g).to(dev)
        y = (torch.rand(B, args.labels, generator=g) < 0.3).float().to(dev)
        batch = (x, torch.rand(B, 5, generator=g).to(dev), y) if demo else (x, y)
        return model, opt, batch

    def timed(demo, model, opt, batch, steps, warmup):
        """W untimed steps, then exactly K steps between barrier+synchronize; max over ranks."""
        run = train_one_epoch_demo if demo else train_one_epoch
        if args.graph:
            if world > 1:
                raise SystemExit("--graph is single-GPU only")
            from ecg_hip.graph import GraphedTrainStep
            gstep = GraphedTrainStep(model, opt, batch)

            def run(_model, loader, _opt, _dev):          # same contract as the loop API: mean loss of the epoch
                for b in loader:
                    gstep(*b)
                return gstep.mean_loss_and_reset(len(loader))

        run(model, ListLoader(batch, warmup), opt, dev)
        barrier()
        t0 = time.perf_counter()
        last_loss = run(model, ListLoader(batch, steps), opt, dev)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = t.item()
        return elapsed, last_loss

    def workload_name(demo):
        return (f"{'ECGMultimodal (FiLM)' if demo else f'ECGCNN({args.labels})'} train step fwd+BCE+bwd+AdamW, "
                f"12x{T} fp32, batch {B}/GPU, global batch {B * world}")

    demo = args.model == "multimodal"
    model, opt, batch = build(demo)
    elapsed, last_loss = timed(demo, model, opt, batch, args.steps, args.warmup)

    # instrumented pass (HIP events around every ABI launch on the launch stream); always eager
    run_eager = train_one_epoch_demo if demo else train_one_epoch
    with _lib.kernel_timing() as kt:
        run_eager(model, ListLoader(batch, min(args.steps, 10)), opt, dev)
    roof, breakdown, instr_ms = kernel_roofline(kt.result, B)

    if rank == 0:
        value = world * B * args.steps / elapsed
        fwd_f, step_f = conv_flops_per_window(T)
        line = {
            "metric": f"ECG windows/s (train step) at 12x{T}, batch {B}", "value": round(value, 1), "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "bf16 conv operands (fwd, input-grad, weight-grad) / f32 accumulate, activations, BN, tail, optimizer",
            "data": "synthetic",
            "config": {"workload": workload_name(demo),
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "loop": "hipGraph replay of the whole step (GraphedTrainStep)" if args.graph else "src.training API + FlatAdamW",
                       "final_loss": round(float(last_loss), 6)},
            "step_conv_tflops": round(value * step_f / 1e12, 2),
            "step_frac_of_fp32_peak": round(value * step_f / 1e12 / (PEAK_F32_TFLOPS * world), 4),
            "roofline": roof, "kernel_breakdown": breakdown,
            "instrumented_ms_per_step": round(instr_ms / min(args.steps, 10), 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(B, T, args.cpu_seconds, demo, args.labels)

    # secondary line item: the other model of the path on the same shape (all ranks take part)
    also = None
    if not args.no_also and not args.graph:
        del model, opt, batch
        m2, o2, b2 = build(not demo)
        e2, _ = timed(not demo, m2, o2, b2, args.steps, min(args.warmup, 5))
        also = {"workload": workload_name(not demo), "value": round(world * B * args.steps / e2, 1),
                "unit": "windows/s", "ms_per_step": round(1e3 * e2 / args.steps, 4)}
    if rank == 0:
        if also:
            line["also"] = also
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
