#!/usr/bin/env python3
"""bench.py — ECG windows/s of the full train step (fwd + BCE + bwd + AdamW) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Primary workload (BASELINE.json configs[1], the configuration the metric is quoted on):
ECGCNN(5), 12x1000 fp32 synthetic windows, batch 256 PER GPU (weak scaling), inputs resident in HBM,
driven through the reference's own loop API (src.training.loop.train_one_epoch) with the flat AdamW;
with N > 1 the gradient exchange is one RCCL all-reduce of the flat 2.9 MB gradient per step (two
buckets issued from backward hooks, or one all-reduce in step(): calibrated on the node, `rccl.exchange`).
Rank 0 prints ONE JSON line.

Protocol per leg: `--priming` untimed steps (allocator, lazy module loading, clock ramp), W untimed
warm-up steps, then EXACTLY K steps between barrier + synchronize (max over ranks) -> `value`,
`ms_per_step` (nothing but the steps is in that region); the same K steps once more with one HIP event per step
boundary on the launch stream -> `step_ms` median / p10 / p90.
A third, event-instrumented pass (events around every ABI launch) prices every conv entry point of
every layer against its roof -> `roofline` (the dominant entry point) and `layers`.

`also` carries the other configurations of BASELINE.json, each measured the same way with its own
roofline: ECGMultimodal (configs[2]/[3]), the headline with the stock torch.optim.AdamW the reference
scripts construct (scripts/03_train_ecg_baseline.py:130-133), and configs[4] (ECGCNN(1), 12x5000,
batch 256) in fp32 and in the opt-in bf16 mode.  The CPU baseline (oracle/ref_models.py, stock torch
CPU ops: "port") runs LAST so that its thread pool cannot disturb a GPU leg.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0
CONV_GEOM = [(12, 32), (32, 64), (64, 128), (128, 256)]
FWD, DGRAD, WGRAD = "fwd", "dgrad", "wgrad"
CONV_ENTRY = {"ecg_conv1d_fwd": (FWD, "f32"),
              "ecg_conv1d_bwd_data": (DGRAD, "f32"), "ecg_conv1d_bwd_data_ld": (DGRAD, "f32"),
              "ecg_conv1d_bwd_weight_bias": (WGRAD, "f32"), "ecg_conv1d_bwd_weight_bias_ld": (WGRAD, "f32"),
              "ecg_conv1d_fwd_bf16_yh": (FWD, "bf16"), "ecg_conv1d_bwd_data_bf16hh": (DGRAD, "bf16"),
              "ecg_conv1d_bwd_weight_bias_bf16_ncl": (WGRAD, "bf16")}
# bytes per element of the two activation operands an entry point streams (reduction-side tensor, result-side tensor);
# everything not listed reads and writes fp32.  ecg_conv1d_fwd_bf16_yh reads fp32 only for the network input (x_bf16 = 0).
IO_BYTES = {"ecg_conv1d_bwd_data_bf16hh": (2, 2), "ecg_conv1d_bwd_weight_bias_bf16_ncl": (2, 2)}


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves (fresh children, before anything touches the GPU)
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    """Environment of one self-launched rank (the torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts of this pool only support dmabuf IPC; with the legacy mode RCCL's (and
    torch's) cross-process device-memory handles fail with `hipIpcGetMemHandle: invalid argument`.  The image exports
    it already — setdefault keeps an explicit choice of the caller."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or world) // world)))
    return env


def _tail(path, n=20):
    try:
        with open(path, "rb") as f:
            return b"\n".join(f.read()[-16384:].splitlines()[-n:]).decode("utf-8", "replace")
    except OSError:
        return ""


def self_launch(n, argv=None, poll_s=0.2, grace_s=15.0):
    """`python bench.py --gpus N` with WORLD_SIZE unset: one child process per rank (never an exec of this process).
    Rank 0's stdout (the JSON line) is forwarded; every rank's stderr goes to its own file and the last 20 lines of any
    rank that fails are echoed.  ALL children are polled: the first non-zero exit ends the others (by handle) after a
    short grace instead of leaving them in a collective until the RCCL timeout.  Exit code = the first failure's."""
    import tempfile
    import threading
    argv = sys.argv[1:] if argv is None else argv
    port = _free_port()
    logdir = tempfile.mkdtemp(prefix="ecg_bench_ranks_")
    procs, logs = [], []
    for r in range(n):
        logs.append(os.path.join(logdir, f"rank{r}.stderr"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=rank_env(r, n, port),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=open(logs[-1], "wb")))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    first_bad, t_bad = None, None
    while True:
        rcs = [p.poll() for p in procs]
        if first_bad is None:
            bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                first_bad, t_bad = bad[0], time.time()
        if all(rc is not None for rc in rcs):
            break
        if first_bad is not None and time.time() - t_bad > grace_s:
            for p in procs:
                if p.poll() is None:
                    p.kill()                           # the exact children we started, by handle
        time.sleep(poll_s)
    reader.join(timeout=5.0)
    sys.stdout.write((out0[0] if out0 else b"").decode("utf-8", "replace"))
    sys.stdout.flush()
    rcs = [p.returncode for p in procs]
    if any(rcs):
        sys.stderr.write(f"bench.py: rank exit codes {rcs}; first failure: rank {first_bad}; stderr files in {logdir}\n")
        order = [first_bad] + [i for i in range(n) if i != first_bad and rcs[i] != 0]
        for i in order[:3]:
            sys.stderr.write(f"---- rank {i} (rc {rcs[i]}), last lines of stderr ----\n{_tail(logs[i])}\n")
        return abs(rcs[first_bad]) or 1
    return 0


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
class ListLoader:
    """The loop API only needs iteration and len(loader.dataset).  Records one HIP event on the current
    stream at every step boundary (the GPU is the bottleneck, so event intervals are per-step GPU time)."""

    def __init__(self, batch, steps, record=False):
        self.batch, self.steps, self.record = batch, steps, record
        self.dataset = range(batch[0].shape[0] * steps)
        self.events = []

    def __iter__(self):
        import torch
        for _ in range(self.steps):
            if self.record:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                self.events.append(e)
            yield self.batch
        if self.record:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.events.append(e)

    def __len__(self):
        return self.steps

    def step_ms(self):
        return [a.elapsed_time(b) for a, b in zip(self.events, self.events[1:])]


def percentiles(v):
    import numpy as np
    a = np.asarray(v, dtype=np.float64)
    return {"median": round(float(np.median(a)), 4), "p10": round(float(np.percentile(a, 10)), 4),
            "p90": round(float(np.percentile(a, 90)), 4), "min": round(float(a.min()), 4),
            "max": round(float(a.max()), 4), "n": int(a.size)}


def conv_flops_per_window(T):
    """Algorithmic conv flops of one window: fwd, and the train step (fwd + wgrad + dgrad without block 0's
    dgrad) — SURVEY §8(d)."""
    L, fwd, step = T, 0.0, 0.0
    for i, (ci, co) in enumerate(CONV_GEOM):
        f = 2.0 * co * ci * 15 * L
        fwd += f
        step += f * (3 if i > 0 else 2)
        L //= 2
    return fwd, step


def csrc_digest():
    """sha256 over the kernel sources: counter traffic collected for other kernels is stale."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ptbxl-multimodal_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


_PMC = None


def leg_tag(model, labels, dtype, length):
    """Name of a measured configuration in profiles/pmc_traffic.json: the same entry-point signature occurs in several legs
    (ECGCNN(5) and ECGMultimodal share every conv shape), so the counter traffic is keyed by leg AND entry."""
    return f"{'mm' if model == 'multimodal' else 'cnn' + str(labels)}_{dtype}_{length}"


def pmc_traffic(key):
    """(HBM bytes per launch, provenance) from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json:
    2*FETCH_SIZE + WRITE_SIZE per the guide's gfx950 correction), or (None, reason) when the file is missing, has
    no entry, or was collected for different kernel sources."""
    global _PMC
    if _PMC is None:
        try:
            _PMC = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except (OSError, ValueError):
            _PMC = {}
    meta = _PMC.get("_meta", {})
    if key not in _PMC.get("entries", {}):
        return None, "not collected"          # (key = "<leg>|<entry>[signature]")
    if meta.get("csrc_digest") != csrc_digest():
        return None, f"stale: collected for csrc {meta.get('csrc_digest')} at {meta.get('commit')}"
    return _PMC["entries"][key], f"profiles/pmc_traffic.json ({meta.get('source')}, commit {meta.get('commit')})"


OTHER_ROWS = []          # the non-conv entry points of the last layer_table() call (BatchNorm / pool passes, tail, loss, optimizer, packs)


def layer_table(timings, leg=""):
    """Instrumented pass -> one row per (conv entry point, layer): live µs per call (all launches of the entry
    point, e.g. weight-gradient MFMA kernel + slab reduce), TFLOP/s, fraction of the MFMA peak for its operand
    type, algorithmic bytes and, when a matching counter collection is committed, HBM traffic / algorithmic."""
    rows, other_ms, n_steps = [], 0.0, None
    OTHER_ROWS.clear()
    for (name, sig), ms in timings.items():
        if name not in CONV_ENTRY:
            other_ms += sum(ms)
            OTHER_ROWS.append({"entry": f"{name}{list(sig)}", "calls": len(ms), "avg_us": round(sum(ms) / len(ms) * 1e3, 2),
                               "group": "bn_pool" if name.startswith("ecg_bn_") else "rest"})
            continue
        op, dt = CONV_ENTRY[name]
        N, ci, co, Lc, K, pad = sig[-6:]
        Lo = Lc + 2 * pad - K + 1
        flops = 2.0 * N * co * ci * K * Lo
        bx, by = IO_BYTES.get(name, (4, 4))
        if name == "ecg_conv1d_fwd_bf16_yh":
            bx, by = (2 if sig[0] else 4), 2
        if name == "ecg_conv1d_bwd_weight_bias_bf16_ncl":        # ints: ldy, x_is_bf16, ldx, N, ... (block 0 reads the fp32 input)
            bx, by = (2 if sig[1] else 4), 2
        abytes = float(N) * (bx * Lc * ci + by * Lo * co) + 4.0 * co * ci * K
        avg = sum(ms) / len(ms)
        peak = PEAK_BF16_TFLOPS if dt == "bf16" else PEAK_F32_TFLOPS
        ach = flops / (avg * 1e-3) / 1e12
        # `flops` is the convolution's arithmetic as the reference defines it (2 K multiplies per output pair and channel pair);
        # the fp32 fast-FIR kernels issue 23 of those 30 (include/ecg_hip.h: ecg_conv1d_multiplies_per_output_pair), so the
        # algorithmic rate `frac` can exceed what the matrix pipe does: `pipe_frac` is the issued share
        mult = 2 * K
        if dt == "f32":
            from ecg_hip import _lib
            mult = _lib.query("ecg_conv1d_multiplies_per_output_pair", {FWD: 0, DGRAD: 1, WGRAD: 2}[op], ci, co, K, pad)
        issued = flops * mult / (2.0 * K)
        key = f"{name}{list(sig)}"
        tr, prov = pmc_traffic(f"{leg}|{key}")
        prov_short = (f"pmc@{_PMC.get('_meta', {}).get('commit')}" if tr else prov.split(":")[0])
        rows.append({"entry": key, "op": op, "operands": dt, "c_in": ci, "c_out": co, "L": Lc, "calls": len(ms),
                     "avg_us": round(avg * 1e3, 2), "tflops": round(ach, 2), "peak": peak, "frac": round(ach / peak, 4),
                     "algorithmic_flops": flops, "algorithmic_bytes": abytes,
                     "multiplies_per_output_pair": mult, "pipe_frac": round(issued / (avg * 1e-3) / 1e12 / peak, 4),
                     "traffic_bytes_from_profile": tr, "traffic_over_algorithmic": (round(tr / abytes, 3) if tr else None),
                     "traffic_source": prov, "traffic_source_short": prov_short})
    rows.sort(key=lambda r: (-r["avg_us"] * r["calls"]))
    return rows, other_ms


def roofline_of(rows):
    """The dominant conv entry point (largest total time; multi-launch entry points included).  Compact: the prose
    lives in DESIGN.md section 4 (fp32 conv: 65-615 flop/B against a ridge of 20 -> the fp32 MFMA peak binds, not HBM)."""
    if not rows:
        return None
    r = rows[0]
    return {"kernel": r["entry"], "op": r["op"], "avg_ms": round(r["avg_us"] / 1e3, 4), "bound": "mfma",
            "achieved": r["tflops"], "peak": r["peak"], "unit": "TFLOP/s", "frac": r["frac"],
            "multiplies_per_output_pair": r.get("multiplies_per_output_pair"), "pipe_frac": r.get("pipe_frac"),
            "frac_is": "algorithmic: the op's 30 multiplies per output pair; the fast-FIR kernels issue 23 (pipe_frac), so frac may pass 1",
            "traffic": r["traffic_bytes_from_profile"], "traffic_over_algorithmic": r["traffic_over_algorithmic"],
            "traffic_source": r["traffic_source_short"],
            "algorithmic_flops": r["algorithmic_flops"], "algorithmic_bytes": r["algorithmic_bytes"],
            "hbm_frac": round(r["algorithmic_bytes"] / (r["avg_us"] * 1e-6) / 1e9 / PEAK_HBM_GBS, 5)}


def frac_by_block(rows, key="frac"):
    """{"fwd": [block 0..3], "dgrad": [...], "wgrad": [...]}: fraction of the MFMA peak per conv entry point, ordered
    by block (C_in ascending); None where a block has no such entry point (block 0 has no input gradient).
    key = "frac": the convolution's algorithmic flops; "pipe_frac": the multiplies the kernel issues (fast-FIR: 23 of 30)."""
    cins = sorted({r["c_in"] for r in rows})
    out = {}
    for op in (FWD, DGRAD, WGRAD):
        by = {r["c_in"]: r.get(key) for r in rows if r["op"] == op}
        out[op] = [by.get(c) for c in cins]
    return out


def host_info():
    import torch
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return {"cpu_model": model, "physical_cores": len(cores) or None, "logical_cpus": os.cpu_count(),
            "torch_threads": torch.get_num_threads(), "torch": torch.__version__}


def cpu_baseline(seconds):
    """The oracle (oracle/ref_models.py: stock-torch CPU restatement of the reference loop body, kind "port") on
    the host cores.  Primary figure = BASELINE.json configs[0] / BASELINE.md §3: ECGCNN(5), batch 32, 12x1000;
    median of >= 30 steps.  The same model at batch 256 (the GPU leg's batch) rides along."""
    import numpy as np
    import torch
    from oracle import ref_models as R
    info = host_info()

    def run(B, T, labels, demo, budget, min_steps):
        R.seed_all(42)
        model = (R.RefECGMultimodal(num_labels=labels) if demo else R.RefECGCNN(num_labels=labels)).train()
        opt = R.make_adamw(model, 1e-4, 1e-4)
        batch = R.synthetic_batch(B, T, labels, demo=demo)
        for _ in range(2):
            R.train_step(model, opt, batch)
        ts, t_all = [], time.perf_counter()
        while len(ts) < min_steps or (time.perf_counter() - t_all < budget and len(ts) < 400):
            t0 = time.perf_counter()
            R.train_step(model, opt, batch)
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all > 3 * budget:
                break
        med = float(np.median(ts))
        return {"windows_per_s": round(B / med, 1), "median_ms": round(med * 1e3, 3), "steps": len(ts),
                "p10_ms": round(float(np.percentile(ts, 10)) * 1e3, 3), "p90_ms": round(float(np.percentile(ts, 90)) * 1e3, 3)}

    # torch's default is one thread per logical core pair (128 here); on 12x1000 windows at batch 32 that
    # oversubscribes oneDNN, so the baseline is the BEST of a short thread-count sweep (stated in the line)
    default_threads = torch.get_num_threads()
    sweep = {}
    for th in sorted({default_threads, 32, 16, 8}):
        if th > default_threads:
            continue
        torch.set_num_threads(th)
        sweep[th] = run(32, 1000, 5, False, 0.12 * seconds, 30 if th == default_threads else 12)
    best = max(sweep, key=lambda t: sweep[t]["windows_per_s"])
    torch.set_num_threads(best)
    c1 = run(32, 1000, 5, False, 0.2 * seconds, 30)
    c256 = run(256, 1000, 5, False, 0.25 * seconds, 5)
    torch.set_num_threads(default_threads)
    return {"value": c1["windows_per_s"], "unit": "windows/s", "cores": best, "kind": "port", "threads": best,
            "sample": f"oracle/ref_models.py train_step, ECGCNN(5) B=32 12x1000 (configs[0]): median of {c1['steps']} steps, "
                      f"{c1['median_ms']} ms (p10 {c1['p10_ms']}, p90 {c1['p90_ms']}), best thread count of the sweep",
            "thread_sweep": {str(t): v["windows_per_s"] for t, v in sweep.items()},
            "batch256": {"value": c256["windows_per_s"], "median_ms": c256["median_ms"], "steps": c256["steps"]},
            **info}


# ------------------------------------------------------------------------------------------------
# input-pipeline workload (unchanged protocol; one GPU)
# ------------------------------------------------------------------------------------------------
def input_pipeline_bench(B0, lengths, iters, cpu_seconds):
    """--workload input: the step before the model, WFDB int16 samples resident in
    HBM -> z-scored fp32 windows.  One JSON line per window length: windows/s of ecg_wfdb16_zscore, its time
    against the 6 B/sample algorithmic HBM traffic, the PCIe-inclusive rate of the packed loader, and the CPU
    baseline (oracle/input_oracle.py = the reference's numpy arithmetic) on a bounded sample."""
    import tempfile
    import numpy as np
    import torch
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
        out["roofline"] = {"kernel": "ecg_wfdb16_zscore", "bound": "latency (sequential fp32 chain)",
                           "achieved": round(alg / (t * 1e-3) / 1e9, 1),
                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(alg / (t * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                           "algorithmic_bytes": alg, "traffic": None,
                           "note": "priced against the HBM roof for its 6 B/sample, but bound by the left-to-right float32 "
                                   "chains (one lane per (window, lead) row) that make the result bit-identical to the "
                                   "reference's numpy arithmetic"}
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
            ld.stage_seconds.update(gather=0.0, enqueue=0.0, slot_wait=0.0, batches=0)      # (the warm-up epoch: page faults, allocator)
            t0 = time.perf_counter()
            cnt = 0
            for ep in range(3):
                ld.set_epoch(ep)
                for bt in ld:
                    cnt += bt[0].shape[0]
            torch.cuda.synchronize()
            out["loader_windows_per_s_pcie_inclusive"] = round(cnt / (time.perf_counter() - t0), 1)
            # where a batch's time goes: host gather (thread pool, out of the mapped file into the pinned slot), the H2D copy
            # of its int16 samples alone (HIP events around one pinned -> device copy), the kernels (ms_per_batch above)
            st, nb = ld.stage_seconds, max(1, ld.stage_seconds["batches"])
            pinned = torch.empty((B, T, 12), dtype=torch.int16).pin_memory()
            dst = torch.empty((B, T, 12), dtype=torch.int16, device="cuda")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dst.copy_(pinned, non_blocking=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                dst.copy_(pinned, non_blocking=True)
            e1.record()
            torch.cuda.synchronize()
            h2d = e0.elapsed_time(e1) / 10
            out["loader_breakdown_ms_per_batch"] = {
                "host_gather": round(st["gather"] / nb * 1e3, 4), "gather_threads": ld.gather_threads, "slots": ld.n_slots,
                "slot_wait": round(st["slot_wait"] / nb * 1e3, 4), "enqueue": round(st["enqueue"] / nb * 1e3, 4),
                "h2d_int16": round(h2d, 4), "h2d_GBps": round(B * T * 12 * 2 / (h2d * 1e-3) / 1e9, 1), "kernels": round(ms, 4)}
            # the same epochs UNSHUFFLED: every batch is one contiguous run of the mapped file
            ld2 = pack.PackedBatchLoader(path, B, shuffle=False)
            for _ in ld2:
                pass
            torch.cuda.synchronize()
            t0, cnt = time.perf_counter(), 0
            for ep in range(3):
                for bt in ld2:
                    cnt += bt[0].shape[0]
            torch.cuda.synchronize()
            out["loader_windows_per_s_unshuffled"] = round(cnt / (time.perf_counter() - t0), 1)
        # CPU baseline: the reference's numpy arithmetic on a bounded sample (the only use of the oracle here)
        from oracle import input_oracle as io_ref
        t0, done = time.perf_counter(), 0
        while time.perf_counter() - t0 < cpu_seconds:
            io_ref.windows_from_wfdb16(d[:16], gain[:16], base[:16])
            done += 16
        out["cpu_baseline"] = {"value": round(done / (time.perf_counter() - t0), 1), "unit": "windows/s", "cores": 1,
                               "kind": "port", "sample": f"{done} windows of 12x{T} through oracle/input_oracle.py (numpy)"}
        print(json.dumps(out))


# ------------------------------------------------------------------------------------------------
# the printed line (compact: the driver keeps 8000 characters of stdout) and its side file
# ------------------------------------------------------------------------------------------------
LINE_BUDGET = 4000
_LEG_KEEP = ("workload", "value", "ms_per_step")


def _short_step_ms(p):
    return {k: p[k] for k in ("median", "p10", "p90", "n") if k in p}


def _short_cpu(c):
    keep = ("value", "unit", "cores", "kind", "threads", "sample", "cpu_model", "physical_cores", "logical_cpus", "torch")
    out = {k: c[k] for k in keep if k in c}
    if "batch256" in c:
        out["batch256_value"] = c["batch256"]["value"]
    return out


def _short_rccl(r, leg):
    out = {k: r[k] for k in ("backend", "ranks_seen_by_allreduce", "rel_diff", "flat_gradient_bytes") if k in r}
    if "exchange_exposed_ms_per_step" in leg:
        out["exchange_exposed_ms_per_step"] = _short_step_ms(leg["exchange_exposed_ms_per_step"])
    if "exchange" in leg:
        out["exchange"] = leg["exchange"]
    return out


def build_line(primary, also, cpu, rccl, *, n_gpus, steps, warmup, batch, length, priming, priming_seconds, n1_value=None,
               small=None):
    """(line, detail): the ONE JSON line rank 0 prints — at most LINE_BUDGET characters, everything the contract and
    the judge read (value, ms_per_step, config, step_ms, roofline, cpu_baseline, rccl, one summary row per other
    configuration) — and the full record (per-entry-point `layers` tables of every leg, the CPU thread sweep, the
    gradient checksums) that goes to a side file whose path the line carries."""
    line = {
        "metric": f"ECG windows/s (train step) at 12x{length}, batch {batch}", "value": primary["value"], "unit": "windows/s",
        "n_gpus": n_gpus, "steps": steps, "warmup": warmup, "ms_per_step": primary["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": primary["dtype_short"], "data": "synthetic",
        "config": {"workload": primary["workload"], "global_batch": batch * n_gpus, "parallelism": f"dp{n_gpus}",
                   "loop": primary["loop"], "optimizer": primary["optimizer"], "priming_steps": priming,
                   "priming_seconds": priming_seconds, "final_loss": primary["final_loss"]},
        "step_ms": _short_step_ms(primary["step_ms"]), "value_at_median_step": primary["value_at_median_step"],
        "step_conv_tflops": primary["step_conv_tflops"], "step_frac_of_mfma_peak": primary["step_frac_of_mfma_peak"],
        "roofline": primary["roofline"], "frac_by_block": primary["frac_by_block"],
        "pipe_frac_by_block": primary.get("pipe_frac_by_block"),
        "instrumented_ms_per_step": primary["instrumented_ms_per_step"],
    }
    if rccl is not None:
        line["rccl"] = _short_rccl(rccl, primary)
        if n1_value:
            line["efficiency_vs_n1"] = round(primary["value"] / (n_gpus * n1_value), 4)
    if also:
        line["also"] = []
        for leg in also:
            row = {k: leg[k] for k in _LEG_KEEP}
            row["workload"] = leg["workload_short"]
            row["frac"] = leg["roofline"]["frac"] if leg.get("roofline") else None
            row["step_frac"] = leg["step_frac_of_mfma_peak"]
            row["other_ms"] = leg["instrumented_ms_per_step"]["everything_else"]
            row["bn_pool_ms"] = leg["instrumented_ms_per_step"].get("bn_pool_passes")
            row["host_gap_ms"] = leg["instrumented_ms_per_step"].get("host_gap")
            if "exchange_exposed_ms_per_step" in leg:
                row["exchange_exposed_ms"] = leg["exchange_exposed_ms_per_step"]["median"]
            if "exchange" in leg:
                row["exchange_mode"] = leg["exchange"]["mode"]
            line["also"].append(row)
    if small:
        # the reference's own batch sizes through train_one_epoch: windows/s with the captured-graph loop and eagerly
        line["ref_batch_sizes"] = [{"workload": p["graph"]["workload_short"], "value": p["graph"]["value"],
                                    "ms_per_step": p["graph"]["ms_per_step"], "eager_value": p["eager"]["value"]} for p in small]
    if cpu is not None:
        line["cpu_baseline"] = _short_cpu(cpu)
    detail = {"line": dict(line), "primary": primary, "also": also, "ref_batch_sizes": small or [], "cpu_baseline": cpu,
              "rccl": rccl}
    # never let the line outgrow what the driver keeps: drop optional blocks, least important first
    for key in ("ref_batch_sizes", "pipe_frac_by_block", "frac_by_block", "instrumented_ms_per_step", "step_conv_tflops", "value_at_median_step", "also"):
        if len(json.dumps(line, separators=(",", ":"))) <= LINE_BUDGET - 80:
            break
        line.pop(key, None)
    return line, detail


# ------------------------------------------------------------------------------------------------
# train workload
# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["train", "input"], default="train",
                    help="train = the headline train step; input = the WFDB int16 -> z-scored window step (one GPU)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--priming", type=int, default=30,
                    help="untimed steps BEFORE the warm-up of every leg (allocator, lazy code-object loading, clock "
                         "ramp after idle); makes short --steps/--warmup runs reproduce long ones")
    ap.add_argument("--priming-seconds", type=float, default=1.5,
                    help="keep priming every leg until this much wall time has passed as well (sustained clocks instead "
                         "of the boost a cold chip shows for the first ~second; also gives an external busy probe "
                         "something to see)")
    ap.add_argument("--detail", default=None,
                    help="side file for the per-entry-point `layers` tables of every leg (default: "
                         "gpurun_out/bench_detail_n<N>.json; the printed line carries its path only)")
    ap.add_argument("--batch", type=int, default=256, help="windows per GPU")
    ap.add_argument("--length", type=int, default=1000)
    ap.add_argument("--model", choices=["multimodal", "cnn"], default="cnn",
                    help="cnn = ECGCNN (BASELINE.json configs[1], the configuration the metric is quoted on); "
                         "multimodal = ECGMultimodal with the FiLM fusion (configs[2]/[3])")
    ap.add_argument("--optim", choices=["flat", "torch"], default="flat",
                    help="flat = ecg_hip.optim.FlatAdamW (one launch); torch = the stock torch.optim.AdamW the reference "
                         "scripts construct")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary legs")
    ap.add_argument("--labels", type=int, default=5, help="output labels (1 = the AF-binary shape of BASELINE config 5)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 (default, the parity path) or bf16 = opt-in mixed precision of BASELINE config 5: "
                         "bf16 conv operands in forward/input-grad/weight-grad and bf16 storage of the tensors between the kernels of a "
                         "block chain (y, p, dp); fp32 accumulate, BatchNorm arithmetic, parameters, tail, optimizer")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step as one captured hipGraph (ecg_hip.graph.GraphedTrainStep); single GPU only")
    ap.add_argument("--cpu-seconds", type=float, default=14.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n1-value", type=float, default=None,
                    help="windows/s of the N=1 run: adds efficiency_vs_n1 = value / (N * n1-value) to an N>1 line")
    args = ap.parse_args()

    # ---- before ANY GPU call: a bare `--gpus N` starts its own ranks --------------------------------------
    if args.workload == "train" and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch

    if args.workload == "input":
        lengths = [args.length] if "--length" in sys.argv else [1000, 5000]
        return input_pipeline_bench(args.batch, lengths, args.steps, min(args.cpu_seconds, 10.0))

    from ecg_hip import _lib, ddp
    from ecg_hip import functional as hipF
    from ecg_hip.optim import FlatAdamW
    from src.models.ecg_cnn import ECGCNN
    from src.models.ecg_multimodal import ECGMultimodal
    from src.training.loop import train_one_epoch
    from src.training.loop_demo import train_one_epoch_demo
    from src.utils.seed import set_seed

    rank, world, local = ddp.init_distributed("nccl", timeout_s=float(os.environ.get("ECG_HIP_DIST_TIMEOUT_S", "180")))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rehearsal = os.environ.get("ECG_HIP_REHEARSE_ON_ONE_GPU") == "1"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.load()
    _lib.call("ecg_check_device")
    dist = torch.distributed
    backend = dist.get_backend() if world > 1 else None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[local]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    def run_leg(spec):
        """One configuration, measured by the protocol in the module docstring.  Every rank takes part."""
        B, T, labels, demo = spec["batch"], spec["length"], spec["labels"], spec["model"] == "multimodal"
        bf16, graph, stock = spec["dtype"] == "bf16", spec.get("graph", False), spec.get("optim", "flat") == "torch"
        hipF.set_conv_precision("bf16" if bf16 else "fp32")
        set_seed(42)
        model = (ECGMultimodal(num_labels=labels) if demo else ECGCNN(num_labels=labels)).to(dev)
        if world > 1:
            ddp.broadcast_module_state(model, 0)
        wrapped = model
        if stock:
            if world > 1:
                wrapped = ddp.FlatGradDDP(model)
            opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
        else:
            opt = FlatAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
        g = torch.Generator().manual_seed(1234 + rank)            # each rank its own shard of the global batch
        x = torch.randn(B, 12, T, generator=g).to(dev)
        y = (torch.rand(B, labels, generator=g) < 0.3).float().to(dev)
        batch = (x, torch.rand(B, 5, generator=g).to(dev), y) if demo else (x, y)
        run_eager = train_one_epoch_demo if demo else train_one_epoch
        run = run_eager
        if graph:
            if world > 1 or stock:
                raise SystemExit("--graph is single-GPU, FlatAdamW only")
            from ecg_hip.graph import GraphedTrainStep
            gstep = GraphedTrainStep(model, opt, batch)

            def run(_model, loader, _opt, _dev):          # same contract as the loop API: mean loss of the epoch
                for b in loader:
                    gstep(*b)
                return gstep.mean_loss_and_reset(len(loader))

        if spec["priming"] > 0:
            run(wrapped, ListLoader(batch, spec["priming"]), opt, dev)
        if spec.get("priming_seconds", 0) > 0:
            # time-based part of the priming: the same on every rank (rank 0's clock decides), outside every timed region
            t_end, chunk = time.perf_counter() + spec["priming_seconds"], max(10, spec["priming"])
            while True:
                run(wrapped, ListLoader(batch, chunk), opt, dev)
                torch.cuda.synchronize()
                go = torch.tensor([1.0 if time.perf_counter() < t_end else 0.0], device=dev)
                if world > 1:
                    dist.broadcast(go, src=0)
                if go.item() == 0.0:
                    break
        exchange_mode = None
        if world > 1 and not stock:
            # Calibrate the exchange on THIS node (outside the timed region): hooked two-bucket all-reduces under backward
            # vs one all-reduce in step().  Every rank times both; the decision is taken on the max over ranks, which
            # the all-reduce makes identical everywhere.
            cal = opt.calibrate_overlap(lambda n: run(wrapped, ListLoader(batch, n), opt, dev), steps=10, warm=3)
            exchange_mode = {"mode": cal["mode"], "calibration_ms_per_step": cal["ms_per_step"]}
        run(wrapped, ListLoader(batch, spec["warmup"]), opt, dev)
        # The timed region carries NO per-step events: an event record between two steps is a barrier packet with a
        # timestamp and costs 2-7 us of idle front end per step (A/B on one box: 1.6514 / 1.6552 ms with, 1.6498 / 1.6482
        # without).  The per-step distribution comes from a second pass of the same K steps right after it.
        loader = ListLoader(batch, spec["steps"])
        barrier()
        t0 = time.perf_counter()
        last_loss = run(wrapped, loader, opt, dev)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = t.item()
        loader = ListLoader(batch, spec["steps"], record=True)
        run(wrapped, loader, opt, dev)
        torch.cuda.synchronize()
        step_ms = percentiles(loader.step_ms())

        # instrumented pass (HIP events around every ABI launch on the launch stream); always eager
        n_instr = 5
        exch = []
        if world > 1 and not stock:           # exposed exchange time: how long the compute stream waits in reduce_gradients
            inner = opt.reduce_gradients

            def timed_reduce():
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                out = inner()
                b.record()
                exch.append((a, b))
                return out
            opt.reduce_gradients = timed_reduce
        with _lib.kernel_timing() as kt:
            run_eager(wrapped, ListLoader(batch, n_instr), opt, dev)
        if exch:
            opt.reduce_gradients = inner
        rows, other_ms = layer_table(kt.result, leg_tag(spec["model"], labels, "bf16" if bf16 else "f32", T))
        # did the loop API replay a captured step (ecg_hip.graph.LoopStepper: FlatAdamW, one rank, no hooks)?
        loop_replays = (not graph) and any(st.graphs for st in getattr(opt, "_ecg_loop_steppers", {}).values())
        conv_ms = sum(r["avg_us"] * r["calls"] for r in rows) / 1e3

        value = world * B * spec["steps"] / elapsed
        fwd_f, step_f = conv_flops_per_window(T)
        peak = PEAK_BF16_TFLOPS if bf16 else PEAK_F32_TFLOPS
        passes = sorted(OTHER_ROWS, key=lambda r: -r["avg_us"] * r["calls"])
        bn_ms = sum(r["avg_us"] * r["calls"] for r in passes if r["group"] == "bn_pool") / 1e3
        res = {
            "workload": (f"{'ECGMultimodal (FiLM)' if demo else f'ECGCNN({labels})'} train step fwd+BCE+bwd+AdamW, 12x{T}, "
                         f"{'bf16 conv operands' if bf16 else 'fp32'}, batch {B}/GPU, global batch {B * world}"),
            "value": round(value, 1), "unit": "windows/s", "ms_per_step": round(1e3 * elapsed / spec["steps"], 4),
            "step_ms": step_ms, "value_at_median_step": round(world * B / (step_ms["median"] * 1e-3), 1),
            "dtype": "f32" if not bf16 else ("bf16 conv operands (fwd, input-grad, weight-grad) and inter-kernel activation storage (y, p, dp) / "
                                             "f32 accumulate, BN arithmetic, parameters, tail, optimizer"),
            "dtype_short": "bf16" if bf16 else "f32",
            "workload_short": (f"{'ECGMultimodal' if demo else f'ECGCNN({labels})'} 12x{T} {'bf16' if bf16 else 'f32'} B={B}"
                               f"{' stockAdamW' if stock else ''}{' graph' if graph else ''}"),
            "frac_by_block": frac_by_block(rows), "pipe_frac_by_block": frac_by_block(rows, "pipe_frac"),
            "optimizer": "torch.optim.AdamW (stock)" if stock else "FlatAdamW",
            "loop": ("hipGraph replay (GraphedTrainStep)" if graph else "train_one_epoch" + ("_demo" if demo else "") +
                     (" (hipGraph replay per batch shape)" if loop_replays else "")),
            "final_loss": round(float(last_loss), 6),
            "step_conv_tflops": round(value * step_f / 1e12, 2),
            "step_frac_of_mfma_peak": round(value * step_f / 1e12 / (peak * world), 4),
            "roofline": roofline_of(rows), "layers": rows,
            # everything_else = BatchNorm/pool passes (HBM-bound streams) + the rest (tail, loss, optimizer, weight packs);
            # host_gap = step time the GPU spends on none of them (launch gaps: the eager loop is host-bound when > 0)
            "instrumented_ms_per_step": {"conv_entry_points": round(conv_ms / n_instr, 4),
                                         "everything_else": round(other_ms / n_instr, 4),
                                         "bn_pool_passes": round(bn_ms / n_instr, 4),
                                         "host_gap": round(max(0.0, 1e3 * elapsed / spec["steps"] - (conv_ms + other_ms) / n_instr), 4)},
            "passes": passes,
        }
        if exch:
            torch.cuda.synchronize()
            res["exchange_exposed_ms_per_step"] = percentiles([a.elapsed_time(b) for a, b in exch])
        if exchange_mode:
            res["exchange"] = exchange_mode
        extra = None
        if world > 1 and not stock:
            # RCCL sanity on real gradients: exchange once by hand and compare checksums
            opt.zero_grad()
            with opt.no_sync():
                out = wrapped(*batch[:-1])
                hipF.backward_from_loss(hipF.binary_cross_entropy_with_logits(out, batch[-1]))
            opt._gather(0, len(opt._params))
            local_sum = opt.flat_grad.double().sum()
            total = local_sum.clone()
            dist.all_reduce(total)
            flat, _scale = opt.reduce_gradients()
            after = flat.double().sum().item()
            opt.zero_grad()
            extra = {"grad_checksum_sum_of_ranks": total.item(), "grad_checksum_after_allreduce": after,
                     "rel_diff": abs(after - total.item()) / max(abs(total.item()), 1e-30),
                     "flat_gradient_bytes": int(opt.flat_grad.numel() * 4)}
        del model, opt, wrapped, batch, x, y
        torch.cuda.empty_cache()
        return res, extra

    base = {"batch": args.batch, "length": args.length, "labels": args.labels, "model": args.model,
            "dtype": args.dtype, "optim": args.optim, "graph": args.graph,
            "steps": args.steps, "warmup": args.warmup, "priming": args.priming,
            "priming_seconds": args.priming_seconds}
    primary, rccl_extra = run_leg(base)

    also = []
    if not args.no_also:
        legs = []
        if args.model == "cnn":
            legs.append(dict(base, model="multimodal", graph=False))
        else:
            legs.append(dict(base, model="cnn", graph=False))
        if world == 1 and args.dtype == "f32" and args.length == 1000 and not args.graph:
            if args.optim == "flat":
                legs.append(dict(base, optim="torch"))
            # BASELINE.json configs[4]: ECGCNN(1), 12x5000, batch 256 — fp32 and with bf16 conv operands
            k5 = max(10, args.steps // 2)
            legs.append(dict(base, model="cnn", labels=1, length=5000, optim="flat", steps=k5, priming=10))
            legs.append(dict(base, model="cnn", labels=1, length=5000, optim="flat", dtype="bf16", steps=k5, priming=10))
        for spec in legs:
            res, _ = run_leg(spec)
            also.append(res)

    # The reference's OWN batch sizes through the unchanged loop API (configs/ecg_baseline.yaml:12 batch 64,
    # configs/af_binary.yaml:8 batch 32; 12x5000 is what its 500 Hz records are, 12x1000 BASELINE's window): there the
    # eager step is bound by Python's enqueue time, and train_one_epoch replays a captured hipGraph instead
    # (ecg_hip.graph.LoopStepper).  Both forms are measured; detail file only, one summary row each in the line.
    small = []
    if world == 1 and not args.no_also and not args.graph and args.dtype == "f32" and args.length == 1000 and args.optim == "flat":
        for B_, T_, labels_ in ((64, 5000, 5), (32, 5000, 1), (32, 1000, 5)):
            pair = {}
            for mode in ("1", "0"):
                os.environ["ECG_HIP_LOOP_GRAPH"] = mode
                res, _ = run_leg(dict(base, model="cnn", labels=labels_, length=T_, batch=B_, optim="flat", graph=False,
                                      steps=max(20, args.steps), priming=10, priming_seconds=0.5))
                pair["graph" if mode == "1" else "eager"] = res
            os.environ.pop("ECG_HIP_LOOP_GRAPH", None)
            small.append(pair)

    ranks_seen = None
    if world > 1:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
    if rank == 0:
        rccl = None
        if world > 1:
            rccl = {"backend": backend + (" (one-GPU rehearsal: every rank on device 0)" if rehearsal else ""),
                    "ranks_seen_by_allreduce": ranks_seen, **(rccl_extra or {})}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_seconds)       # LAST: its thread pool must not disturb a GPU leg
        detail_path = args.detail or os.path.join(ROOT, "gpurun_out", f"bench_detail_n{world}.json")
        line, detail = build_line(primary, also, cpu, rccl, n_gpus=world, steps=args.steps, warmup=args.warmup,
                                  batch=args.batch, length=args.length, priming=args.priming,
                                  priming_seconds=args.priming_seconds, n1_value=args.n1_value, small=small)
        try:
            os.makedirs(os.path.dirname(detail_path), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(detail, f, indent=1)
            line["detail"] = os.path.relpath(detail_path, ROOT)
        except OSError as e:
            line["detail"] = f"not written: {e.__class__.__name__}"
        print(json.dumps(line, separators=(",", ":")), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
