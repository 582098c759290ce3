"""Packed on-disk window cache + GPU batch loader.

The reference reads every item with `wfdb.rdsamp` inside `Dataset.__getitem__`
(src/datasets/ptbxl.py:25,140-144), converts to float64 physical units, casts, transposes and
z-scores on the host, then the DataLoader collates and the loop copies fp32 to the device
(src/training/loop.py:27-28).  Here the records are packed ONCE into a single memory-mappable file
of raw int16 samples; a batch is a gather of B records into a pinned buffer, one H2D copy of
2 bytes/sample on a copy stream, and `wfdb16_to_windows` on the GPU (gain/baseline, float32 cast,
transpose, per-lead z-score — bit-identical to the reference's arithmetic).

File layout (little-endian):
    bytes 0..7     magic  b"ECGPACK1"
    bytes 8..15    uint64 length of the JSON header
    JSON header    {"n", "T", "leads", "n_labels", "demo_dim", "sections": {name: [offset, nbytes]}}
    sections, each 4096-byte aligned:
        samples  int16   [n][T][leads]     (time-major, as in a format-16 .dat)
        gain     float64 [n][leads]
        baseline int32   [n][leads]
        labels   float32 [n][n_labels]
        demo     float32 [n][demo_dim]     (absent when demo_dim == 0)
        ids      int64   [n]
"""
import json
import os
import time

import numpy as np
import torch

from . import _lib as L
from . import wfdb16

MAGIC = b"ECGPACK1"
_ALIGN = 4096
_DTYPES = {"samples": "<i2", "gain": "<f8", "baseline": "<i4", "labels": "<f4", "demo": "<f4", "ids": "<i8"}


# ---------------------------------------------------------------------------------------------
# demographic features (host logic; reference src/datasets/ptbxl_ecg_multimodal.py:106-164)
# ---------------------------------------------------------------------------------------------
def _as_float(v, default=0.0):
    try:
        return float(v)
    except Exception:
        return default


def build_demo_vector(row):
    """[age/100, sex_id, height/250, weight/200, pacemaker] as float32; `row` is any mapping with
    .get (a pandas row, a dict).  Same rules as the reference, including its quirks: age >= 300
    (PTB-XL's anonymised 'older than 89') becomes 90, sex is matched against the STRINGS "M"/"F" and is
    0.5 for anything else (so numeric CSV sex columns always give 0.5), non-positive or missing
    height/weight give 0."""
    age = _as_float(row.get("age", np.nan))
    if (not np.isfinite(age)) or age < 0:
        age = 0.0
    if age >= 300:
        age = 90.0
    sex = row.get("sex", "UNKNOWN")
    if isinstance(sex, str) and sex == "M":
        sex_id = 0.0
    elif isinstance(sex, str) and sex == "F":
        sex_id = 1.0
    else:
        sex_id = 0.5
    height = _as_float(row.get("height", np.nan))
    if (not np.isfinite(height)) or height <= 0:
        height = 0.0
    weight = _as_float(row.get("weight", np.nan))
    if (not np.isfinite(weight)) or weight <= 0:
        weight = 0.0
    pace = _as_float(row.get("pacemaker", 0))
    if not np.isfinite(pace):
        pace = 0.0
    return np.array([age / 100.0, sex_id, height / 250.0, weight / 200.0, pace], dtype=np.float32)


def build_demo_matrix(rows):
    """rows: iterable of mappings -> float32 [n, 5]."""
    out = [build_demo_vector(r) for r in rows]
    return np.stack(out) if out else np.zeros((0, 5), np.float32)


# ---------------------------------------------------------------------------------------------
# pack file
# ---------------------------------------------------------------------------------------------
def _round_up(x, a=_ALIGN):
    return (x + a - 1) // a * a


def write_pack(path, samples, gain, baseline, labels, demo=None, ids=None):
    """samples int16 [n, T, leads]; gain float64 / baseline int32 [n, leads]; labels float32 [n, C];
    demo float32 [n, D] or None; ids int64 [n] or None."""
    samples = np.ascontiguousarray(samples, dtype="<i2")
    if samples.ndim != 3:
        raise ValueError("samples must be [n, T, leads]")
    n, T, leads = samples.shape
    arrays = {
        "samples": samples,
        "gain": np.ascontiguousarray(gain, dtype="<f8").reshape(n, leads),
        "baseline": np.ascontiguousarray(baseline, dtype="<i4").reshape(n, leads),
        "labels": np.ascontiguousarray(labels, dtype="<f4").reshape(n, -1),
        "ids": np.ascontiguousarray(ids if ids is not None else np.arange(n), dtype="<i8").reshape(n),
    }
    if demo is not None:
        arrays["demo"] = np.ascontiguousarray(demo, dtype="<f4").reshape(n, -1)
    header = {"n": n, "T": T, "leads": leads, "n_labels": int(arrays["labels"].shape[1]),
              "demo_dim": int(arrays["demo"].shape[1]) if demo is not None else 0, "sections": {}}
    # two passes: section offsets depend on the header length
    for _ in range(2):
        hdr = json.dumps(header).encode()
        off = _round_up(16 + len(hdr) + 256)          # slack so the second pass cannot move the data
        for name, a in arrays.items():
            header["sections"][name] = [off, int(a.nbytes)]
            off = _round_up(off + a.nbytes)
    hdr = json.dumps(header).encode()
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(np.uint64(len(hdr)).tobytes())
        f.write(hdr)
        for name, a in arrays.items():
            f.seek(header["sections"][name][0])
            a.tofile(f)
        f.truncate(_round_up(f.tell()))
    os.replace(tmp, path)
    return header


def build_pack_from_wfdb(path, base_dir, rel_paths, labels, demo=None, ids=None, verify_checksum=True):
    """Pack PTB-XL style records (`os.path.join(base_dir, rel)` + .hea/.dat, the `filename_hr` column of
    the reference's dataframe) in the given order.  All records must share T and the lead count."""
    recs = [wfdb16.read_record(os.path.join(base_dir, rel), verify_checksum) for rel in rel_paths]
    if not recs:
        raise ValueError("no records")
    T, leads = recs[0].d.shape
    for rel, r in zip(rel_paths, recs):
        if r.d.shape != (T, leads):
            raise wfdb16.WfdbFormatError(f"{rel}: shape {r.d.shape} differs from the first record's {(T, leads)}")
    return write_pack(path, np.stack([r.d for r in recs]), np.stack([r.gain for r in recs]),
                      np.stack([r.baseline for r in recs]), labels, demo, ids)


class EcgPack:
    """Read-only memory-mapped view of a pack file."""

    def __init__(self, path):
        self.path = path
        with open(path, "rb") as f:
            if f.read(8) != MAGIC:
                raise ValueError(f"{path}: not an ECGPACK1 file")
            hlen = int(np.frombuffer(f.read(8), dtype="<u8")[0])
            self.header = json.loads(f.read(hlen).decode())
        h = self.header
        self.n, self.T, self.leads = h["n"], h["T"], h["leads"]
        self.n_labels, self.demo_dim = h["n_labels"], h["demo_dim"]
        shapes = {"samples": (self.n, self.T, self.leads), "gain": (self.n, self.leads),
                  "baseline": (self.n, self.leads), "labels": (self.n, self.n_labels),
                  "demo": (self.n, self.demo_dim), "ids": (self.n,)}
        size = os.path.getsize(path)
        for name, (off, nbytes) in h["sections"].items():
            want = int(np.prod(shapes[name])) * np.dtype(_DTYPES[name]).itemsize
            if nbytes != want or off + nbytes > size:
                raise ValueError(f"{path}: section {name} is truncated or mis-sized")
            setattr(self, name, np.memmap(path, dtype=_DTYPES[name], mode="r", offset=off, shape=shapes[name]))
        if "demo" not in h["sections"]:
            self.demo = None

    def __len__(self):
        return self.n


# ---------------------------------------------------------------------------------------------
# batch loader
# ---------------------------------------------------------------------------------------------
def epoch_indices(n, shuffle, seed, epoch, rank=0, world_size=1, drop_last=False, batch_size=1):
    """Record order of one epoch for one rank.  Every rank draws the SAME permutation (seed + epoch)
    and takes the rank-strided slice — windows are independent, no data-path collective.  The order is
    padded by wrapping (like torch's DistributedSampler) so that all ranks run the same number of
    steps; with drop_last the per-rank tail that does not fill a batch is dropped instead."""
    if shuffle:
        order = np.random.default_rng(np.random.SeedSequence([int(seed), int(epoch)])).permutation(n)
    else:
        order = np.arange(n)
    if world_size > 1:
        per = -(-n // world_size)
        if per * world_size > n:
            order = np.concatenate([order, order[:per * world_size - n]])
        order = order[rank::world_size]
    if drop_last:
        order = order[:len(order) // batch_size * batch_size]
    return order


class _IteratedWindows:
    """`loader.dataset` of a PackedBatchLoader: what `len(loader.dataset)` must be for the loops' epoch-loss
    normalisation (src/training/loop.py:38,73 divide by it) — the number of windows THIS rank iterates in the
    current epoch (after sharding, wrap-padding and drop_last), re-evaluated on every len()."""

    def __init__(self, loader):
        self._loader = loader

    def __len__(self):
        return len(self._loader.indices())

    def __getitem__(self, i):
        ld = self._loader
        rec = int(ld.indices()[i])
        p = ld.pack
        out = (p.samples[rec], p.labels[rec])
        return out if not ld.with_demo else (p.samples[rec], p.demo[rec], p.labels[rec])


class PackedBatchLoader:
    """Iterable of device batches, shaped like the reference's DataLoader output so that
    `train_one_epoch(model, loader, optimizer, device)` (src/training/loop.py:14) and
    `train_one_epoch_demo` (loop_demo.py:13) consume it unchanged:
        (x [B,12,T] fp32, y [B,C] fp32)                 when the pack has no demographics / with_demo=False
        (x [B,12,T] fp32, x_demo [B,5] fp32, y [B,C])   otherwise
    Batches are produced on a side stream one step ahead (pinned double buffer -> async H2D ->
    wfdb16_to_windows); the consumer's stream waits on the batch's event."""

    def __init__(self, pack, batch_size, shuffle=False, seed=0, drop_last=False, device="cuda",
                 rank=0, world_size=1, with_demo=None, normalize=True, gather_threads=None, slots=3):
        self.pack = pack if isinstance(pack, EcgPack) else EcgPack(pack)
        self.batch_size, self.shuffle, self.seed, self.drop_last = int(batch_size), shuffle, seed, drop_last
        self.rank, self.world_size = rank, world_size
        self.with_demo = (self.pack.demo is not None) if with_demo is None else with_demo
        if self.with_demo and self.pack.demo is None:
            raise ValueError("with_demo=True but the pack holds no demographics")
        self.normalize = normalize
        self.device = torch.device(device)
        self.epoch = 0
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise L.EcgHipError("PackedBatchLoader prepares batches on the GPU; "
                                "pass device='cuda' on a machine with an MI355X")
        L.load()
        self._stream = torch.cuda.Stream(device=self.device)
        self._slots = None
        # host gather: the records of a batch are copied out of the mapped file into the pinned slot by a small thread
        # pool (numpy releases the GIL for these copies); round 4's single-threaded gather ran at 18.7 GB/s = 156 k
        # windows/s of 12x5000, below the bf16 train step it feeds.  gather_threads=1 keeps everything on the caller.
        self.gather_threads = max(1, int(gather_threads if gather_threads is not None
                                         else min(8, (os.cpu_count() or 2) // 2)))
        self.n_slots = max(2, int(slots))
        self._pool = None
        self.stage_seconds = {"gather": 0.0, "enqueue": 0.0, "slot_wait": 0.0, "batches": 0}
        self.dataset = _IteratedWindows(self)      # train_one_epoch / eval_one_epoch end with len(loader.dataset)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def indices(self):
        return epoch_indices(len(self.pack), self.shuffle, self.seed, self.epoch, self.rank, self.world_size,
                             self.drop_last, self.batch_size)

    def __len__(self):
        n = len(self.indices())
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _make_slots(self):
        p, B = self.pack, self.batch_size
        slots = []
        for _ in range(self.n_slots):
            s = {"d": torch.empty((B, p.T, p.leads), dtype=torch.int16).pin_memory(),
                 "gain": torch.empty((B, p.leads), dtype=torch.float64).pin_memory(),
                 "base": torch.empty((B, p.leads), dtype=torch.int32).pin_memory(),
                 "y": torch.empty((B, p.n_labels), dtype=torch.float32).pin_memory(),
                 "free": None}
            if self.with_demo:
                s["demo"] = torch.empty((B, p.demo_dim), dtype=torch.float32).pin_memory()
            slots.append(s)
        return slots

    def _stage(self, slot, idx):
        """Gather records `idx` into the slot's pinned buffers and run the GPU preparation on the side
        stream.  Returns (tensors..., ready_event)."""
        from . import functional as F
        p, b = self.pack, len(idx)
        t0 = time.perf_counter()
        if slot["free"] is not None:
            slot["free"].synchronize()           # the H2D copies that last used this slot are done
        t1 = time.perf_counter()
        self._gather_samples(slot["d"].numpy(), idx)
        for key, src in (("gain", p.gain), ("base", p.baseline), ("y", p.labels)) + \
                ((("demo", p.demo),) if self.with_demo else ()):
            np.take(src, idx, axis=0, out=slot[key].numpy()[:b])
        t2 = time.perf_counter()
        st = self.stage_seconds
        st["slot_wait"] += t1 - t0
        st["gather"] += t2 - t1
        st["batches"] += 1
        with torch.cuda.stream(self._stream):
            dev = {k: slot[k][:b].to(self.device, non_blocking=True)
                   for k in ("d", "gain", "base", "y") + (("demo",) if self.with_demo else ())}
            copied = torch.cuda.Event()
            copied.record(self._stream)
            x = F.wfdb16_to_windows(dev["d"], dev["gain"], dev["base"], normalize=self.normalize)
            ready = torch.cuda.Event()
            ready.record(self._stream)
        slot["free"] = copied
        st["enqueue"] += time.perf_counter() - t2
        out = (x, dev["demo"], dev["y"]) if self.with_demo else (x, dev["y"])
        return out, ready

    def _gather_samples(self, dst, idx):
        """int16 records `idx` of the mapped file -> rows 0.. of the pinned slot, by `ecg_host_gather_rows` (a memcpy per
        run of consecutive records, GIL released for the whole call), the batch cut into contiguous shares for the gather
        threads.  (Measured on the way: per-record numpy copies from a Python thread pool were SLOWER than one thread — 3.9 ms
        against 0.6 ms per batch of 256 x 12x1000 — every copy hands the GIL over; numpy's fancy indexing reaches a third
        of the memcpy rate on rows this long.)"""
        src = self.pack.samples
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        n, row = len(idx), src[0].nbytes
        if n == 0:
            return
        if idx.min() < 0 or idx.max() >= len(src):
            raise IndexError("PackedBatchLoader: record index out of range")
        lib = L.load()
        s_ptr, d_ptr, i_ptr = src.ctypes.data, dst.ctypes.data, idx.ctypes.data

        def work(a, b):
            rc = lib.ecg_host_gather_rows(s_ptr, row, i_ptr + 8 * a, b - a, d_ptr + a * row)
            if rc != 0:
                raise L.EcgHipError(f"ecg_host_gather_rows failed: {L.last_error()}")
        W = min(self.gather_threads, n, max(1, (n * row) >> 20))         # at least ~1 MB per share
        if W <= 1:
            return work(0, n)
        cuts = [n * k // W for k in range(W + 1)]
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self.gather_threads, thread_name_prefix="ecgpack-gather")
        futs = [self._pool.submit(work, cuts[k], cuts[k + 1]) for k in range(1, W)]
        work(cuts[0], cuts[1])
        for f in futs:
            f.result()

    def __iter__(self):
        if self._slots is None:
            self._slots = self._make_slots()
        order = self.indices()
        B = self.batch_size
        nb = len(self)
        batches = [order[i * B:(i + 1) * B] for i in range(nb)]
        pending = None
        for i, idx in enumerate(batches):
            staged = self._stage(self._slots[i % self.n_slots], idx)
            if pending is not None:
                yield self._hand_over(pending)
            pending = staged
        if pending is not None:
            yield self._hand_over(pending)

    def _hand_over(self, staged):
        out, ready = staged
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        for t in out:
            t.record_stream(cur)                 # allocated on the side stream, consumed on this one
        return out
