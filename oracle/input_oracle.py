"""CPU restatement of the reference's INPUT path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (--workload input) may import this
module; the product path (ecg_hip.functional.wfdb16_to_windows, ecg_hip.pack) never does.

What it restates
  * `wfdb.rdsamp` for a format-16 record — third-party dependency absent from /root/reference and
    from this image: wfdb==4.3.0 (reference requirements.txt:63).  Published algorithm
    (wfdb/io/_signal.py, SignalMixin.dac, float64 result): p = (d - baseline) / adc_gain evaluated in
    float64, samples equal to the format's invalid value (-32768 for format 16) become NaN.
  * `_load_ecg`   reference src/datasets/ptbxl.py:14-41  (float32 cast of rdsamp's [T, 12], then .T)
  * `_normalize`  reference src/datasets/ptbxl.py:122-127 (same code: ptbxl_ecg_multimodal.py:98-103,
    ptbxl_af.py) — kept as the SAME numpy calls on the SAME transposed view, because the summation
    order numpy picks for a strided reduction (left to right per lead) is part of the result.
  * `_build_demo_vector`  reference src/datasets/ptbxl_ecg_multimodal.py:106-164

Pinned by tests/golden/g8_input_pipeline.npz: three of the reference's committed demo windows
(data/demo/*.npy — outputs of exactly this path on real PTB-XL records) together with int16 records
recovered from them; this module reproduces those windows bit for bit (tests/test_input_oracle.py).
The demo-vector rules are pinned only by the seven committed multimodal demo vectors (the raw
PTB-XL rows are not in the tree).
"""
import numpy as np

FMT16_INVALID = -32768


def wfdb16_physical(d, gain, baseline):
    """d int16 [T, n_sig], gain float64 [n_sig], baseline int [n_sig] -> float64 [T, n_sig] (rdsamp's `sig`)."""
    d = np.asarray(d)
    p = d.astype(np.float64)
    p = (p - np.asarray(baseline, dtype=np.float64)) / np.asarray(gain, dtype=np.float64)
    p[d == FMT16_INVALID] = np.nan
    return p


def load_ecg(d, gain, baseline):
    """reference _load_ecg: float32 [n_sig, T] VIEW of a [T, n_sig] buffer (ptbxl.py:29,41)."""
    sig = np.asarray(wfdb16_physical(d, gain, baseline), dtype=np.float32)
    return sig.T


def normalize_per_lead(x):
    """reference _normalize (ptbxl.py:122-127), verbatim semantics on whatever view it is handed."""
    mean = x.mean(axis=1, keepdims=True)
    std = x.std(axis=1, keepdims=True) + 1e-6
    return (x - mean) / std


def windows_from_wfdb16(d, gain, baseline):
    """Batch form: d [B, T, n_sig], gain/baseline [B, n_sig] -> float32 [B, n_sig, T]."""
    return np.stack([np.ascontiguousarray(normalize_per_lead(load_ecg(d[i], gain[i], baseline[i])))
                     for i in range(len(d))])


def _to_float(v, default):
    try:
        return float(v)
    except Exception:
        return default


def build_demo_vector(row):
    """reference _build_demo_vector (ptbxl_ecg_multimodal.py:106-164); row: mapping with .get."""
    age = _to_float(row.get("age", np.nan), 0.0)
    if (not np.isfinite(age)) or (age < 0):
        age = 0.0
    if age >= 300:
        age = 90.0
    sex = row.get("sex", "UNKNOWN")
    sex_id = 0.0 if sex == "M" else (1.0 if sex == "F" else 0.5)
    height = _to_float(row.get("height", np.nan), 0.0)
    if (not np.isfinite(height)) or (height <= 0):
        height = 0.0
    weight = _to_float(row.get("weight", np.nan), 0.0)
    if (not np.isfinite(weight)) or (weight <= 0):
        weight = 0.0
    pace = _to_float(row.get("pacemaker", 0), 0.0)
    if not np.isfinite(pace):
        pace = 0.0
    return np.array([age / 100.0, sex_id, height / 250.0, weight / 200.0, pace], dtype=np.float32)
