"""Minimal WFDB reader/writer for the records PTB-XL ships: one segment, every signal in ONE
format-16 .dat file (little-endian int16, samples interleaved by time).

It replaces the `wfdb.rdsamp(record_path)` call of the reference's `_load_ecg`
(src/datasets/ptbxl.py:25; wfdb==4.3.0 in requirements.txt:63) WITHOUT converting to physical
units: the digital samples stay int16 and travel to the GPU as they are (2 bytes/sample);
`ecg_hip.functional.wfdb16_to_windows` applies gain/baseline, the float32 cast, the transpose and
the per-lead z-score there.  Header grammar follows the WFDB header specification (header(5)):

    record line : name[/segments] n_sig [fs[/counter_freq][(base_counter)] [n_samp [time [date]]]]
    signal line : file fmt[xSPF][:skew][+offset] [gain[(baseline)][/units] [adc_res [adc_zero
                  [init_value [checksum [block_size [description]]]]]]]

Defaults as in the specification: gain 200 adu/unit when absent or 0, baseline = adc_zero when not
given, units mV.  Anything this path cannot stream verbatim (other formats, several samples per
frame, skew, byte offsets, multi-segment or multi-file records) is refused loudly.
"""
import os
import re
from dataclasses import dataclass, field
from typing import List

import numpy as np

FMT16_INVALID = -32768
_SIG_RE = re.compile(
    r"^(?P<file>\S+)\s+(?P<fmt>\d+)(?:x(?P<spf>\d+))?(?::(?P<skew>\d+))?(?:\+(?P<off>\d+))?"
    r"(?:\s+(?P<gain>-?[\d.eE+-]+?)?(?:\((?P<base>-?\d+)\))?(?:/(?P<units>\S+))?"
    r"(?:\s+(?P<res>-?\d+)(?:\s+(?P<zero>-?\d+)(?:\s+(?P<init>-?\d+)(?:\s+(?P<csum>-?\d+)"
    r"(?:\s+(?P<blk>-?\d+)(?:\s+(?P<desc>.*))?)?)?)?)?)?)?\s*$")


class WfdbFormatError(ValueError):
    pass


@dataclass
class WfdbRecord:
    name: str
    fs: float
    d: np.ndarray                     # int16 [n_samp, n_sig], time-major (the .dat layout)
    gain: np.ndarray                  # float64 [n_sig]   adu per physical unit
    baseline: np.ndarray              # int32 [n_sig]     adu value of physical zero
    units: List[str] = field(default_factory=list)
    sig_names: List[str] = field(default_factory=list)
    checksums: List[int] = field(default_factory=list)

    @property
    def n_sig(self):
        return self.d.shape[1]

    @property
    def n_samp(self):
        return self.d.shape[0]


def checksum16(d):
    """Per-signal WFDB checksum: sum of the samples as a signed 16-bit integer."""
    s = np.asarray(d, dtype=np.int64).sum(axis=0)
    return [int(((int(v) + 32768) % 65536) - 32768) for v in s]


def parse_header(text):
    """-> dict(name, n_sig, fs, n_samp, signals=[dict(file, fmt, gain, baseline, units, adc_zero,
    init_value, checksum, description)])"""
    lines = [ln.strip() for ln in text.splitlines()]
    lines = [ln for ln in lines if ln and not ln.startswith("#")]
    if not lines:
        raise WfdbFormatError("empty header")
    rec = lines[0].split()
    if len(rec) < 2:
        raise WfdbFormatError(f"bad record line: {lines[0]!r}")
    if "/" in rec[0]:
        raise WfdbFormatError("multi-segment records are not supported")
    n_sig = int(rec[1])
    fs = 250.0
    if len(rec) > 2:
        fs = float(re.match(r"[\d.eE+-]+", rec[2]).group(0))
    n_samp = int(rec[3]) if len(rec) > 3 else None
    if len(lines) - 1 < n_sig:
        raise WfdbFormatError(f"header announces {n_sig} signals but has {len(lines) - 1} signal lines")
    signals = []
    for ln in lines[1:1 + n_sig]:
        m = _SIG_RE.match(ln)
        if not m:
            raise WfdbFormatError(f"bad signal line: {ln!r}")
        g = m.groupdict()
        gain = float(g["gain"]) if g["gain"] else 0.0
        if gain == 0.0:
            gain = 200.0
        zero = int(g["zero"]) if g["zero"] is not None else 0
        signals.append(dict(
            file=g["file"], fmt=int(g["fmt"]), spf=int(g["spf"] or 1), skew=int(g["skew"] or 0),
            offset=int(g["off"] or 0), gain=gain,
            baseline=int(g["base"]) if g["base"] is not None else zero,
            units=g["units"] or "mV", adc_zero=zero,
            init_value=int(g["init"]) if g["init"] is not None else None,
            checksum=int(g["csum"]) if g["csum"] is not None else None,
            description=(g["desc"] or "").strip()))
    return dict(name=rec[0], n_sig=n_sig, fs=fs, n_samp=n_samp, signals=signals)


def read_record(record_path, verify_checksum=True):
    """record_path without extension (as the reference passes to wfdb.rdsamp) -> WfdbRecord."""
    with open(record_path + ".hea", "r") as f:
        hdr = parse_header(f.read())
    sigs = hdr["signals"]
    if hdr["n_sig"] < 1:
        raise WfdbFormatError(f"{record_path}: no signals")
    files = {s["file"] for s in sigs}
    if len(files) != 1:
        raise WfdbFormatError(f"{record_path}: signals spread over {len(files)} files")
    for s in sigs:
        if s["fmt"] != 16 or s["spf"] != 1 or s["skew"] != 0 or s["offset"] != 0:
            raise WfdbFormatError(f"{record_path}: only plain format 16 is supported "
                                  f"(got format {s['fmt']}, x{s['spf']}, skew {s['skew']}, offset {s['offset']})")
    dat = os.path.join(os.path.dirname(record_path), sigs[0]["file"])
    raw = np.fromfile(dat, dtype="<i2")
    n_sig = hdr["n_sig"]
    n_samp = hdr["n_samp"] if hdr["n_samp"] is not None else raw.size // n_sig
    if raw.size < n_samp * n_sig:
        raise WfdbFormatError(f"{dat}: {raw.size} samples on disk, header promises {n_samp}x{n_sig}")
    d = raw[:n_samp * n_sig].reshape(n_samp, n_sig)
    sums = checksum16(d)
    if verify_checksum:
        for i, s in enumerate(sigs):
            if s["checksum"] is not None and s["checksum"] != sums[i]:
                raise WfdbFormatError(f"{record_path}: checksum mismatch on signal {i} "
                                      f"(header {s['checksum']}, data {sums[i]})")
    return WfdbRecord(name=hdr["name"], fs=hdr["fs"], d=d,
                      gain=np.array([s["gain"] for s in sigs], np.float64),
                      baseline=np.array([s["baseline"] for s in sigs], np.int32),
                      units=[s["units"] for s in sigs], sig_names=[s["description"] for s in sigs],
                      checksums=sums)


def write_record(record_path, d, fs, gain, baseline, units=None, sig_names=None):
    """Write d int16 [n_samp, n_sig] as <record_path>.hea/.dat (format 16)."""
    d = np.ascontiguousarray(d, dtype="<i2")
    n_samp, n_sig = d.shape
    name = os.path.basename(record_path)
    units = units or ["mV"] * n_sig
    sig_names = sig_names or [f"sig{i}" for i in range(n_sig)]
    sums = checksum16(d)
    d.tofile(record_path + ".dat")
    with open(record_path + ".hea", "w") as f:
        f.write(f"{name} {n_sig} {fs:g} {n_samp}\n")
        for i in range(n_sig):
            f.write(f"{name}.dat 16 {float(gain[i])!r}({int(baseline[i])})/{units[i]} 16 0 "
                    f"{int(d[0, i])} {sums[i]} 0 {sig_names[i]}\n")
