"""CPU checks of the input-pipeline row: the oracle against the reference's
committed demo windows (bit for bit), the WFDB format-16 reader/writer, the pack file, the epoch
sampler and the demographic-vector rules.  No GPU, no /root/reference."""
import os

import numpy as np
import pytest

from util import golden

from oracle import input_oracle as io_ref
from ecg_hip import pack, wfdb16


def test_oracle_reproduces_committed_reference_windows_bit_for_bit():
    g = golden("g8_input_pipeline")
    got = io_ref.windows_from_wfdb16(g["d"], g["gain"], g["baseline"])
    assert got.dtype == np.float32 and got.shape == (3, 12, 5000)
    assert np.array_equal(got, g["x"])          # float32, every bit
    # the summation order is part of the answer: a contiguous copy makes numpy sum pairwise and
    # lands on different bits, which is why the kernels walk each lead left to right
    x = np.ascontiguousarray(io_ref.load_ecg(g["d"][0], g["gain"][0], g["baseline"][0]))
    assert not np.array_equal(io_ref.normalize_per_lead(x), g["x"][0])
    np.testing.assert_allclose(io_ref.normalize_per_lead(x), g["x"][0], rtol=1e-5, atol=1e-5)


def test_oracle_invalid_sample_and_calibration():
    d = np.array([[0, 100], [-32768, -100], [5, 7]], np.int16)
    p = io_ref.wfdb16_physical(d, [200.0, 1000.0], [0, -100])
    assert np.isnan(p[1, 0]) and p[1, 1] == 0.0 and p[0, 1] == 0.2 and p[2, 0] == 0.025


def test_demo_vector_rules_and_committed_vectors():
    g = golden("g8_input_pipeline")
    cols = ["age", "sex", "height", "weight", "pacemaker"]
    for r, want in zip(g["demo_rows"], g["demo_vectors"]):
        row = dict(zip(cols, r))
        assert np.array_equal(pack.build_demo_vector(row), want)
        assert np.array_equal(io_ref.build_demo_vector(row), want)
    cases = [
        ({"age": 300.0, "sex": "M", "height": 180, "weight": 80, "pacemaker": 1}, [0.9, 0.0, 0.72, 0.4, 1.0]),
        ({"age": -3, "sex": "F", "height": -1, "weight": 0, "pacemaker": np.nan}, [0.0, 1.0, 0.0, 0.0, 0.0]),
        ({"age": "n/a", "sex": 1, "height": None, "weight": "70", "pacemaker": "x"}, [0.0, 0.5, 0.0, 0.35, 0.0]),
        ({}, [0.0, 0.5, 0.0, 0.0, 0.0]),
        ({"age": np.inf, "sex": "m", "height": np.nan, "weight": np.inf}, [0.0, 0.5, 0.0, 0.0, 0.0]),
        ({"age": 299.9, "sex": "F", "height": 250, "weight": 200, "pacemaker": 2}, [2.999, 1.0, 1.0, 1.0, 2.0]),
    ]
    for row, want in cases:
        v = pack.build_demo_vector(row)
        assert v.dtype == np.float32
        assert np.array_equal(v, np.array(want, np.float32)), row
        assert np.array_equal(v, io_ref.build_demo_vector(row)), row
    assert pack.build_demo_matrix([c[0] for c in cases]).shape == (6, 5)


PTBXL_HEADER = """00001_hr 12 500 5000
00001_hr.dat 16 1000.0(0)/mV 16 0 -115 13047 0 I
00001_hr.dat 16 1000.0(0)/mV 16 0 -50 11561 0 II
00001_hr.dat 16 1000.0(0)/mV 16 0 65 64005 0 III
00001_hr.dat 16 1000.0(0)/mV 16 0 82 50433 0 AVR
00001_hr.dat 16 1000.0(0)/mV 16 0 -90 25400 0 AVL
00001_hr.dat 16 1000.0(0)/mV 16 0 7 5202 0 AVF
00001_hr.dat 16 1000.0(0)/mV 16 0 -65 58191 0 V1
00001_hr.dat 16 1000.0(0)/mV 16 0 -40 64297 0 V2
00001_hr.dat 16 1000.0(0)/mV 16 0 -5 58392 0 V3
00001_hr.dat 16 1000.0(0)/mV 16 0 -35 39859 0 V4
00001_hr.dat 16 1000.0(0)/mV 16 0 -35 30725 0 V5
00001_hr.dat 16 1000.0(0)/mV 16 0 -75 65158 0 V6
# a comment line
"""


def test_header_grammar():
    h = wfdb16.parse_header(PTBXL_HEADER)
    assert (h["name"], h["n_sig"], h["fs"], h["n_samp"]) == ("00001_hr", 12, 500.0, 5000)
    s = h["signals"][3]
    assert (s["file"], s["fmt"], s["gain"], s["baseline"], s["units"]) == ("00001_hr.dat", 16, 1000.0, 0, "mV")
    assert (s["init_value"], s["checksum"], s["description"]) == (82, 50433, "AVR")
    # defaults of the specification: gain 200 when missing/0, baseline = adc_zero, units mV
    h = wfdb16.parse_header("r 2 360 10\nr.dat 16\nr.dat 16 0(12)/uV 11 1024 995 -2 0 lead two\n")
    assert h["signals"][0]["gain"] == 200.0 and h["signals"][0]["baseline"] == 0 and h["signals"][0]["units"] == "mV"
    assert h["signals"][1]["gain"] == 200.0 and h["signals"][1]["baseline"] == 12 and h["signals"][1]["units"] == "uV"
    assert h["signals"][1]["description"] == "lead two"
    h = wfdb16.parse_header("r 1 250/1(0) 5\nr.dat 16 2.5e2 12 -7\n")
    assert h["fs"] == 250.0 and h["signals"][0]["gain"] == 250.0 and h["signals"][0]["baseline"] == -7
    for bad in ["", "r\n", "r/3 2 500 10\n", "r 2 500 10\nr.dat 16\n"]:
        with pytest.raises(wfdb16.WfdbFormatError):
            wfdb16.parse_header(bad)


def test_record_round_trip_checksum_and_refusals(tmp_path):
    rng = np.random.default_rng(1)
    d = rng.integers(-3000, 3000, size=(500, 12)).astype(np.int16)
    gain, base = np.full(12, 1000.0), np.arange(12, dtype=np.int32) - 5
    path = str(tmp_path / "00007_hr")
    wfdb16.write_record(path, d, 500, gain, base, sig_names=[f"L{i}" for i in range(12)])
    r = wfdb16.read_record(path)
    assert np.array_equal(r.d, d) and r.d.dtype == np.int16 and (r.n_samp, r.n_sig) == (500, 12)
    assert np.array_equal(r.gain, gain) and np.array_equal(r.baseline, base) and r.fs == 500.0
    assert r.sig_names[3] == "L3" and r.checksums == wfdb16.checksum16(d)
    # corrupt one sample: the header checksum catches it
    raw = np.fromfile(path + ".dat", dtype="<i2")
    raw[100] += 1
    raw.tofile(path + ".dat")
    with pytest.raises(wfdb16.WfdbFormatError, match="checksum"):
        wfdb16.read_record(path)
    assert wfdb16.read_record(path, verify_checksum=False).d[100 // 12, 100 % 12] == d[100 // 12, 100 % 12] + 1
    # truncated data file
    raw[:100].tofile(path + ".dat")
    with pytest.raises(wfdb16.WfdbFormatError, match="on disk"):
        wfdb16.read_record(path)
    # formats this path does not stream
    hea = open(path + ".hea").read()
    open(path + ".hea", "w").write(hea.replace(".dat 16 ", ".dat 212 "))
    with pytest.raises(wfdb16.WfdbFormatError, match="format 16"):
        wfdb16.read_record(path)
    assert wfdb16.checksum16(np.array([[32767], [1]], np.int16)) == [-32768]


def _toy_pack(tmp_path, n=11, T=64, demo=True):
    rng = np.random.default_rng(5)
    d = rng.integers(-2000, 2000, size=(n, T, 12)).astype(np.int16)
    gain = np.full((n, 12), 1000.0)
    base = rng.integers(-3, 3, size=(n, 12)).astype(np.int32)
    y = (rng.random((n, 5)) < 0.3).astype(np.float32)
    dm = rng.random((n, 5)).astype(np.float32) if demo else None
    path = str(tmp_path / "toy.ecgpack")
    pack.write_pack(path, d, gain, base, y, dm, ids=np.arange(100, 100 + n))
    return path, d, gain, base, y, dm


def test_pack_round_trip_and_validation(tmp_path):
    path, d, gain, base, y, dm = _toy_pack(tmp_path)
    p = pack.EcgPack(path)
    assert (len(p), p.T, p.leads, p.n_labels, p.demo_dim) == (11, 64, 12, 5, 5)
    assert np.array_equal(p.samples, d) and np.array_equal(p.gain, gain) and np.array_equal(p.baseline, base)
    assert np.array_equal(p.labels, y) and np.array_equal(p.demo, dm) and p.ids[3] == 103
    for name, (off, _) in p.header["sections"].items():
        assert off % 4096 == 0, name
    p2 = pack.EcgPack(_toy_pack(tmp_path, demo=False)[0])
    assert p2.demo is None and p2.demo_dim == 0
    with open(path, "r+b") as f:
        f.truncate(os.path.getsize(path) // 2)
    with pytest.raises(ValueError, match="truncated"):
        pack.EcgPack(path)
    open(path, "wb").write(b"not a pack")
    with pytest.raises(ValueError, match="ECGPACK1"):
        pack.EcgPack(path)


def test_pack_from_wfdb_directory(tmp_path):
    rng = np.random.default_rng(9)
    rels, ds = [], []
    for i in range(4):
        rel = f"records500/00000/{i:05d}_hr"
        os.makedirs(os.path.dirname(tmp_path / rel), exist_ok=True)
        d = rng.integers(-500, 500, size=(80, 12)).astype(np.int16)
        wfdb16.write_record(str(tmp_path / rel), d, 500, np.full(12, 1000.0), np.zeros(12, np.int32))
        rels.append(rel)
        ds.append(d)
    y = np.eye(4, 5, dtype=np.float32)
    out = str(tmp_path / "split.ecgpack")
    pack.build_pack_from_wfdb(out, str(tmp_path), rels, y, ids=[7, 8, 9, 10])
    p = pack.EcgPack(out)
    assert np.array_equal(p.samples, np.stack(ds)) and np.array_equal(p.labels, y) and list(p.ids) == [7, 8, 9, 10]
    wfdb16.write_record(str(tmp_path / "odd"), ds[0][:40], 500, np.full(12, 1000.0), np.zeros(12, np.int32))
    with pytest.raises(wfdb16.WfdbFormatError, match="differs"):
        pack.build_pack_from_wfdb(out, str(tmp_path), rels + ["odd"], np.zeros((5, 5), np.float32))


def test_epoch_sampler_shards_without_overlap():
    n, B = 103, 8
    full = pack.epoch_indices(n, True, 42, 3)
    assert sorted(full) == list(range(n))
    assert not np.array_equal(full, pack.epoch_indices(n, True, 42, 4))        # reshuffled per epoch
    assert np.array_equal(full, pack.epoch_indices(n, True, 42, 3))            # deterministic
    for world in (2, 8):
        shards = [pack.epoch_indices(n, True, 42, 3, rank=r, world_size=world) for r in range(world)]
        assert len({len(s) for s in shards}) == 1                              # same step count on every rank
        seen = np.concatenate(shards)
        assert set(seen) == set(range(n)) and len(seen) - n < world            # only wrap-around padding repeats
        assert np.array_equal(np.stack(shards, 1).reshape(-1)[:n], full)       # rank-strided slices of ONE permutation
    dl = pack.epoch_indices(n, False, 0, 0, rank=1, world_size=2, drop_last=True, batch_size=B)
    assert len(dl) % B == 0 and len(dl) == (52 // B) * B and dl[0] == 1 and dl[1] == 3


def test_loader_refuses_cpu(tmp_path):
    import torch
    from ecg_hip import EcgHipError
    path = _toy_pack(tmp_path)[0]
    if not torch.cuda.is_available():
        with pytest.raises(EcgHipError, match="prepares batches on the GPU"):
            pack.PackedBatchLoader(path, 4, device="cuda")
    with pytest.raises(EcgHipError, match="prepares batches on the GPU"):
        pack.PackedBatchLoader(path, 4, device="cpu")


def test_loader_gather_through_the_native_row_copy(tmp_path):
    """The host side of PackedBatchLoader._stage without a GPU: records -> pinned slot through ecg_host_gather_rows (a
    stateless memcpy loop of libecg_hip.so, one call per gather thread); any thread count gives the same bytes, runs of
    consecutive records included, and a bad index is refused before anything is copied."""
    L = pack.PackedBatchLoader
    rng = np.random.default_rng(0)
    samples = rng.integers(-3000, 3000, size=(64, 20000, 12)).astype(np.int16)     # 480 KB per record: the pool is used

    class FakePack:
        pass
    for idx in (np.arange(8, 40), rng.permutation(64)[:37], np.array([3]), np.array([7, 8, 9, 1, 2, 60, 61, 62, 63, 0]),
                np.array([], dtype=np.int64)):
        want = samples[idx]
        for threads in (1, 2, 5, 8):
            ld = object.__new__(L)
            ld.pack, ld.gather_threads, ld._pool = FakePack(), threads, None
            ld.pack.samples = samples
            dst = np.zeros((40, 20000, 12), np.int16)
            ld._gather_samples(dst, idx)
            assert np.array_equal(dst[:len(idx)], want) and not dst[len(idx):].any(), (len(idx), threads)
            if ld._pool is not None:
                ld._pool.shutdown()
    ld = object.__new__(L)
    ld.pack, ld.gather_threads, ld._pool = FakePack(), 2, None
    ld.pack.samples = samples
    for bad in ([0, 64], [-1, 3]):
        with pytest.raises(IndexError):
            ld._gather_samples(np.zeros((4, 20000, 12), np.int16), np.array(bad))
