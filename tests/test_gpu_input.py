"""GPU parity of the input-pipeline row through the C ABI: WFDB format-16
samples -> z-scored fp32 windows, BIT-EXACT against the oracle and against the reference's committed
demo windows; the packed loader end to end; and size-independent properties at the full batch."""
import numpy as np
import pytest
import torch

from util import golden

from oracle import input_oracle as io_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available()
    import ecg_hip
    from ecg_hip import _lib, functional
    ecg_hip.load()
    _lib.call("ecg_check_device")
    return functional


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def test_committed_reference_windows_bit_for_bit(hip):
    g = golden("g8_input_pipeline")
    x = hip.wfdb16_to_windows(dev(g["d"]), dev(g["gain"]), dev(g["baseline"].astype(np.int32)))
    assert x.dtype == torch.float32 and tuple(x.shape) == (3, 12, 5000)
    assert np.array_equal(host(x), g["x"])


# (B, T, leads): ragged transpose tiles, T not a multiple of 4 (scalar walk), one lead, 16 leads, T=1
# T <= 1344 (12 leads) runs the fused LDS-resident kernel, longer windows the three streaming launches;
# 1344/1345 straddle the switch
@pytest.mark.parametrize("shape", [(5, 1000, 12), (3, 257, 12), (2, 5000, 12), (4, 63, 1), (2, 300, 16),
                                   (7, 1, 12), (1, 1022, 3), (33, 128, 12), (2, 16000, 12), (2, 20000, 3),
                                   (1, 40001, 2), (1, 40000, 2), (3, 5001, 12), (2, 1344, 12), (2, 1345, 12), (2, 1300, 12)])
def test_wfdb16_to_windows_vs_oracle_exact(hip, shape):
    B, T, leads = shape
    rng = np.random.default_rng(B * 7 + T + leads)
    d = rng.integers(-4000, 4000, size=shape).astype(np.int16)
    d[0, 0, 0], d[-1, -1, -1] = 32767, -32767
    gain = rng.choice([200.0, 1000.0, 1000.5, 3.3333e3], size=(B, leads))
    base = rng.integers(-50, 50, size=(B, leads)).astype(np.int32)
    phys = host(hip.wfdb16_to_windows(dev(d), dev(gain), dev(base), normalize=False))
    want_p = np.stack([np.ascontiguousarray(io_ref.load_ecg(d[i], gain[i], base[i])) for i in range(B)])
    assert np.array_equal(phys, want_p)
    xt, stats = hip.wfdb16_to_windows(dev(d), dev(gain), dev(base), return_stats=True)
    x = host(xt)
    if leads > 1:
        want_mean = np.concatenate([io_ref.load_ecg(d[i], gain[i], base[i]).mean(axis=1) for i in range(B)])
        assert np.array_equal(host(stats)[:, 0], want_mean)
    if leads == 1:
        # a one-lead [T,1] buffer is contiguous either way, numpy then sums pairwise: not the layout the
        # reference ever sees (12 leads) — compare with the left-to-right restatement instead
        want = np.stack([_seq_norm(want_p[i]) for i in range(B)])
    else:
        want = io_ref.windows_from_wfdb16(d, gain, base)
    assert np.array_equal(x, want, equal_nan=True)


def _seq_norm(p):
    out = np.empty_like(p)
    for l in range(p.shape[0]):
        n = np.float32(p.shape[1])
        mean = np.float32(np.cumsum(p[l], dtype=np.float32)[-1] / n)
        devi = (p[l] - mean).astype(np.float32)
        std = np.float32(np.sqrt(np.float32(np.cumsum(devi * devi, dtype=np.float32)[-1] / n)) + np.float32(1e-6))
        out[l] = devi / std
    return out


def test_invalid_samples_and_flat_leads(hip):
    d = np.zeros((2, 100, 12), np.int16)
    d[0] = np.arange(1200).reshape(100, 12) % 37
    d[0, 50, 3] = -32768                   # format-16 invalid sample -> NaN poisons that lead only (as numpy)
    d[1, :, 5] = 123                       # flat lead: std ~ 0, the 1e-6 floor divides the float32 rounding residue of the mean
    gain, base = np.full((2, 12), 1000.0), np.zeros((2, 12), np.int32)
    x = host(hip.wfdb16_to_windows(dev(d), dev(gain), dev(base)))
    want = io_ref.windows_from_wfdb16(d, gain, base)
    assert np.array_equal(x, want, equal_nan=True)
    assert np.isnan(x[0, 3]).all() and not np.isnan(np.delete(x[0], 3, axis=0)).any()
    # a flat lead comes out constant, finite and tiny — not exactly 0: the left-to-right float32 mean of
    # 100 x 0.123 is not 0.123, and the reference divides that residue by (0 + 1e-6); reproduced exactly above
    assert np.isfinite(x[1, 5]).all() and (x[1, 5] == x[1, 5, 0]).all() and abs(x[1, 5, 0]) < 1.0


def test_zscore_rows_in_place_and_stats(hip):
    g = golden("g3_eval_known_answer")
    x = (g["ecg"][0] * 37.5 + 4.0).astype(np.float32)            # un-normalised-looking [12, 5000]
    xt = np.ascontiguousarray(x.T).T                             # the strided view the reference normalises
    want = io_ref.normalize_per_lead(xt)
    out, stats = hip.zscore_per_lead(dev(x), return_stats=True)
    assert np.array_equal(host(out), want)
    assert np.array_equal(host(stats)[:, 0], xt.mean(axis=1))
    assert np.array_equal(host(stats)[:, 1], xt.std(axis=1) + np.float32(1e-6))
    buf = dev(x)
    assert hip.zscore_per_lead(buf, out=buf).data_ptr() == buf.data_ptr()
    assert np.array_equal(host(buf), want)
    # idempotence up to rounding: a z-scored window is a fixed point (mean 0, std 1/(1+1e-6))
    again = host(hip.zscore_per_lead(buf))
    np.testing.assert_allclose(again, want, rtol=2e-5, atol=5e-5)


def test_full_batch_properties(hip):
    """BASELINE.json batch (256 windows, 12x5000 and 12x1000): per-lead mean 0 / std 1, invariance to
    the calibration (gain, baseline cancel in a z-score), and agreement with the oracle on a sample."""
    rng = np.random.default_rng(11)
    for T in (1000, 5000):
        d = rng.integers(-3000, 3000, size=(256, T, 12)).astype(np.int16)
        gain, base = np.full((256, 12), 1000.0), np.zeros((256, 12), np.int32)
        x = hip.wfdb16_to_windows(dev(d), dev(gain), dev(base))
        m, s = x.double().mean(dim=2), x.double().std(dim=2, unbiased=False)
        assert float(m.abs().max()) < 1e-5 and float((s - 1).abs().max()) < 1e-4
        x2 = hip.wfdb16_to_windows(dev(d), dev(gain * 0.25), dev(base))      # power-of-two gain: exact rescale
        np.testing.assert_allclose(host(x2), host(x), atol=2e-5)
        pick = [0, 100, 255]
        assert np.array_equal(host(x[pick]), io_ref.windows_from_wfdb16(d[pick], gain[pick], base[pick]))


def test_abi_argument_checks(hip):
    from ecg_hip import _lib as L
    with pytest.raises(L.EcgHipError, match="leads"):
        hip.wfdb16_to_windows(torch.zeros(1, 8, 17, dtype=torch.int16, device="cuda"),
                              torch.ones(1, 17, dtype=torch.float64, device="cuda"),
                              torch.zeros(1, 17, dtype=torch.int32, device="cuda"))
    with pytest.raises(L.EcgHipError, match="int16"):
        hip.wfdb16_to_windows(torch.zeros(1, 8, 12, device="cuda"), torch.ones(1, 12, dtype=torch.float64, device="cuda"),
                              torch.zeros(1, 12, dtype=torch.int32, device="cuda"))


def test_packed_loader_end_to_end(hip, tmp_path):
    from ecg_hip import pack
    rng = np.random.default_rng(3)
    n, T = 37, 1000
    d = rng.integers(-2500, 2500, size=(n, T, 12)).astype(np.int16)
    gain, base = np.full((n, 12), 1000.0), rng.integers(-9, 9, size=(n, 12)).astype(np.int32)
    y = (rng.random((n, 5)) < 0.3).astype(np.float32)
    dm = rng.random((n, 5)).astype(np.float32)
    path = str(tmp_path / "t.ecgpack")
    pack.write_pack(path, d, gain, base, y, dm)
    want = io_ref.windows_from_wfdb16(d, gain, base)
    # sequential, ragged last batch (drop_last unset in the reference's loaders)
    ld = pack.PackedBatchLoader(path, 8)
    assert len(ld) == 5
    got = list(ld)
    assert [b[0].shape[0] for b in got] == [8, 8, 8, 8, 5]
    assert all(t.is_cuda for b in got for t in b)
    assert np.array_equal(np.concatenate([host(b[0]) for b in got]), want)
    assert np.array_equal(np.concatenate([host(b[1]) for b in got]), dm)
    assert np.array_equal(np.concatenate([host(b[2]) for b in got]), y)
    # shuffled, two ranks: every record once (plus wrap padding), matching its own labels
    seen = []
    for r in range(2):
        ld = pack.PackedBatchLoader(path, 8, shuffle=True, seed=5, rank=r, world_size=2, with_demo=False)
        ld.set_epoch(2)
        idx = ld.indices()
        batches = list(ld)
        xs = np.concatenate([host(b[0]) for b in batches])
        ys = np.concatenate([host(b[1]) for b in batches])
        assert np.array_equal(xs, want[idx]) and np.array_equal(ys, y[idx])
        seen.append(idx)
    assert set(np.concatenate(seen)) == set(range(n))


def test_loader_feeds_the_training_loop(hip, tmp_path):
    """The reference's step driver consumes the loader unchanged (src/training/loop_demo.py:13-43)."""
    from ecg_hip import pack
    from ecg_hip.optim import FlatAdamW
    from src.models.ecg_multimodal import ECGMultimodal
    from src.training.loop_demo import train_one_epoch_demo
    from src.utils.seed import set_seed
    rng = np.random.default_rng(8)
    n, T = 16, 1000
    d = rng.integers(-2500, 2500, size=(n, T, 12)).astype(np.int16)
    path = str(tmp_path / "t.ecgpack")
    pack.write_pack(path, d, np.full((n, 12), 1000.0), np.zeros((n, 12), np.int32),
                    (rng.random((n, 5)) < 0.3).astype(np.float32), rng.random((n, 5)).astype(np.float32))
    set_seed(42)
    model = ECGMultimodal(in_leads=12, feat_dim=256, demo_dim=5, num_labels=5, demo_hidden_dim=64).cuda()
    opt = FlatAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    losses = [train_one_epoch_demo(model, pack.PackedBatchLoader(path, 8, shuffle=True, seed=1), opt, "cuda")
              for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_packed_loader_has_a_dataset_for_the_single_input_loops(hip, tmp_path):
    """train_one_epoch / eval_one_epoch end with len(loader.dataset) (src/training/loop.py:38,73): the packed
    loader without demographics must give the same epoch loss as a DataLoader over the same windows."""
    from ecg_hip import pack
    from src.models.ecg_cnn import ECGCNN
    from src.training.loop import eval_one_epoch, train_one_epoch
    from src.utils.seed import set_seed
    rng = np.random.default_rng(11)
    n, T = 21, 1000                                   # ragged last batch: 8 + 8 + 5
    d = rng.integers(-2500, 2500, size=(n, T, 12)).astype(np.int16)
    gain, base = np.full((n, 12), 1000.0), np.zeros((n, 12), np.int32)
    y = (rng.random((n, 5)) < 0.3).astype(np.float32)
    path = str(tmp_path / "s.ecgpack")
    pack.write_pack(path, d, gain, base, y, None)
    ld = pack.PackedBatchLoader(path, 8)
    assert len(ld.dataset) == n and len(ld) == 3
    ld2 = pack.PackedBatchLoader(path, 8, rank=1, world_size=2, drop_last=True)
    assert len(ld2.dataset) == 8 and len(ld2) == 1      # 21 -> wrap-padded to 22 -> 11 per rank -> one full batch
    xw = torch.from_numpy(io_ref.windows_from_wfdb16(d, gain, base))
    ds = torch.utils.data.TensorDataset(xw, torch.from_numpy(y))
    losses = []
    for loader in (ld, torch.utils.data.DataLoader(ds, batch_size=8, shuffle=False)):
        set_seed(42)
        model = ECGCNN(num_labels=5).cuda()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
        tr = train_one_epoch(model, loader, opt, "cuda")
        ev = eval_one_epoch(model, loader, "cuda")
        losses.append((tr, ev["bce_loss"]))
    assert abs(losses[0][0] - losses[1][0]) < 1e-6 and abs(losses[0][1] - losses[1][1]) < 1e-6, losses
