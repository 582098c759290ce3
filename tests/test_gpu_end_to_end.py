"""The whole path in one piece on synthetic records: WFDB format-16 files -> packed cache -> device batch
loader (int16 over PCIe, z-score on the GPU) -> ECGMultimodal trained through the reference's loop API with
FlatAdamW -> eval loop + metrics -> checkpoint loaded into the stock-torch restatement (state_dict
interchange), which must reproduce the GPU model's predictions."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_models as R

pytestmark = pytest.mark.gpu


def _synthetic_records(root, n, T, rng):
    """12-lead int16 records whose labels are recoverable: label c adds a sinusoid of its own frequency on
    two leads; demographics leak label 4.  Returns rel paths, labels [n,5], demo rows."""
    from ecg_hip import wfdb16
    rels, ys, rows = [], [], []
    t = np.arange(T) / 100.0
    for i in range(n):
        y = (rng.random(5) < 0.35).astype(np.float32)
        sig = rng.standard_normal((T, 12)) * 40.0
        for c in range(4):
            if y[c]:
                sig[:, 2 * c] += 220.0 * np.sin(2 * np.pi * (1.5 + 1.3 * c) * t + rng.uniform(0, 6.28))
                sig[:, 2 * c + 1] += 160.0 * np.cos(2 * np.pi * (1.5 + 1.3 * c) * t)
        sig += rng.integers(-300, 300, size=(1, 12))            # per-lead DC offset: the z-score removes it
        rel = f"records100/{i // 50:05d}/{i:05d}_lr"
        os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True)
        wfdb16.write_record(os.path.join(root, rel), np.clip(np.round(sig), -32000, 32000).astype(np.int16), 100,
                            np.full(12, 1000.0), np.zeros(12, np.int32))
        rels.append(rel)
        ys.append(y)
        rows.append({"age": 80.0 if y[4] else 30.0, "sex": "M" if i % 2 else "F", "height": 170 + i % 20,
                     "weight": 60 + i % 30, "pacemaker": 0})
    return rels, np.stack(ys), rows


def test_records_to_trained_model_to_stock_torch(tmp_path):
    from ecg_hip import pack
    from ecg_hip.optim import FlatAdamW
    from src.models.ecg_multimodal import ECGMultimodal
    from src.training.loop_demo import eval_one_epoch_demo, train_one_epoch_demo
    from src.utils.seed import set_seed
    rng = np.random.default_rng(0)
    T = 1000
    rels, y, rows = _synthetic_records(str(tmp_path), 192, T, rng)
    demo = pack.build_demo_matrix(rows)
    tr, va = slice(0, 160), slice(160, 192)
    for name, sl in (("train", tr), ("val", va)):
        pack.build_pack_from_wfdb(str(tmp_path / f"{name}.ecgpack"), str(tmp_path), rels[sl], y[sl], demo[sl],
                                  ids=np.arange(192)[sl])
    train = pack.PackedBatchLoader(str(tmp_path / "train.ecgpack"), 32, shuffle=True, seed=7)
    val = pack.PackedBatchLoader(str(tmp_path / "val.ecgpack"), 32)
    set_seed(42)
    model = ECGMultimodal().cuda()
    opt = FlatAdamW(model.parameters(), lr=2e-3, weight_decay=1e-4)
    losses = []
    for epoch in range(6):
        train.set_epoch(epoch)
        losses.append(train_one_epoch_demo(model, train, opt, "cuda"))
    assert losses[-1] < 0.6 * losses[0], losses
    out = eval_one_epoch_demo(model, val, "cuda")
    assert np.isfinite(out["bce_loss"]) and out["auroc_macro"] > 0.9, out
    # checkpoint interchange: the GPU-trained weights in the stock-torch model give the same predictions
    ckpt = str(tmp_path / "best.pth")
    torch.save(model.state_dict(), ckpt)
    ref = R.RefECGMultimodal()
    ref.load_state_dict(torch.load(ckpt, map_location="cpu"), strict=True)
    ref.eval()
    model.eval()
    (x, xd, yb), = list(pack.PackedBatchLoader(str(tmp_path / "val.ecgpack"), 32))
    with torch.no_grad():
        got = model(x, xd).cpu()
        want = ref(x.cpu(), xd.cpu())
    assert (got - want).abs().max() <= 1e-4
    assert torch.equal(torch.sigmoid(got) >= 0.5, torch.sigmoid(want) >= 0.5)
    assert torch.equal(yb.cpu(), torch.from_numpy(y[va]))
