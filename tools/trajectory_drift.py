#!/usr/bin/env python3
"""Drift of a few AdamW steps from the float64 trajectory: the HIP path next to the stock-torch CPU fp32 path, at the batch
sizes tests/test_gpu_model.py::test_trajectory_error_vs_float64_no_worse_than_the_cpu_fp32_path pins (the numbers DESIGN.md
section 2 quotes).  Prints one line per case: mean |parameter - float64| for both runs and max |loss - float64 loss|.

    python tools/trajectory_drift.py
"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    sys.path.insert(0, p)
from oracle import ref_models as R  # noqa: E402
from ecg_hip import functional as hipF  # noqa: E402
from ecg_hip.optim import FlatAdamW  # noqa: E402
from src.models.ecg_cnn import ECGCNN  # noqa: E402
from src.models.ecg_multimodal import ECGMultimodal  # noqa: E402
from src.utils.seed import set_seed  # noqa: E402

for name, B, steps in (("cnn5", 32, 8), ("cnn5", 256, 3), ("mm", 32, 8), ("mm", 256, 3)):
    demo = name == "mm"
    ctor, rctor = (ECGMultimodal, R.RefECGMultimodal) if demo else (lambda: ECGCNN(num_labels=5), lambda: R.RefECGCNN(num_labels=5))
    lr = 1e-3
    batch = R.synthetic_batch(B, 1000, 5, demo=demo)
    set_seed(42)
    model = ctor().cuda().train()
    R.seed_all(42)
    ref32 = rctor().train()
    ref64 = copy.deepcopy(ref32).double()
    b64 = tuple(t.double() for t in batch)
    opt = FlatAdamW(model.parameters(), lr=lr, weight_decay=1e-4)
    o32, o64 = R.make_adamw(ref32, lr, 1e-4), R.make_adamw(ref64, lr, 1e-4)
    db = [t.cuda() for t in batch]
    lh, l32, l64 = [], [], []
    for _ in range(steps):
        opt.zero_grad()
        loss = hipF.binary_cross_entropy_with_logits(model(*db[:-1]), db[-1])
        loss.backward()
        opt.step()
        lh.append(loss.item())
        l32.append(R.train_step(ref32, o32, batch)[1])
        l64.append(R.train_step(ref64, o64, b64)[1])
    th = tr = n = 0.0
    for (k, a), b32, b64_ in zip(model.state_dict().items(), ref32.state_dict().values(), ref64.state_dict().values()):
        if k.endswith("num_batches_tracked"):
            continue
        th += float((a.detach().cpu().double() - b64_).abs().sum())
        tr += float((b32.double() - b64_).abs().sum())
        n += a.numel()
    lh, l32, l64 = map(np.array, (lh, l32, l64))
    print(f"{name} B={B} steps={steps}: mean drift HIP {th / n:.3e}  CPU fp32 {tr / n:.3e}  (HIP/CPU {th / tr:.2f});  "
          f"max |loss - float64| HIP {np.abs(lh - l64).max():.2e}  CPU fp32 {np.abs(l32 - l64).max():.2e}", flush=True)
