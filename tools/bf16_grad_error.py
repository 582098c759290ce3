#!/usr/bin/env python3
"""Per-tensor gradient error of the opt-in bf16 train step against the fp32 CPU oracle (rel-L2 and 1 - cos), for the cases of
tests/test_gpu_model.py::test_bf16_mixed_precision_train_step_config5 — the measurements its bars are set from.

    python tools/bf16_grad_error.py out.json
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ptbxl-multimodal_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import ref_models as R
from ecg_hip import functional as hipF
from src.models.ecg_cnn import ECGCNN
from src.utils.seed import set_seed
out = {}
for (B, T) in [(8, 5000), (19, 1000), (256, 5000)]:
    for act in (True,):          # (round 5: the mixed-precision step has one form, bf16 rows between the kernels)
        set_seed(42); model = ECGCNN(num_labels=1).cuda().train()
        R.seed_all(42); ref = R.RefECGCNN(num_labels=1).train()
        x, y = R.synthetic_batch(B, T, 1)
        with hipF.conv_precision("bf16"):
            logits = model(x.cuda()); loss = hipF.binary_cross_entropy_with_logits(logits, y.cuda()); loss.backward()
        rl = ref(x); rloss = torch.nn.functional.binary_cross_entropy_with_logits(rl, y); rloss.backward()
        rec = {"dlogit": float((logits.detach().cpu() - rl.detach()).abs().max()), "dloss": abs(loss.item() - rloss.item())}
        for (k, a), (_, b) in zip(model.named_parameters(), ref.named_parameters()):
            if ".net.0.bias" in k: continue
            g, r = a.grad.cpu().numpy().ravel().astype(np.float64), b.grad.numpy().ravel().astype(np.float64)
            rec[k] = [float(np.linalg.norm(g - r) / (np.linalg.norm(r) + 1e-12)), float(1 - (g @ r) / (np.linalg.norm(g) * np.linalg.norm(r) + 1e-20))]
        out[f"{B}x{T}x{int(act)}"] = rec
        print(B, T, act, json.dumps(rec), flush=True)
json.dump(out, open(sys.argv[1], "w"), indent=1)
