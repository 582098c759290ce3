"""Single-input train / eval epochs with the semantics of the reference's src/training/loop.py
(train_one_epoch :14-38, eval_one_epoch :41-73), running the loss and the sigmoid on the HIP
path (ecg_bce_logits_fwd / ecg_sigmoid_fwd).

Epoch loss is sample-weighted: sum(loss_b * B_b) / len(loader.dataset).  The reference reads
`loss.item()` every step (a device sync per step); here the same double-precision running sum
is kept on the device and read once per epoch — same value, no per-step stall.

With an `ecg_hip.optim.FlatAdamW` on one GPU and an unhooked model the step CAN be replayed as one captured hipGraph
per batch shape (ecg_hip.graph.LoopStepper; opt-in: ECG_HIP_LOOP_GRAPH=1) — same kernels, same values, one host call
per step.  Measured: no faster than the eager loop at any batch size while the GPU is the bottleneck (graph.py).
"""
from typing import Dict

import numpy as np
import torch
from torch.utils.data import DataLoader

from ecg_hip import functional as hipF
from ecg_hip.graph import LoopStepper
from ecg_hip.optim import adopt_stock_adamw
from src.training.metrics import compute_metrics

try:
    from tqdm import tqdm
except ImportError:                      # progress bars are cosmetic
    def tqdm(it, **_):
        return it


def _logits(out):
    # models may return logits or (logits, features, ...)
    return out[0] if isinstance(out, tuple) else out


def _eager_step(model, optimizer, x, y, weighted):
    optimizer.zero_grad()
    # the BCE launch also does `weighted += loss * batch_size` (double, on the device)
    loss = hipF.binary_cross_entropy_with_logits(_logits(model(x)), y, weighted, x.size(0))
    hipF.backward_from_loss(loss)            # loss.backward() minus two one-element launches
    optimizer.step()


def train_one_epoch(model, loader: DataLoader, optimizer, device) -> float:
    model.train()
    optimizer = adopt_stock_adamw(optimizer)     # the scripts' torch.optim.AdamW steps through the fused flat launch
    stepper = LoopStepper.for_loop(model, optimizer, loss_weight_is_batch=True)
    weighted = None if stepper is None else stepper.running
    for x, y in tqdm(loader, desc="Train", leave=False, disable=None):
        x, y = x.to(device), y.to(device)
        if weighted is None:
            weighted = torch.zeros((), dtype=torch.float64, device=x.device)
        if stepper is None or not stepper.step((x, y)):
            _eager_step(model, optimizer, x, y, weighted)
    return (0.0 if weighted is None else weighted.item()) / len(loader.dataset)


def eval_one_epoch(model, loader: DataLoader, device) -> Dict[str, float]:
    model.eval()
    targets, probs, weighted = [], [], None
    with torch.no_grad():
        for x, y in tqdm(loader, desc="Eval", leave=False, disable=None):
            x, y = x.to(device), y.to(device)
            logits = _logits(model(x))
            if weighted is None:
                weighted = torch.zeros((), dtype=torch.float64, device=x.device)
            hipF.binary_cross_entropy_with_logits(logits, y, weighted, x.size(0))
            probs.append(hipF.sigmoid(logits))
            targets.append(y)
    y_true = torch.cat(targets).cpu().numpy()
    y_prob = torch.cat(probs).cpu().numpy()
    out = compute_metrics(y_true, y_prob, threshold=0.5)
    out["bce_loss"] = (0.0 if weighted is None else weighted.item()) / len(loader.dataset)
    return out
