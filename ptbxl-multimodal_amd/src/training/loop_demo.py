"""ECG + demographics train / eval epochs with the semantics of the reference's
src/training/loop_demo.py (train_one_epoch_demo :13-43, eval_one_epoch_demo :46-85).

Unlike loop.py the epoch loss here is the plain mean of per-batch means, and a module-level
`bce_loss_fn` object is kept because callers import it.  The running sum lives on the device
in double precision and is read once per epoch (the reference syncs twice per step).
With a FlatAdamW on one GPU and an unhooked model the step is replayed as one captured hipGraph per batch
shape (ecg_hip.graph.LoopStepper, as in loop.py; opt-in: ECG_HIP_LOOP_GRAPH=1).
"""
import numpy as np
import torch

from ecg_hip import functional as hipF
from ecg_hip.graph import LoopStepper
from ecg_hip.optim import adopt_stock_adamw
from src.training.metrics import compute_metrics

try:
    from tqdm import tqdm
except ImportError:
    def tqdm(it, **_):
        return it


class _HipBCEWithLogits(torch.nn.Module):
    """BCEWithLogitsLoss() (mean) on the HIP path."""

    def forward(self, logits, target, running=None, weight=1.0):
        return hipF.binary_cross_entropy_with_logits(logits, target, running, weight)


bce_loss_fn = _HipBCEWithLogits()


def _eager_step(model, optimizer, x_ecg, x_demo, y, running):
    optimizer.zero_grad()
    loss = bce_loss_fn(model(x_ecg, x_demo), y, running, 1.0)    # running += loss inside the launch
    hipF.backward_from_loss(loss)            # loss.backward() minus two one-element launches
    optimizer.step()


def train_one_epoch_demo(model, loader, optimizer, device):
    model.train()
    optimizer = adopt_stock_adamw(optimizer)     # the scripts' torch.optim.AdamW steps through the fused flat launch
    stepper = LoopStepper.for_loop(model, optimizer, loss_weight_is_batch=False)
    running, batches = (None if stepper is None else stepper.running), 0
    for x_ecg, x_demo, y in tqdm(loader, desc="Train-ECG+Demo", leave=False, disable=None):
        x_ecg, x_demo, y = x_ecg.to(device), x_demo.to(device), y.to(device)
        if running is None:
            running = torch.zeros((), dtype=torch.float64, device=x_ecg.device)
        if stepper is None or not stepper.step((x_ecg, x_demo, y)):
            _eager_step(model, optimizer, x_ecg, x_demo, y, running)
        batches += 1
    return (0.0 if running is None else running.item()) / max(1, batches)


def eval_one_epoch_demo(model, loader, device):
    model.eval()
    running, batches, probs, targets = None, 0, [], []
    with torch.no_grad():
        for x_ecg, x_demo, y in tqdm(loader, desc="Val-ECG+Demo", leave=False, disable=None):
            x_ecg, x_demo, y = x_ecg.to(device), x_demo.to(device), y.to(device)
            logits = model(x_ecg, x_demo)
            if running is None:
                running = torch.zeros((), dtype=torch.float64, device=x_ecg.device)
            bce_loss_fn(logits, y, running, 1.0)
            batches += 1
            probs.append(hipF.sigmoid(logits))
            targets.append(y)
    out = compute_metrics(torch.cat(targets).cpu().numpy(), torch.cat(probs).cpu().numpy())
    out["bce_loss"] = float((0.0 if running is None else running.item()) / max(1, batches))
    return out
