"""Macro AUROC / AUPRC / F1 for multi-label predictions (reference src/training/metrics.py:5-42).
Epoch-end, host-side, O(N*C) on a few thousand rows: stays on sklearn."""
import numpy as np
from sklearn import metrics as skm


def _nan_on_value_error(fn, *args, **kw):
    try:
        return fn(*args, **kw)
    except ValueError:          # e.g. a label column with a single class present
        return float("nan")


def compute_metrics(y_true: np.ndarray, y_prob: np.ndarray, threshold: float = 0.5):
    """y_true, y_prob: [N, L].  Returns {"auroc_macro", "auprc_macro", "f1_macro"}."""
    hard = (y_prob >= threshold).astype(int)
    return {
        "auroc_macro": _nan_on_value_error(skm.roc_auc_score, y_true, y_prob, average="macro"),
        "auprc_macro": _nan_on_value_error(skm.average_precision_score, y_true, y_prob, average="macro"),
        "f1_macro": skm.f1_score(y_true, hard, average="macro", zero_division=0),
    }
