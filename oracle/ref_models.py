"""Stock-PyTorch CPU restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
The reference's Python files cannot travel to the GPU box, so this module restates,
with stock torch.nn layers and torch autograd, what these reference items compute:

  * ConvBlock / ECGCNN                — src/models/ecg_cnn.py:10-20, 32-68
  * ECGBackbone / DemoEncoder / FiLM  — src/models/ecg_multimodal.py:19-41, 44-59, 67-99
  * the train-step body               — src/training/loop.py:23-36, loop_demo.py:24-41
  * optimizer construction            — scripts/03_train_ecg_baseline.py:130-133

Module/attribute names match the reference so state_dicts are interchangeable; it is
pinned against the reference import by tests/golden (make_golden.py) and is the
large-shape checker + the "port" CPU baseline of bench.py.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

_WIDTHS = (32, 64, 128, 256)


class _Block(nn.Module):
    def __init__(self, cin, cout, k=15, p=2):
        super().__init__()
        self.net = nn.Sequential(OrderedDict([
            ("0", nn.Conv1d(cin, cout, k, padding=k // 2)),
            ("1", nn.BatchNorm1d(cout)),
            ("2", nn.ReLU(inplace=True)),
            ("3", nn.MaxPool1d(p)),
        ]))

    def forward(self, x):
        return self.net(x)


def _stack(in_leads):
    chans = (in_leads,) + _WIDTHS
    return nn.Sequential(*[_Block(a, b) for a, b in zip(chans[:-1], chans[1:])])


class RefECGCNN(nn.Module):
    def __init__(self, in_leads=12, feat_dim=256, num_labels=3):
        super().__init__()
        self.backbone = _stack(in_leads)
        self.gap = nn.AdaptiveAvgPool1d(1)
        self.proj = nn.Linear(_WIDTHS[-1], feat_dim)
        self.head = nn.Linear(feat_dim, num_labels)

    def forward(self, x, return_features=False):
        z = self.proj(self.gap(self.backbone(x)).squeeze(-1))
        logits = self.head(z)
        return (logits, z) if return_features else logits


class _RefBackbone(nn.Module):
    def __init__(self, in_leads=12, feat_dim=256):
        super().__init__()
        self.backbone = _stack(in_leads)
        self.gap = nn.AdaptiveAvgPool1d(1)
        self.proj = nn.Linear(_WIDTHS[-1], feat_dim)

    def forward(self, x):
        return self.proj(self.gap(self.backbone(x)).squeeze(-1))


class _RefDemo(nn.Module):
    def __init__(self, demo_dim=5, hidden_dim=64):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(demo_dim, 64), nn.ReLU(inplace=True),
                                 nn.Linear(64, hidden_dim), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.mlp(x)


class RefECGMultimodal(nn.Module):
    def __init__(self, in_leads=12, feat_dim=256, demo_dim=5, num_labels=5,
                 demo_hidden_dim=64, ecg_feat_dim=None, **_):
        super().__init__()
        feat_dim = feat_dim if ecg_feat_dim is None else ecg_feat_dim
        self.ecg_backbone = _RefBackbone(in_leads, feat_dim)
        self.demo_encoder = _RefDemo(demo_dim, demo_hidden_dim)
        self.film_gen = nn.Linear(demo_hidden_dim, 2 * feat_dim)
        self.head = nn.Linear(feat_dim, num_labels)

    def forward(self, x_ecg, x_demo):
        z = self.ecg_backbone(x_ecg)
        g_raw, beta = self.film_gen(self.demo_encoder(x_demo)).chunk(2, dim=-1)
        return self.head((1.0 + torch.tanh(g_raw)) * z + beta)


def seed_all(seed=42):
    """src/utils/seed.py:7-14 (CPU part)."""
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def synthetic_batch(B, T, C, gen_seed=1234, demo=False):
    """SURVEY §8(d) synthetic inputs: x~N(0,1), y~Bernoulli(0.3), demo~U(0,1)."""
    g = torch.Generator().manual_seed(gen_seed)
    x = torch.randn(B, 12, T, generator=g)
    y = (torch.rand(B, C, generator=g) < 0.3).float()
    if demo:
        return x, torch.rand(B, 5, generator=g), y
    return x, y


def train_step(model, opt, batch):
    """zero_grad -> forward -> BCE-with-logits(mean) -> backward -> step -> loss.item()."""
    opt.zero_grad()
    *inp, y = batch
    out = model(*inp)
    logits = out[0] if isinstance(out, tuple) else out
    loss = F.binary_cross_entropy_with_logits(logits, y)
    loss.backward()
    opt.step()
    return logits.detach(), float(loss.item())


def make_adamw(model, lr, weight_decay):
    return torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay)
