"""Seeding helper with the behaviour of the reference's src/utils/seed.py:7-14."""
import random

import numpy as np
import torch


def set_seed(seed: int = 42) -> None:
    """Seed python, numpy and torch (all devices) and pin the deterministic backend flags."""
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
